"""A/B of two installed trees on one box: python3 scripts/ab_host.py <tree> ... (e.g. damar_amd build/prev_tree; a tree
holds bin/daligner and the libdamar_hip.so its rpath names).  Per tree: the contract command alone and two in a row.
(Never LD_PRELOAD one build of the library over another: two copies of the kernels and their device globals in one
process end in a GPU memory fault.)"""
import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from damar_amd import api
work = tempfile.mkdtemp(prefix="ab_", dir="/dev/shm")
cfg = bench.CONFIGS[2]
nb = api.sim_write_db(work, "SIM", cfg["genome"], coverage=cfg["coverage"], seed=cfg["seed"], block_mbp=cfg["block"])
with open(os.path.join(work, "plan.txt"), "w") as f:
    f.write(bench.plan_text("SIM", nb))
def clean():
    for root, dirs, files in os.walk(work):
        for f in files:
            if f.endswith(".las"): os.remove(os.path.join(root, f))
for rep in range(3):
    for libdir in sys.argv[1:]:
        env = dict(os.environ)
        exe = os.path.join(bench.ROOT, libdir, "bin", "daligner")
        out = []
        for n in (1, 1, 2):
            clean()
            t0 = time.time()
            for _ in range(n):
                subprocess.run([exe, "-P", "plan.txt"], cwd=work, env=env, check=True, stdout=subprocess.DEVNULL)
            out.append((time.time() - t0) / n)
            time.sleep(1.0)
        tidy = []
        for _ in range(3):                               # to process exit, like the reference is timed
            clean()
            t0 = time.time()
            subprocess.run([exe, "-P", "plan.txt"], cwd=work, env=dict(env, DAMAR_PLAN_TIDY="1"), check=True, stdout=subprocess.DEVNULL)
            tidy.append(time.time() - t0)
            time.sleep(1.0)
        print("%-16s returns after %.3f %.3f   two in a row, per command %.3f   to process exit (tidy) %s"
              % (libdir, out[0], out[1], out[2], " ".join("%.3f" % t for t in tidy)), flush=True)
shutil.rmtree(work)

"""The contract command (daligner -P plan, config 2) under GPU_MAX_HW_QUEUES = default / 2 / 3 on one box: seconds until the
command returns and to process exit (DAMAR_PLAN_TIDY=1), three rounds; then the in-process step with the same settings
(python3 scripts/ab_hwq.py).  The HIP runtime reads the variable when it comes up."""
import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from damar_amd import api
work = tempfile.mkdtemp(prefix="hwq_", dir="/dev/shm")
cfg = bench.CONFIGS[2]
nb = api.sim_write_db(work, "SIM", cfg["genome"], coverage=cfg["coverage"], seed=cfg["seed"], block_mbp=cfg["block"])
with open(os.path.join(work, "plan.txt"), "w") as f:
    f.write(bench.plan_text("SIM", nb))
def clean():
    for root, dirs, files in os.walk(work):
        for f in files:
            if f.endswith(".las"): os.remove(os.path.join(root, f))
exe = os.path.join(bench.ROOT, "damar_amd", "bin", "daligner")
for rep in range(3):
    for q in (None, "2", "3", "1"):
        env = dict(os.environ)
        if q: env["GPU_MAX_HW_QUEUES"] = q
        ret, tidy = [], []
        for _ in range(2):
            clean(); t0 = time.time()
            subprocess.run([exe, "-P", "plan.txt"], cwd=work, env=env, check=True, stdout=subprocess.DEVNULL)
            ret.append(time.time() - t0); time.sleep(1.0)
        for _ in range(3):
            clean(); t0 = time.time()
            subprocess.run([exe, "-P", "plan.txt"], cwd=work, env=dict(env, DAMAR_PLAN_TIDY="1"), check=True, stdout=subprocess.DEVNULL)
            tidy.append(time.time() - t0); time.sleep(1.0)
        print("GPU_MAX_HW_QUEUES=%-7s returns after %s   to process exit %s" % (q or "default", " ".join("%.3f" % t for t in ret), " ".join("%.3f" % t for t in tidy)), flush=True)
shutil.rmtree(work)
for q in (None, "2", "3"):
    env = dict(os.environ)
    if q: env["GPU_MAX_HW_QUEUES"] = q
    r = subprocess.run([sys.executable, os.path.join(bench.ROOT, "bench.py"), "--no-cpu", "--no-legs", "--no-e2e", "--no-trace"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    import json
    l = [x for x in r.stdout.split("\n") if x.startswith("{")]
    d = json.loads(l[-1]) if l else {}
    print("GPU_MAX_HW_QUEUES=%-7s in-process ms per step %s (synced %s) parity %s" % (q or "default", d.get("ms_per_step"), d.get("ms_per_step_synced"), (d.get("parity") or {}).get("identical")), flush=True)

"""Timeline of the contract command (`daligner -P plan` on the config-2 database, cold): DAMAR_HOSTPROF=1 output of the worker."""
import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from damar_amd import api
base = "/dev/shm"
work = tempfile.mkdtemp(prefix="e2eprof_", dir=base)
cfg = bench.CONFIGS[2]
nb = api.sim_write_db(work, "SIM", cfg["genome"], coverage=cfg["coverage"], seed=cfg["seed"], block_mbp=cfg["block"])
with open(os.path.join(work, "plan.txt"), "w") as f:
    f.write(bench.plan_text("SIM", nb))
exe = os.path.join(bench.ROOT, "damar_amd", "bin", "daligner")
for env in ({}, {"DAMAR_HOSTPROF": "1"}, {}):
    for f in os.listdir(work):
        if f.endswith(".las"): os.remove(os.path.join(work, f))
    e = dict(os.environ); e.update(env)
    t0 = time.time()
    r = subprocess.run([exe, "-P", "plan.txt"], cwd=work, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print(env, "wall %.3f" % (time.time() - t0), "rc", r.returncode)
    if env: print(r.stderr[-6000:])
    time.sleep(0.8)
shutil.rmtree(work)

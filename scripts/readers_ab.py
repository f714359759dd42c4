"""The contract command with 1..4 block reader threads (DAMAR_PLAN_READERS), alternating, cold runs a second apart."""
import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from damar_amd import api
work = tempfile.mkdtemp(prefix="rd_", dir="/dev/shm")
cfg = bench.CONFIGS[2]
nb = api.sim_write_db(work, "SIM", cfg["genome"], coverage=cfg["coverage"], seed=cfg["seed"], block_mbp=cfg["block"])
open(os.path.join(work, "plan.txt"), "w").write(bench.plan_text("SIM", nb))
exe = os.path.join(bench.ROOT, "damar_amd", "bin", "daligner")
res = {}
for rep in range(4):
    for n in (2, 4, 3, 1):
        e = dict(os.environ, DAMAR_PLAN_READERS=str(n), DAMAR_CLIPROF="1")
        t0 = time.time()
        r = subprocess.run([exe, "-P", "plan.txt"], cwd=work, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        dt = time.time() - t0
        res.setdefault(n, []).append(dt)
        time.sleep(1.2)
for n in sorted(res):
    print("readers %d: %s   best %.3f" % (n, " ".join("%.3f" % x for x in res[n]), min(res[n])))
shutil.rmtree(work)

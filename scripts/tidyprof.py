import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.getcwd())
import bench
from damar_amd import api
work = tempfile.mkdtemp(prefix="tp_", dir="/dev/shm")
cfg = bench.CONFIGS[2]
nb = api.sim_write_db(work, "SIM", cfg["genome"], coverage=cfg["coverage"], seed=cfg["seed"], block_mbp=cfg["block"])
open(os.path.join(work, "plan.txt"), "w").write(bench.plan_text("SIM", nb))
exe = os.path.join(bench.ROOT, "damar_amd", "bin", "daligner")
for rep in range(2):
    e = dict(os.environ, DAMAR_PLAN_TIDY="1", DAMAR_CLIPROF="1", DAMAR_INITPROF="1")
    t0 = time.time()
    r = subprocess.run([exe, "-P", "plan.txt"], cwd=work, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    print("tidy wall %.3f" % (time.time() - t0))
    print("\n".join(l for l in r.stderr.splitlines() if l.startswith(("cli:", "init:"))))
    time.sleep(1)
shutil.rmtree(work)

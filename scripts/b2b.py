"""Two (and three) contract commands in a row on the config-2 database, each command's own wall, with a pause between them:
what the early return of `daligner -P` (its forked worker tears down behind the caller) costs the next command."""
import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from damar_amd import api
work = tempfile.mkdtemp(prefix="b2b_", dir="/dev/shm")
cfg = bench.CONFIGS[2]
nb = api.sim_write_db(work, "SIM", cfg["genome"], coverage=cfg["coverage"], seed=cfg["seed"], block_mbp=cfg["block"])
with open(os.path.join(work, "plan.txt"), "w") as f:
    f.write(bench.plan_text("SIM", nb))
exe = os.path.join(bench.ROOT, "damar_amd", "bin", "daligner")
def run(env_extra=None):
    env = dict(os.environ); env.update(env_extra or {})
    t0 = time.time()
    subprocess.run([exe, "-P", "plan.txt"], cwd=work, env=env, check=True, stdout=subprocess.DEVNULL)
    return time.time() - t0
run(); time.sleep(1.5)
for pause in (0.0, 0.0, 0.1, 0.3, 0.0, 1.0):
    w = []
    for k in range(3):
        w.append(run())
        if pause: time.sleep(pause)
    print("pause %.1f s: %s" % (pause, "  ".join("%.3f" % x for x in w)), flush=True)
    time.sleep(1.5)
print("tidy: %.3f" % run({"DAMAR_PLAN_TIDY": "1"}))
shutil.rmtree(work)

"""A/B of the datander command of two installed trees on one box (see scripts/ab_host.py): python3 scripts/ab_tan.py <tree> ..."""
import os, sys, time, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from damar_amd import api
work = tempfile.mkdtemp(prefix="abt_", dir="/dev/shm")
nb = api.sim_write_db(work, "SIM", 27., coverage=20., seed=2, block_mbp=135, tandem_frac=0.3)
blocks = ["SIM.%d" % i for i in range(1, nb + 1)]
for rep in range(3):
    for tree in sys.argv[1:]:
        exe = os.path.join(bench.ROOT, tree, "bin", "datander")
        shutil.rmtree(os.path.join(work, "tan"), ignore_errors=True)
        t0 = time.time()
        subprocess.run([exe, "-j16"] + blocks, cwd=work, check=True, stdout=subprocess.DEVNULL)
        print("%-16s %.3f s" % (tree, time.time() - t0), flush=True)
        time.sleep(0.8)
shutil.rmtree(work)

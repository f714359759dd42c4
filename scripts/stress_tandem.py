"""One-off stress: datander option combinations on the tandem fixtures, GPU vs oracle_datander."""
import os, random, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, link_db
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for it in range(n):
    name = rng.choice(["tandem", "tandem", "tiny2", "noisy"])
    opts = ["-k%d" % rng.choice([8, 10, 11, 12, 13, 14, 16]), "-w%d" % rng.choice([2, 3, 4, 5]), "-h%d" % rng.choice([20, 28, 35, 50]),
            "-e%g" % rng.choice([.6, .7, .8]), "-l%d" % rng.choice([200, 400, 500, 1000]), "-s%d" % rng.choice([50, 100, 126]),
            "-j%d" % rng.choice([1, 2, 4, 8])]
    w = tempfile.mkdtemp(dir="/dev/shm")
    res = []
    for sub, exe in (("g", os.path.join(ROOT, "damar_amd", "bin", "datander")), ("o", os.path.join(ROOT, "oracle", "oracle_datander"))):
        d = os.path.join(w, sub); link_db(os.path.join(GOLDEN, name), d)
        subprocess.run([exe] + opts + ["G.1"], cwd=d, check=True, stdout=subprocess.DEVNULL, timeout=120)
        res.append(open(os.path.join(d, "tan", "G.1.G.1.las"), "rb").read())
    ok = res[0] == res[1]
    bad += not ok
    print(it, name, " ".join(opts), len(res[1]), "ok" if ok else "BAD", flush=True)
    shutil.rmtree(w)
print("combos", n, "bad", bad)

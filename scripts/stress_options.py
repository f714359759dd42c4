"""One-off stress: many random option combinations on several fixture DBs, GPU (in-process driver)
vs oracle_daligner; prints the first mismatch.  python scripts/stress_options.py [ncombos] [seed]"""
import os, random, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, link_db
from damar_amd import api, driver
api.lib().damar_hip_init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dbs = {"mask_dust": (2, ["dust", "rnd"]), "tandem": (1, []), "fusion": (1, []), "noisy": (1, [])}
bad = 0
for it in range(n):
    name = rng.choice(list(dbs))
    nb, tracks = dbs[name]
    o = dict(k=rng.choice([12, 13, 14, 15, 16, 17, 20]), w=rng.choice([3, 4, 5, 6, 7, 8]), h=rng.choice([20, 28, 35, 50, 70]),
             e=rng.choice([.6, .65, .7, .75, .8, .85]), l=rng.choice([300, 500, 1000, 2000, 4000]),
             s=rng.choice([40, 50, 100, 125, 126, 150, 200]), t=rng.choice([0, 0, 0, 5, 8, 20]), j=rng.choice([1, 2, 3, 4, 8, 16]),
             identity=rng.choice([0, 1]), symmetric=rng.choice([1, 1, 0]), biased=rng.choice([0, 0, 1]))
    o["masks"] = rng.choice([[], tracks[:1], tracks]) if tracks else []
    w = tempfile.mkdtemp(dir="/dev/shm")
    gdir, odir = os.path.join(w, "g"), os.path.join(w, "o")
    link_db(os.path.join(GOLDEN, name), gdir); link_db(os.path.join(GOLDEN, name), odir)
    blocks = {i: driver.Block(os.path.join(gdir, "G.%d" % i)) for i in range(1, nb + 1)}
    plan = driver.Plan(**o)
    for a, bs in driver.hpc_plan(nb):
        plan.run_line(blocks[a], [blocks[b] for b in bs], gdir)
    plan.finish()
    opts = ["-k%d" % o["k"], "-w%d" % o["w"], "-h%d" % o["h"], "-e%g" % o["e"], "-l%d" % o["l"], "-s%d" % o["s"], "-j%d" % o["j"]] + \
           (["-t%d" % o["t"]] if o["t"] else []) + (["-I"] if o["identity"] else []) + ([] if o["symmetric"] else ["-A"]) + \
           ["-m" + m for m in o["masks"]] + (["-b"] if o["biased"] else [])
    for a, bs in driver.hpc_plan(nb):
        subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner")] + opts + ["G.%d" % a] + ["G.%d" % b for b in bs],
                       cwd=odir, check=True, stdout=subprocess.DEVNULL)
    ok = True
    for dp, _, fs in os.walk(odir):
        for f in fs:
            if f.endswith(".las"):
                rel = os.path.relpath(os.path.join(dp, f), odir)
                if open(os.path.join(dp, f), "rb").read() != open(os.path.join(gdir, rel), "rb").read():
                    ok = False
                    print("MISMATCH", name, opts, rel, flush=True)
    # the records' next consumer: edit scripts of one of the files, random mode / PTS or MID, GPU vs oracle
    las = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(gdir) for f in fs if f.endswith(".las"))
    tr = ""
    if las:
        f = rng.choice(las)
        mode, mid = rng.choice([0, 0, 1, -1]), rng.choice([0, 1])
        g, oo = os.path.join(w, "g.bin"), os.path.join(w, "o.bin")
        r1 = subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "lastrace"), "-m%d" % mode] + (["-M"] if mid else []) +
                            [os.path.join(gdir, "G"), os.path.join(gdir, "G"), f, g])
        r2 = subprocess.run([os.path.join(ROOT, "oracle", "oracle_lastrace"), os.path.join(gdir, "G"), f, oo, str(mode)] +
                            (["mid"] if mid else []))
        same = r1.returncode == r2.returncode and (r1.returncode != 0 or open(g, "rb").read() == open(oo, "rb").read())
        tr = " trace(m%d,%s)=%s" % (mode, "MID" if mid else "PTS", "ok" if same else "BAD")
        if r1.returncode != 0:
            tr += "[both report a bad alignment]" if same else "[exit codes differ]"
        ok = ok and same
    bad += not ok
    print(it, name, " ".join(opts), ("ok" if ok else "BAD") + tr, flush=True)
    shutil.rmtree(w)
print("combos", n, "bad", bad)

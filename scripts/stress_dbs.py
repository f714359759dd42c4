"""One-off stress: random small simulated databases (several blocks, varying read lengths and
coverage), whole HPCdaligner plan, GPU in-process driver vs oracle_daligner."""
import os, random, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from damar_amd import api, driver
api.lib().damar_hip_init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for it in range(n):
    w = tempfile.mkdtemp(dir="/dev/shm")
    g, c = rng.choice([.05, .08, .12, .2]), rng.choice([6, 10, 16, 25])
    mean, sd, short = rng.choice([(3000, 800, 1000), (6000, 1500, 2000), (10000, 2000, 4000), (25000, 8000, 5000)])
    nb = api.sim_write_db(w, "R", g, coverage=c, seed=rng.randrange(1, 1000), block_mbp=1, rmean=mean, rsdev=sd, rshort=short,
                          erate=rng.choice([.10, .15, .18]))
    o = dict(j=rng.choice([1, 4, 16]), symmetric=rng.choice([1, 1, 0]), identity=rng.choice([0, 1]), t=rng.choice([0, 0, 15]),
             l=rng.choice([1000, 1500]))
    gdir = os.path.join(w, "g")
    opts = ["-k14", "-j%d" % o["j"], "-l%d" % o["l"]] + (["-t%d" % o["t"]] if o["t"] else []) + (["-I"] if o["identity"] else []) + \
           ([] if o["symmetric"] else ["-A"])
    if os.environ.get("DAMAR_STRESS_CLI"):     # the command-line driver (block reader thread, preloaded blocks) instead
        os.makedirs(gdir)
        for f in ("R.db", ".R.idx", ".R.bps"):
            os.symlink(os.path.join(w, f), os.path.join(gdir, f))
        for a, bs in driver.hpc_plan(nb):
            subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "daligner")] + opts + ["R.%d" % a] + ["R.%d" % b for b in bs],
                           cwd=gdir, check=True, stdout=subprocess.DEVNULL)
    else:
        blocks = {i: driver.Block(os.path.join(w, "R.%d" % i)) for i in range(1, nb + 1)}
        plan = driver.Plan(**o)
        for a, bs in driver.hpc_plan(nb):
            plan.run_line(blocks[a], [blocks[b] for b in bs], gdir)
        plan.finish()
    odir = os.path.join(w, "o"); os.makedirs(odir)
    for f in ("R.db", ".R.idx", ".R.bps"):
        os.symlink(os.path.join(w, f), os.path.join(odir, f))
    for a, bs in driver.hpc_plan(nb):
        subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner")] + opts + ["R.%d" % a] + ["R.%d" % b for b in bs],
                       cwd=odir, check=True, stdout=subprocess.DEVNULL)
    ok, nl = True, 0
    for dp, _, fs in os.walk(odir):
        for f in fs:
            if f.endswith(".las"):
                rel = os.path.relpath(os.path.join(dp, f), odir); nl += 1
                if open(os.path.join(dp, f), "rb").read() != open(os.path.join(gdir, rel), "rb").read():
                    ok = False; print("MISMATCH", rel, flush=True)
    bad += not ok
    print(it, "genome", g, "cov", c, "reads", mean, "blocks", nb, " ".join(opts), "files", nl, "ok" if ok else "BAD", flush=True)
    shutil.rmtree(w)
print("dbs", n, "bad", bad)

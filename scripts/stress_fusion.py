"""CPU stress of the PRODUCT's host tail (host/redundancy.c, host/bridge.c) behind the oracle's CPU front
(oracle/oracle_daligner_hosttail) against the real reference (oracle/_ref/daligner): fusion-heavy derived databases
(tests/golden/make_golden.py's `fusion<seed>` and `tandem` recipes) under random options.  Needs /root/reference-built
oracle/_ref, so it runs in the build container only.   python3 scripts/stress_fusion.py [n=8] [seed=1]"""
import os, random, subprocess, sys, tempfile, shutil, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = tot_f = tot_b = 0
for it in range(n):
    w = tempfile.mkdtemp(dir="/dev/shm")
    kind = rng.choice(["fusion%d" % rng.randrange(40, 4000)] * 3 + ["noisy"])
    dbdir, root = mg.derive(kind, w)
    opts = ["-k14", "-j%d" % rng.choice([1, 4]), "-s%d" % rng.choice([100, 100, 50, 126, 200])] + \
           rng.choice([[], ["-I"], ["-A"], ["-l500"]])
    outs = {}
    for tag, exe in (("ref", os.path.join(ROOT, "oracle", "_ref", "daligner")),
                     ("ht", os.path.join(ROOT, "oracle", "oracle_daligner_hosttail"))):
        d = os.path.join(w, tag); os.makedirs(d)
        for f in os.listdir(dbdir):
            if f.startswith(root + ".") or f.startswith("." + root + "."):
                os.symlink(os.path.join(dbdir, f), os.path.join(d, f))
        r = subprocess.run([exe] + (["-v"] if tag == "ht" else []) + opts + [root + ".1", root + ".1"], cwd=d, check=True,
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        if tag == "ht":
            m = re.findall(r"redundancy calls (\d+) fusions (\d+) bridges (\d+)", r.stdout)
            f_, b_ = sum(int(x[1]) for x in m), sum(int(x[2]) for x in m)
        las = {}
        for dp, _, fs in os.walk(d):
            for f in fs:
                if f.endswith(".las"):
                    las[os.path.relpath(os.path.join(dp, f), d)] = open(os.path.join(dp, f), "rb").read()
        outs[tag] = las
    ok = outs["ref"] == outs["ht"] and len(outs["ref"]) > 0
    bad += not ok; tot_f += f_; tot_b += b_
    print(it, kind, " ".join(opts), "files", len(outs["ref"]), "fusions", f_, "bridges", b_, "ok" if ok else "BAD", flush=True)
    shutil.rmtree(w)
print("dbs", n, "bad", bad, "fusions", tot_f, "bridges", tot_b)

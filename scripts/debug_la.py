"""Debug aid: the batch Local_Alignment entry (damar_local_alignment_batch) against the oracle on the seeds of a golden
block, printing the first mismatches field by field.  python scripts/debug_la.py [comp] [ntasks] [golden]"""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as O
from damar_amd import api
comp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ntask = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
gold = sys.argv[3] if len(sys.argv) > 3 else "indel"
if os.environ.get("DBG_LIBDIR"):               # an experiment build of the library (scripts/build_exp.sh)
    import damar_amd.lib as dl
    dl.lib_path = lambda: os.path.join(ROOT, os.environ["DBG_LIBDIR"], "libdamar_hip.so")
L = api.lib()
assert L.damar_hip_init(0) >= 1
L.Set_Filter_Params(14, 6, 0, 35, 4)
an = os.path.join(ROOT, "tests", "golden", gold, "G.1")
adb, bdb = api.read_block(an), api.read_block(an)
oadb, obdb = O.read_block(an), O.read_block(an)
if comp:
    L.damar_complement_block(C.byref(bdb), 1)
    O.lib().damar_complement_block(C.byref(obdb), 1)
prm = O.params()
pa, na, _ = O.sort_kmers(oadb, prm)
pb, nb, _ = O.sort_kmers(obdb, prm)
seeds = O.seed_pairs(oadb, obdb, pa, na, pb, nb, 0, comp, prm)
rng = random.Random(5 + comp)
pick = sorted(rng.sample(range(len(seeds)), min(ntask, len(seeds))))
tasks = []
for i in pick:
    s = seeds[i]
    if s["aread"] == s["bread"] and not comp:
        continue
    tasks += [int(s["aread"]), int(s["bread"]), int(s["diag"]), int(2 * s["apos"] - s["diag"])]
if os.environ.get("DBG_ONLY"):                 # only these task indexes (debugging a difference that depends on what shares the wavefront)
    keep = [int(x) for x in os.environ["DBG_ONLY"].split(",")]
    tasks = sum((tasks[4 * t:4 * t + 4] for t in keep), [])
nt = len(tasks) // 4
ospec = O.lib().New_Align_Spec(.70, 100, oadb.freq, 1, 1, 0, 0, 1)
spec = L.New_Align_Spec(.70, 100, adb.freq, 1, 1, 0, 0, 1)
ablk, bblk = L.damar_block_upload(C.byref(adb)), L.damar_block_upload(C.byref(bdb))
paths = (C.c_int * (12 * nt))()
toff = (api.c_int64 * (2 * nt))()
cap = nt * 1200
traces = (C.c_uint16 * cap)()
rc = L.damar_local_alignment_batch(ablk, bblk, comp, spec, (C.c_int * len(tasks))(*tasks), nt, paths, toff, traces, cap)
print("rc", rc, "tasks", nt)
maxtp = 4 * (max(adb.maxlen, bdb.maxlen) // 100 + 4)
bad = 0
kinds = {}
names = ["abpos", "bbpos", "aepos", "bepos", "diffs", "tlen", "B.abpos", "B.bbpos", "B.aepos", "B.bepos", "B.diffs", "B.tlen"]
for t in range(nt):
    ar, br, dg, anti = tasks[4 * t:4 * t + 4]
    want, wat, wbt = O.local_alignment(oadb, obdb, ar, br, comp, dg, anti, ospec, maxtp)
    got = list(paths[12 * t:12 * t + 12])
    gat = list(traces[toff[2 * t]:toff[2 * t] + got[5]])
    gbt = list(traces[toff[2 * t + 1]:toff[2 * t + 1] + got[11]])
    if got != want or gat != wat or gbt != wbt:
        bad += 1
        diff = tuple(n for n, g, w in zip(names, got, want) if g != w) + (("atrace",) if gat != wat else ()) + (("btrace",) if gbt != wbt else ())
        kinds[diff] = kinds.get(diff, 0) + 1
        if bad <= 6:
            print("task", t, "ar", ar, "br", br, "diag", dg, "anti", anti, "alen", adb.reads[ar].rlen, "blen", bdb.reads[br].rlen)
            print("  got ", got)
            print("  want", want)
            if gat != wat:
                print("  atrace got ", gat[:24], "...", gat[-8:])
                print("  atrace want", wat[:24], "...", wat[-8:])
            if gbt != wbt:
                print("  btrace got ", gbt[:24], "...", gbt[-8:])
                print("  btrace want", wbt[:24], "...", wbt[-8:])
print("bad", bad, "of", nt)
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1]):
    print(v, k)

if hasattr(L, "damar_dbg_read"):
    buf = (C.c_int * (12 * 500))()
    n = L.damar_dbg_read(buf)
    print("loop exits", n)
    for i in range(min(n, 500)):
        d = buf[12 * i:12 * i + 12]
        print("  slot %d reason %s la %d be %d ls %d hs %d dif(entry) %d m %d left %d bk %d kb %d besta(entry) %d" % (d[0], bin(d[1]), d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11]))

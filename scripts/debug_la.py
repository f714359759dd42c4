import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as O
from damar_amd import api
comp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nsel = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
L = api.lib(); L.damar_hip_init(0); L.Set_Filter_Params(14, 6, 0, 35, 4)
an = os.path.join(ROOT, os.environ.get("DBG_DB", "tests/golden/indel/G.1"))
adb, bdb = api.read_block(an), api.read_block(an)
oadb, obdb = O.read_block(an), O.read_block(an)
if comp:
    L.damar_complement_block(C.byref(bdb), 1); O.lib().damar_complement_block(C.byref(obdb), 1)
prm = O.params()
pa, na, _ = O.sort_kmers(oadb, prm); pb, nb, _ = O.sort_kmers(obdb, prm)
seeds = O.seed_pairs(oadb, obdb, pa, na, pb, nb, 0, comp, prm)
rng = random.Random(5 + comp); pick = sorted(rng.sample(range(len(seeds)), nsel))
tasks = []
for i in pick:
    s = seeds[i]
    if s["aread"] == s["bread"] and not comp: continue
    tasks += [int(s["aread"]), int(s["bread"]), int(s["diag"]), int(2 * s["apos"] - s["diag"])]
nt = len(tasks) // 4
print("tasks", nt, flush=True)
spec = L.New_Align_Spec(.70, 100, adb.freq, 1, 1, 0, 0, 1)
ablk, bblk = L.damar_block_upload(C.byref(adb)), L.damar_block_upload(C.byref(bdb))
print("uploaded", flush=True)
paths = (C.c_int * (12 * nt))(); toff = (api.c_int64 * (2 * nt))(); cap = nt * 1200; traces = (C.c_uint16 * cap)()
if len(sys.argv) > 3:
    for t in range(nt):
        one = tasks[4*t:4*t+4]
        print("task", t, one, flush=True)
        rc = L.damar_local_alignment_batch(ablk, bblk, comp, spec, (C.c_int * 4)(*one), 1, paths, toff, traces, cap)
        print("  rc", rc, list(paths[:12]), flush=True)
else:
    rc = L.damar_local_alignment_batch(ablk, bblk, comp, spec, (C.c_int * len(tasks))(*tasks), nt, paths, toff, traces, cap)
    print("batch rc", rc, flush=True)

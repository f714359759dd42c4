import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from damar_amd import api, driver
import oracle_api as O
L = api.lib(); L.damar_hip_init(0)
mode = sys.argv[1]
an = os.path.join(ROOT, "tests/golden/tiny2/G.1")
if mode == "pipe_first":
    tmp = tempfile.mkdtemp()
    for f in ("G.db", ".G.idx", ".G.bps"):
        os.symlink(os.path.join(ROOT, "tests/golden/tiny2", f), os.path.join(tmp, f))
    blk = driver.Block(os.path.join(tmp, "G.1"))
    plan = driver.Plan(j=4)
    plan.run_line(blk, [blk], tmp)
    print("pipeline ok", plan.counts, flush=True)
L.Set_Filter_Params(14, 6, 0, 35, 4)
adb, bdb = api.read_block(an), api.read_block(an)
spec = L.New_Align_Spec(.70, 100, adb.freq, 1, 1, 0, 0, 1)
ablk, bblk = L.damar_block_upload(C.byref(adb)), L.damar_block_upload(C.byref(bdb))
tasks = [12, 0, -2482, 10664]
paths = (C.c_int * 12)(); toff = (api.c_int64 * 2)(); traces = (C.c_uint16 * 4000)()
print("calling batch", flush=True)
rc = L.damar_local_alignment_batch(ablk, bblk, 1 if mode == "comp" else 0, spec, (C.c_int * 4)(*tasks), 1, paths, toff, traces, 4000)
print("rc", rc, list(paths), flush=True)

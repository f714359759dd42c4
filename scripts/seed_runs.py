"""How long are the (bread, aread) runs of a comparison's sorted seeds?  (what pair_work_mark's screen works on)
   python3 scripts/seed_runs.py c4|c2      -> prints seeds, read pairs, heads (runs of >= 3 seeds) and the histogram of their lengths"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from damar_amd import api  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "c4"
    w = tempfile.mkdtemp(dir="/dev/shm")
    if shape == "c4":
        subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "simdb"), w, "SIM", "248", "-c80", "-m15000", "-s3000", "-e.15", "-r4", "-S78", "-N2"],
                       check=True, stdout=subprocess.DEVNULL)
    else:
        api.sim_write_db(w, "SIM", 27., coverage=20., seed=2, block_mbp=135)
    L = api.lib()
    assert L.damar_hip_init(0) >= 1
    L.damar_set_async(0)
    L.Set_Filter_Params(14, 6, 0, 35, 8)
    api.set_globals()
    adb, bdb = api.read_block(os.path.join(w, "SIM.1")), api.read_block(os.path.join(w, "SIM.2"))
    n = C.c_int(0)
    ablk, bblk = L.damar_block_upload(C.byref(adb)), L.damar_block_upload(C.byref(bdb))
    aidx, bidx = L.damar_index_build(ablk, 0, C.byref(n)), L.damar_index_build(bblk, 0, C.byref(n))
    spec = L.New_Align_Spec(.70, 100, adb.freq, 8, 1, 0, 0, 1)
    L.damar_last_seeds(None, 1)
    cnt = (api.c_int64 * 3)()
    L.damar_match(C.byref(adb), C.byref(bdb), aidx, bidx, 0, 0, spec, cnt)
    dt = np.dtype([("diag", "<i4"), ("apos", "<i4"), ("aread", "<i4"), ("bread", "<i4")])
    got = np.zeros(int(cnt[0]), dtype=dt)
    L.damar_last_seeds(got.ctypes.data, len(got))
    L.damar_last_seeds(None, 0)
    pair = got["bread"].astype(np.int64) << 32 | got["aread"].astype(np.int64)
    starts = np.flatnonzero(np.concatenate(([True], pair[1:] != pair[:-1])))
    lens = np.diff(np.concatenate((starts, [len(pair)])))
    print("%s: %d seeds, %d read pairs, %d with >= 3 seeds (heads)" % (shape, len(pair), len(lens), int((lens >= 3).sum())))
    h = lens[lens >= 3]
    for lo, hi in ((3, 3), (4, 4), (5, 5), (6, 6), (7, 8), (9, 16), (17, 48), (49, 1 << 30)):
        m = (h >= lo) & (h <= hi)
        print("  runs of %2d..%-10d %9d  (%.2f %% of the heads, %.2f %% of the seeds)" % (lo, hi, int(m.sum()), 100. * m.sum() / max(1, len(h)), 100. * h[m].sum() / len(pair)))
    tile = 4096
    per = np.bincount(starts[lens >= 3] // tile, minlength=(len(pair) + tile - 1) // tile)
    print("  heads per tile of %d seeds: mean %.0f, max %d" % (tile, per.mean(), per.max()))
    subprocess.run(["rm", "-rf", w])


main()

"""How many seed pairs survive the first cut of the report stage?  One block pair of the bench's config-2 database:
fraction of the seeds that belong to (bread, aread) pairs with at least minhit = 3 seeds (filter.c:2257)."""
import ctypes as C, os, sys, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from damar_amd import api, driver
L = api.lib()
L.damar_hip_init(0)
work = tempfile.mkdtemp(dir="/dev/shm")
api.sim_write_db(work, "SIM", 27.0, coverage=20.0, seed=101, block_mbp=135)
a = driver.Block(os.path.join(work, "SIM.2")); b = driver.Block(os.path.join(work, "SIM.1"))
plan = driver.Plan()
n = C.c_int(0)
aidx = L.damar_index_build(a.upload(), 0, C.byref(n)); bidx = L.damar_index_build(b.upload(), 0, C.byref(n))
spec = L.New_Align_Spec(.70, 100, a.db.freq, 4, 1, 0, 0, 1)
L.damar_set_async(0)
L.damar_last_seeds(None, 1)
cnt = (api.c_int64 * 3)()
L.damar_match(C.byref(a.db), C.byref(b.db), aidx, bidx, 0, 0, spec, cnt)
dt = np.dtype([("diag", "<i4"), ("apos", "<i4"), ("aread", "<i4"), ("bread", "<i4")])
got = np.zeros(int(cnt[0]), dtype=dt)
L.damar_last_seeds(got.ctypes.data, len(got))
pid = got["bread"].astype(np.int64) << 32 | got["aread"]
chg = np.flatnonzero(np.diff(pid)) + 1
starts = np.concatenate([[0], chg]); lens = np.diff(np.concatenate([starts, [len(pid)]]))
for m in (2, 3, 4, 6):
    print("pairs with >= %d seeds: %d of %d pairs, %.2f %% of the %d seeds" % (m, (lens >= m).sum(), len(lens), 100. * lens[lens >= m].sum() / len(pid), len(pid)))
print("counts", list(cnt))
shutil.rmtree(work)

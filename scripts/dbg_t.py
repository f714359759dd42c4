import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as O
from damar_amd import api
L = api.lib(); L.damar_hip_init(0)
name, k, t = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
an = os.path.join(ROOT, "tests", "golden", name, "G.1")
adb, oadb = api.read_block(an), O.read_block(an)
prm = O.params(k=k, t=t, j=1)
pa, na, recs = O.sort_kmers(oadb, prm)
L.Set_Filter_Params(k, 6, t, 35, 1)
n = C.c_int(0)
blk = L.damar_block_upload(C.byref(adb))
idx = L.damar_index_build(blk, 0, C.byref(n))
print("oracle n", na, "gpu n", n.value)
buf = np.zeros(n.value, dtype=O.KMER_DT)
L.damar_index_download(idx, buf.ctypes.data)
m = min(na, n.value)
eq = (buf[:m] == recs[:m])
print("equal prefix", int(np.argmin(eq)) if not eq.all() else m)
if not eq.all() or na != n.value:
    i = int(np.argmin(eq)) if not eq.all() else m
    print("oracle", recs[max(0,i-3):i+4]); print("gpu", buf[max(0,i-3):i+4])

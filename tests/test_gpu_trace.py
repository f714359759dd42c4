"""(f)4 on the GPU: damar_trace_pts / Compute_Trace_PTS (kernels/trace_pts.hip) against the REAL reference.

The expected values are the md5 of the dumps oracle/ref_lastrace wrote around the reference's own
Compute_Trace_PTS (tests/golden/trace_ref_md5.txt); damar_amd/bin/lastrace writes the same format from the
GPU results, so equality is byte for byte: every edit-script value and every difference count."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, read_case

pytestmark = pytest.mark.gpu


def ref_lines(fixture="trace_ref_md5.txt"):
    out = []
    for ln in open(os.path.join(GOLDEN, fixture)):
        md5, name, las, mode = ln.split()
        out.append((md5, name, las, int(mode)))
    return out


def run_jobs(jobs, tmp_path, extra_env=None):
    """jobs = [(case dict, las, mode, mid)] -> md5 of each dump; one lastrace process for all of them."""
    env = dict(os.environ)
    env.update(extra_env or {})
    lst = str(tmp_path / "jobs.txt")
    with open(lst, "w") as f:
        for i, (c, las, mode, mid) in enumerate(jobs):
            db = os.path.join(c["dbdir"], "G")
            f.write("%d %d %s %s %s %s\n" % (mode, 1 if mid else 0, db, db, os.path.join(c["lasdir"], las), str(tmp_path / ("g%d.bin" % i))))
    subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "lastrace"), "-L" + lst], check=True, env=env)
    out = []
    for i in range(len(jobs)):
        p = str(tmp_path / ("g%d.bin" % i))
        out.append(hashlib.md5(open(p, "rb").read()).hexdigest())
        os.unlink(p)
    return out


@pytest.mark.parametrize("fixture,mid", [("trace_ref_md5.txt", False), ("trace_mid_ref_md5.txt", True)])
def test_trace_expansion_equals_reference_on_golden(built, tmp_path, fixture, mid):
    """Compute_Trace_PTS (align.c:5577) and Compute_Trace_MID (align.c:5694): every golden record, 3 modes."""
    lines = ref_lines(fixture)
    got = run_jobs([(read_case(name), las, mode, mid) for _, name, las, mode in lines], tmp_path)
    bad = [(name, las, mode, mid) for (md5, name, las, mode), g in zip(lines, got) if g != md5]
    assert not bad, bad
    assert len(lines) >= 100


def test_trace_expansion_small_stripes_take_the_deferred_launch(built, tmp_path):
    """With the slot kernel squeezed to 9 rows (waves 0..6) almost every segment is deferred and redone by
    the per-lane stripe kernel; results must not change."""
    jobs, want = [], []
    for fixture, mid in (("trace_ref_md5.txt", False), ("trace_mid_ref_md5.txt", True)):
        ref = {(n, l, m): h for h, n, l, m in ref_lines(fixture)}
        for name in ("tiny_I", "fusion", "tan_tandem", "tiny_s"):
            c = read_case(name)
            for las in c["las"]:
                for mode in (0, 1, -1):
                    jobs.append((c, las, mode, mid))
                    want.append(ref[(name, las, mode)])
    got = run_jobs(jobs, tmp_path, {"DAMAR_TRACE_ROWS": "9", "DAMAR_TRACE_BLOCKS": "7"})
    bad = [(j[0]["name"], j[1], j[2], j[3]) for j, g, w in zip(jobs, got, want) if g != w]
    assert not bad, bad


def test_trace_expansion_in_many_batches(built, tmp_path):
    """A file is cut into batches of at most 2^24 segments; with the limit lowered to 3000 the golden files take
    up to a hundred batches each, whose scripts, offsets and difference counts must concatenate unchanged."""
    jobs, want = [], []
    for fixture, mid in (("trace_ref_md5.txt", False), ("trace_mid_ref_md5.txt", True)):
        ref = {(n, l, m): h for h, n, l, m in ref_lines(fixture)}
        for name in ("tiny_I", "prod", "tan_tandem"):
            c = read_case(name)
            for las in c["las"]:
                jobs.append((c, las, 0, mid))
                want.append(ref[(name, las, 0)])
    got = run_jobs(jobs, tmp_path, {"DAMAR_TRACE_MAXSEGS": "3000"})
    bad = [(j[0]["name"], j[1], j[3]) for j, g, w in zip(jobs, got, want) if g != w]
    assert not bad, bad


def test_compute_trace_pts_c_abi_single_record(built):
    """align.h's Compute_Trace_PTS(align, work, tspace, mode) for single records against the oracle."""
    from damar_amd import api
    lib = api.lib()
    ora = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))

    class Path(ctypes.Structure):
        _fields_ = [("trace", ctypes.c_void_p), ("tlen", ctypes.c_int), ("diffs", ctypes.c_int),
                    ("abpos", ctypes.c_int), ("bbpos", ctypes.c_int), ("aepos", ctypes.c_int), ("bepos", ctypes.c_int)]

    class Alignment(ctypes.Structure):
        _fields_ = [("path", ctypes.POINTER(Path)), ("flags", ctypes.c_uint32), ("aseq", ctypes.c_void_p),
                    ("bseq", ctypes.c_void_p), ("alen", ctypes.c_int), ("blen", ctypes.c_int)]

    lib.New_Work_Data.restype = ctypes.c_void_p
    lib.Compute_Trace_PTS.argtypes = [ctypes.POINTER(Alignment), ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.Compute_Trace_MID.argtypes = [ctypes.POINTER(Alignment), ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    ora.oracle_compute_trace_mid.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.POINTER(Path), ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_int)]
    lib.Free_Work_Data.argtypes = [ctypes.c_void_p]
    ora.oracle_compute_trace_pts.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.POINTER(Path), ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(5)
    work = lib.New_Work_Data()
    tspace = 100
    for trial in range(6):
        alen = int(rng.integers(900, 2500))
        a = rng.integers(0, 4, alen).astype(np.int8)
        # B = A with indels and substitutions; trace points from the true correspondence
        b, pts, diffs, last_b = [], [], 0, 0
        abpos, aepos = int(rng.integers(0, 150)), alen - int(rng.integers(0, 150))
        for i in range(abpos, aepos):
            r = rng.random()
            if r < 0.05:
                diffs += 1                      # deletion from B
            elif r < 0.10:
                b += [int(rng.integers(0, 4)), int(a[i])]
                diffs += 1
            elif r < 0.15:
                b.append(int((a[i] + 1 + rng.integers(0, 3)) % 4))
                diffs += 1
            else:
                b.append(int(a[i]))
            if (i + 1) % tspace == 0 or i + 1 == aepos:
                pts += [min(diffs + 2, 60000), len(b) - last_b]
                last_b, diffs = len(b), 0
        pre = rng.integers(0, 4, int(rng.integers(0, 90))).astype(np.int8)
        bseq = np.concatenate([[4], pre, np.array(b, dtype=np.int8), rng.integers(0, 4, 40).astype(np.int8), [4]]).astype(np.int8)
        aseq = np.concatenate([[4], a, [4]]).astype(np.int8)
        blen = len(bseq) - 2
        for mode, kind in ((0, 0), (1, 0), (-1, 0), (0, 1), (1, 1), (-1, 1)):
            res = []
            for which in ("gpu", "oracle"):
                tp = np.array(pts, dtype=np.uint16)
                p = Path(tp.ctypes.data, len(tp), 0, abpos, len(pre), aepos, len(pre) + len(b))
                al = Alignment(ctypes.pointer(p), 0, aseq.ctypes.data + 1, bseq.ctypes.data + 1, alen, blen)
                if which == "gpu":
                    assert (lib.Compute_Trace_MID if kind else lib.Compute_Trace_PTS)(ctypes.byref(al), work, tspace, mode) == 0
                    out = np.ctypeslib.as_array(ctypes.cast(p.trace, ctypes.POINTER(ctypes.c_int)), shape=(max(p.tlen, 1),))[:p.tlen].copy()
                    res.append((out.tolist(), p.diffs))
                else:
                    script = np.zeros(alen + blen + 16, dtype=np.int32)
                    d = ctypes.c_int(0)
                    n = (ora.oracle_compute_trace_mid if kind else ora.oracle_compute_trace_pts)(aseq.ctypes.data + 1, alen, bseq.ctypes.data + 1, blen, ctypes.byref(p),
                                                     tspace, mode, script.ctypes.data, ctypes.byref(d))
                    assert n >= 0
                    res.append((script[:n].tolist(), d.value))
            assert res[0] == res[1], (trial, mode, kind)
    lib.Free_Work_Data(work)


@pytest.mark.parametrize("spacing", [40, 300, 1000])
def test_trace_expansion_other_spacings_against_oracle(built, tmp_path, spacing):
    """Trace spacings the golden files do not have: -s40 (short segments), -s300 and -s1000 (16-bit trace
    points; segments longer than the 240 bases the slot kernel stages, so every one takes the per-lane
    stripe kernel).  Records by the GPU daligner, edit scripts GPU vs oracle (which is pinned to the
    reference), all three modes."""
    from conftest import link_db
    d = str(tmp_path / "w")
    link_db(os.path.join(GOLDEN, "tiny2"), d)
    subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "daligner"), "-k14", "-j4", "-s%d" % spacing, "G.1", "G.2", "G.1"],
                   cwd=d, check=True, stdout=subprocess.DEVNULL)
    n = 0
    for dp, _, fs in os.walk(d):
        for f in fs:
            if not f.endswith(".las"):
                continue
            las = os.path.join(dp, f)
            for mode, mid in ((0, 0), (1, 0), (-1, 0), (0, 1), (1, 1)):
                g, o = str(tmp_path / "g.bin"), str(tmp_path / "o.bin")
                subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "lastrace"), "-m%d" % mode] + (["-M"] if mid else []) +
                               [os.path.join(d, "G"), os.path.join(d, "G"), las, g], check=True)
                subprocess.run([os.path.join(ROOT, "oracle", "oracle_lastrace"), os.path.join(d, "G"), las, o, str(mode)] +
                               (["mid"] if mid else []), check=True)
                assert open(g, "rb").read() == open(o, "rb").read(), (f, mode, mid)
                n += 1
    assert n >= 6

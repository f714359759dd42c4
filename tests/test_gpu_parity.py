"""Parity of the HIP path (through the C-ABI of libdamar_hip.so) with the oracle, the golden
reference outputs and known answers.  Everything here needs a real MI355X."""
import ctypes as C
import hashlib
import os
import random
import sys

import numpy as np
import pytest

from conftest import ROOT, GOLDEN, golden_cases, read_case, link_db, compare_las, opts_to_plan_kwargs

pytestmark = pytest.mark.gpu

NEEDS_BRIDGE = set()


@pytest.fixture(scope="module")
def gpu(built):
    from damar_amd import api
    L = api.lib()
    assert L.damar_hip_init(0) >= 1
    return L


def run_plan(case, workdir):
    from damar_amd import driver
    link_db(case["dbdir"], workdir)
    if case["tool"] == "datander":
        kw = opts_to_plan_kwargs(case["opts"])
        for a, _ in case["lines"]:
            driver.run_datander(driver.Block(os.path.join(workdir, "G." + a)), workdir, **kw)
        return None
    plan = driver.Plan(**opts_to_plan_kwargs(case["opts"]))
    blocks = {}
    for a, bs in case["lines"]:
        for x in [a] + bs:
            if x not in blocks:
                blocks[x] = driver.Block(os.path.join(workdir, "G." + x))
    for a, bs in case["lines"]:
        plan.run_line(blocks[a], [blocks[b] for b in bs], workdir)
    plan.finish()
    return plan


@pytest.mark.parametrize("name", golden_cases())
def test_gpu_las_equals_reference_golden(gpu, tmp_path, name):
    if name in NEEDS_BRIDGE:
        pytest.xfail("Bridge realignment is not built yet (host/bridge.c stops the run loudly)")
    case = read_case(name)
    run_plan(case, str(tmp_path))
    assert compare_las(case, str(tmp_path)) == []


@pytest.mark.parametrize("name", ["tiny2", "tan_tandem", "mask_two", "bias_mask", "wide", "tan_wide"])
def test_gpu_cli_binary_equals_reference_golden(gpu, tmp_path, name):
    """The C host drivers (the drop-in daligner / datander commands).  `wide`: two reads of 2.1 Mb -- 21 000 trace spacings
    at -s100, beyond the 16 000 a packed pebble can name -- and two of 30 kb: the pairs with a long read go through the
    wide kernel (16-byte pebbles, kernels/report.hip report_wide_kernel), the short pair through the two-pair kernel; the
    .las must be the reference's (md5 in tests/golden/wide/las.md5)."""
    from conftest import run_cli
    case = read_case(name)
    run_cli(os.path.join(ROOT, "damar_amd", "bin", "daligner"), case, str(tmp_path))
    assert compare_las(case, str(tmp_path)) == []


@pytest.mark.parametrize("name", ["tiny2", "tandem", "tiny_I", "tiny_k12", "mask_two"])
def test_gpu_cli_merge_general_path_equals_reference_golden(gpu, tmp_path, name):
    """The seed merge has a fast path per tile of A entries (a bucket table over the tile's code range, no search, taken
    when no cap on mutual k-mer matches can apply) and the general path of round 4; every other test takes whichever a
    tile chooses -- here every tile goes the general way (test hook DAMAR_MERGE_GENERAL)."""
    from conftest import run_cli
    case = read_case(name)
    run_cli(os.path.join(ROOT, "damar_amd", "bin", "daligner"), case, str(tmp_path), env=dict(os.environ, DAMAR_MERGE_GENERAL="1"))
    assert compare_las(case, str(tmp_path)) == []


@pytest.mark.parametrize("mode", [{"DAMAR_SORT_PAIR": "1"}, {"DAMAR_SORT_PAIR": "0"}, {"DAMAR_SORT_PAIR": "1", "DAMAR_TEST_RUN_MAX": "6"}])
@pytest.mark.parametrize("name", ["tiny2", "tandem", "tiny_I", "tiny_k12", "mask_two", "bias_mask"])
def test_gpu_cli_seed_sort_over_the_read_pair_equals_reference_golden(gpu, tmp_path, name, mode):
    """The seed pairs are sorted on (bread, aread) only -- 4 radix passes instead of 6 -- and the A-position order inside a read
    pair, which only the report kernel's walk over a run needs (filter.c:2268-2297 adds up apos differences in order), is made
    for the runs of the kept heads alone, where they lie (kernels/seed_merge.hip order_runs; the screen takes a run in any
    order).  DAMAR_SORT_PAIR=0: the sort over all the key bits of rounds 1-6.  DAMAR_TEST_RUN_MAX=6: runs of more than 6 seeds
    count as too long for order_runs, so the comparison is sorted over all the bits after all and its work list made again
    (the path of a run beyond 2048 seeds); the line of DAMAR_PLAN_STATS says how often."""
    import json
    import subprocess
    from damar_amd import api
    case = read_case(name)
    link_db(case["dbdir"], str(tmp_path))
    with open(os.path.join(str(tmp_path), "plan.txt"), "w") as f:
        for a, bs in case["lines"]:
            f.write("daligner %s G.%s %s\n" % (" ".join(case["opts"]), a, " ".join("G." + b for b in bs)))
    st = os.path.join(str(tmp_path), "stats.json")
    subprocess.run([api.daligner_binary(), "-P", "plan.txt"], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL,
                   env=dict(os.environ, DAMAR_PLAN_STATS=st, **mode))
    assert compare_las(case, str(tmp_path)) == []
    resorted = json.load(open(st))["resorted"]
    assert (resorted > 0) == ("DAMAR_TEST_RUN_MAX" in mode)


@pytest.mark.parametrize("name", ["tiny2", "tandem", "tiny_I", "tiny_k12", "mask_two"])
def test_gpu_cli_work_list_in_two_steps_equals_reference_golden(gpu, tmp_path, name):
    """The work list of a comparison -- the heads of the (bread, aread) runs report_thread would enter (filter.c:2212-2215)
    that survive the screen of pass 1's bucket sums (filter.c:2268-2297) -- is made in one pass over the sorted seeds
    (kernels/seed_merge.hip pair_work_mark: every other test); DAMAR_WORK_TWOSTEP=1 is the path of rounds 1-5: the list of
    heads, then one thread per head for the screen, a scan and a compaction."""
    from conftest import run_cli
    case = read_case(name)
    run_cli(os.path.join(ROOT, "damar_amd", "bin", "daligner"), case, str(tmp_path), env=dict(os.environ, DAMAR_WORK_TWOSTEP="1"))
    assert compare_las(case, str(tmp_path)) == []


@pytest.mark.parametrize("cov,genome,opts", [(0.4, 3.0, "-k14"), (12.0, 0.3, "-k14"), (12.0, 0.3, "-k14 -h60 -t8"), (25.0, 0.1, "-k12 -w5")])
def test_gpu_work_list_in_one_pass_has_the_items_of_heads_then_screen(gpu, tmp_path, cov, genome, opts):
    """Both ways of making the work list (test above) over whole plans: the same number of work items -- a screen that kept
    a pair the other drops would only cost time, one that dropped a pair the other keeps loses records -- and the same files.
    Sparse coverage (almost every run dies in the screen), ordinary coverage, a higher -h with a -t cap (longer runs are
    needed, capped k-mers), and a deep small genome with -k12 -w5 (runs beyond the screen's 48 seeds, narrow buckets)."""
    import filecmp
    import json
    import subprocess
    from damar_amd import api
    dirs = [os.path.join(str(tmp_path), x) for x in ("one", "two")]
    os.makedirs(dirs[0])
    os.makedirs(dirs[1])
    nb = api.sim_write_db(dirs[0], "S", genome, coverage=cov, seed=31, block_mbp=1)
    for f in ("S.db", ".S.idx", ".S.bps"):
        os.symlink(os.path.join(dirs[0], f), os.path.join(dirs[1], f))
    items = []
    for d, extra in zip(dirs, ({}, {"DAMAR_WORK_TWOSTEP": "1"})):
        with open(os.path.join(d, "plan.txt"), "w") as f:
            for a in range(1, nb + 1):
                f.write("daligner %s -j4 S.%d %s\n" % (opts, a, " ".join("S.%d" % b for b in range(a, 0, -1))))
        subprocess.run([api.daligner_binary(), "-P", "plan.txt"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       env=dict(os.environ, DAMAR_PLAN_STATS=os.path.join(d, "stats.json"), **extra))
        st = json.load(open(os.path.join(d, "stats.json")))
        items.append((st["work_items"], st["seed_pairs"], st["local_alignments"], st["records"]))
    assert items[0] == items[1] and items[0][0] > 0
    n = 0
    for dp, _, fs in os.walk(dirs[1]):
        for f in fs:
            if f.endswith(".las"):
                rel = os.path.relpath(os.path.join(dp, f), dirs[1])
                assert filecmp.cmp(os.path.join(dp, f), os.path.join(dirs[0], rel), shallow=False), rel
                n += 1
    assert n == nb * nb


@pytest.mark.parametrize("name", ["tan_tandem", "tan_k18"])
def test_gpu_cli_datander_with_blocks_unpacked_on_the_host_equals_reference_golden(gpu, tmp_path, name):
    """The datander command keeps a block as its stretch of the .bps file and lets the GPU unpack it (the default, taken
    by every other datander test); DAMAR_DB_UNPACKED=1 is the path of rounds 1-4: damar_read_block + Match_Self."""
    from conftest import run_cli
    case = read_case(name)
    run_cli(os.path.join(ROOT, "damar_amd", "bin", "daligner"), case, str(tmp_path), env=dict(os.environ, DAMAR_DB_UNPACKED="1"))
    assert compare_las(case, str(tmp_path)) == []


@pytest.mark.parametrize("name", ["tan_tandem", "tan_k18"])
def test_gpu_cli_datander_pebble_overflow_goes_to_the_wide_kernel(gpu, tmp_path, name):
    """scrub/tandem.c:1026 calls Local_Alignment on reads of any length (align.c:505-513 grows its vectors); the two-pair
    kernel's packed pebbles hold 2^18 per direction and 16 000 trace spacings.  What does not fit goes to the wide kernel
    behind the datander launch (kernels/report.hip tandem_wide_kernel, 16-byte pebbles) -- by read length (golden
    `tan_wide` in the tests above: a read of 2.1 Mb) or, here, because the pool ran over at its largest: the test hook
    DAMAR_TEST_MAX_CELLS makes it so small that ordinary self-alignments overflow it.  The files are the reference's."""
    import subprocess
    case = read_case(name)
    link_db(case["dbdir"], str(tmp_path))
    exe = os.path.join(ROOT, "damar_amd", "bin", "datander")
    err = ""
    for a, _ in case["lines"]:
        r = subprocess.run([exe] + case["opts"] + ["G." + a], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, text=True, env=dict(os.environ, DAMAR_TEST_MAX_CELLS="24"))
        err += r.stderr
    assert compare_las(case, str(tmp_path)) == []
    assert "wide kernel" in err


def test_gpu_local_alignment_batch_pebble_overflow_goes_to_the_wide_kernel():
    """The batch entry of Local_Alignment (damar_local_alignment_batch) with a pebble pool so small that most tasks
    overflow it: they are done again by la_batch_kernel<1> behind the two-pair launch, and every path and trace still
    equals the oracle's.  The pool's limit is read once per process: the comparison runs in a child."""
    import subprocess
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-s", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_gpu_local_alignment_batch_equals_oracle"],
                       cwd=ROOT, env=dict(os.environ, DAMAR_TEST_MAX_CELLS="128"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "2 passed" in r.stdout and "wide kernel" in r.stderr


def test_gpu_cli_memory_limit_known_answer(gpu, tmp_path):
    """daligner -M1 on a 25 Mbp block of 250x coverage: the REAL reference (oracle/_ref/daligner
    -v -k14 -M1 -j8 R R in the build container, tests/golden/make_golden.py memlimit) lowers the
    cap on mutual k-mer matches to 210 (N) / 184 (C) and writes this .las."""
    import subprocess
    d = str(tmp_path)
    subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "simdb"), d, "R", "0.1", "-c250", "-r5", "-e.15", "-S200"],
                   check=True, stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "daligner"), "-v", "-k14", "-M1", "-j8", "R", "R"],
                         cwd=d, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240).stdout
    caps = [ln.split()[5] for ln in out.splitlines() if "Capping mutual k-mer matches over" in ln]
    assert caps == ["210", "184"]
    assert "Hit count = 16418362" in out and "Hit count = 13604118" in out          # the reference prints 16,418,362
    las = os.path.join(d, "R.las")
    assert os.path.getsize(las) == 164343834
    assert hashlib.md5(open(las, "rb").read()).hexdigest() == "759225b7cbbae785076eec906a56859f"


def _index_as_records(L, idx, n):
    import oracle_api as O
    buf = np.zeros(n, dtype=O.KMER_DT)
    L.damar_index_download(idx, buf.ctypes.data)
    return buf


@pytest.mark.parametrize("k,t", [(14, 0), (12, 0), (14, 10), (16, 0), (17, 0), (24, 6), (32, 0)])
def test_gpu_kmer_index_equals_oracle(gpu, k, t):
    """K1-K3: the device index in reference KmerPos layout vs oracle_sort_kmers."""
    import oracle_api as O
    from damar_amd import api
    L = gpu
    name = os.path.join(GOLDEN, "tiny2", "G.1")
    db = api.read_block(name)
    assert L.Set_Filter_Params(k, 6, t, 35, 4) == 0
    n = C.c_int(0)
    blk = L.damar_block_upload(C.byref(db))
    idx = L.damar_index_build(blk, 0, C.byref(n))
    got = _index_as_records(L, idx, n.value)
    odb = O.read_block(name)
    p, on, want = O.sort_kmers(odb, O.params(k=k, t=t))
    assert on == n.value and n.value > 0
    assert np.array_equal(got, want)
    O.lib().free(p)
    L.damar_index_free(idx)
    L.damar_block_free(blk)
    L.Set_Filter_Params(14, 6, 0, 35, 4)


@pytest.mark.parametrize("comp", [0, 1])
@pytest.mark.parametrize("cross", [0, 1])
def test_gpu_seed_pairs_equal_oracle(gpu, comp, cross):
    """K4 + seed sort: sorted SeedPair list vs oracle_seed_pairs, self and cross blocks."""
    import oracle_api as O
    from damar_amd import api
    L = gpu
    L.Set_Filter_Params(14, 6, 0, 35, 4)
    L.damar_set_async(0)
    api.set_globals()
    an = os.path.join(GOLDEN, "tiny2", "G.2")
    bn = os.path.join(GOLDEN, "tiny2", "G.1") if cross else an
    adb, bdb = api.read_block(an), api.read_block(bn)
    oadb, obdb = O.read_block(an), O.read_block(bn)
    if comp:
        L.damar_complement_block(C.byref(bdb), 1)
        O.lib().damar_complement_block(C.byref(obdb), 1)
    prm = O.params()
    pa, na, _ = O.sort_kmers(oadb, prm)
    if cross or comp:
        pb, nb, _ = O.sort_kmers(obdb, prm)
    else:
        pb, nb = pa, na
    want = O.seed_pairs(oadb, obdb, pa, na, pb, nb, 0 if cross else 1, comp, prm)

    n = C.c_int(0)
    ablk = L.damar_block_upload(C.byref(adb))
    aidx = L.damar_index_build(ablk, 0, C.byref(n))
    if cross or comp:
        bblk = L.damar_block_upload(C.byref(bdb))
        bidx = L.damar_index_build(bblk, 0, C.byref(n))
    else:
        bblk, bidx = None, aidx
    spec = L.New_Align_Spec(.70, 100, adb.freq, 4, 1, 0, 0, 1)
    L.damar_last_seeds(None, 1)                       # keep the seeds of the next match
    cnt = (api.c_int64 * 3)()
    L.damar_match(C.byref(adb), C.byref(bdb), aidx, bidx, 0 if cross else 1, comp, spec, cnt)
    got = np.zeros(int(cnt[0]), dtype=O.SEED_DT)
    assert L.damar_last_seeds(got.ctypes.data, len(got)) == len(want)
    L.damar_last_seeds(None, 0)
    assert len(got) == len(want) and len(want) > 1000
    assert np.array_equal(got, want)
    if bblk:
        L.damar_index_free(bidx)
        L.damar_block_free(bblk)
    L.damar_index_free(aidx)
    L.damar_block_free(ablk)


@pytest.mark.parametrize("cross", [0, 1])
def test_gpu_memory_limit_lowers_the_cap_like_the_oracle(gpu, cross):
    """filter.c:2634-2699: when the seeds would not fit the host memory limit the cap on mutual
    k-mer matches drops from 10000 to the first count at which they no longer fit.  MEM_LIMIT is
    squeezed until the oracle's rule selects a cap well below 10000; the GPU path must select the
    same cap and keep exactly the same seeds."""
    import oracle_api as O
    from damar_amd import api
    L = gpu
    L.Set_Filter_Params(14, 6, 0, 35, 4)
    L.damar_set_async(0)
    api.set_globals()
    an = os.path.join(GOLDEN, "tandem", "G.1")          # repeat-rich reads: many codes with large counts
    adb, oadb = api.read_block(an), O.read_block(an)
    bdb, obdb = api.read_block(an), O.read_block(an)
    if cross:
        L.damar_complement_block(C.byref(bdb), 1)
        O.lib().damar_complement_block(C.byref(obdb), 1)
    prm = O.params()
    pa, na, _ = O.sort_kmers(oadb, prm)
    pb, nb = (O.sort_kmers(obdb, prm)[:2]) if cross else (pa, na)
    full = len(O.seed_pairs(oadb, obdb, pa, na, pb, nb, 0 if cross else 1, cross, prm))
    dbb = 88 + 32 * (oadb.nreads + 2) + oadb.totlen + oadb.nreads + 4 + len(oadb.path or b"") + 1
    want, mem = None, 0
    for frac in (.9, .8, .7, .6):                       # avail as a fraction of the uncapped seeds
        avail = int(full * frac / .98) + 8
        words = (2 * avail + na) if not cross else (avail + na + nb)     # undo filter.c:2646-2650
        if cross and words > na + 2 * nb:
            words = 2 * avail + na
        mem = 16 * words + 2 * dbb
        prm.mem_limit = mem
        w = O.seed_pairs(oadb, obdb, pa, na, pb, nb, 0 if cross else 1, cross, prm)
        if 2 < O.LAST_LIMIT < 5000 and 0 < len(w) < full:
            want = w
            break
    assert want is not None, "no memory limit found that lowers the cap on this fixture"
    lim = O.LAST_LIMIT
    memvar = C.c_uint64.in_dll(L, "MEM_LIMIT")
    old = memvar.value
    try:
        memvar.value = mem
        n = C.c_int(0)
        ablk = L.damar_block_upload(C.byref(adb))
        aidx = L.damar_index_build(ablk, 0, C.byref(n))
        if cross:
            bblk = L.damar_block_upload(C.byref(bdb))
            bidx = L.damar_index_build(bblk, 0, C.byref(n))
        else:
            bblk, bidx = None, aidx
        spec = L.New_Align_Spec(.70, 100, adb.freq, 4, 1, 0, 0, 1)
        L.damar_last_seeds(None, 1)
        cnt = (api.c_int64 * 3)()
        L.damar_match(C.byref(adb), C.byref(bdb), aidx, bidx, 0 if cross else 1, cross, spec, cnt)
        got = np.zeros(int(cnt[0]), dtype=O.SEED_DT)
        assert L.damar_last_seeds(got.ctypes.data, len(got)) == len(want)
        L.damar_last_seeds(None, 0)
        assert L.damar_last_limit() == lim
        assert np.array_equal(got, want)
    finally:
        memvar.value = old
    if bblk:
        L.damar_index_free(bidx)
        L.damar_block_free(bblk)
    L.damar_index_free(aidx)
    L.damar_block_free(ablk)


@pytest.mark.parametrize("comp", [0, 1])
def test_gpu_local_alignment_batch_equals_oracle(gpu, comp):
    """K6 alone: thousands of (read pair, diagonal, anti-diagonal) seeds -- including seeds the
    band filter would never fire -- through the wave kernel vs oracle_local_alignment."""
    import oracle_api as O
    from damar_amd import api
    L = gpu
    L.Set_Filter_Params(14, 6, 0, 35, 4)
    an, bn = os.path.join(GOLDEN, "indel", "G.1"), os.path.join(GOLDEN, "indel", "G.1")
    adb, bdb = api.read_block(an), api.read_block(bn)
    oadb, obdb = O.read_block(an), O.read_block(bn)
    if comp:
        L.damar_complement_block(C.byref(bdb), 1)
        O.lib().damar_complement_block(C.byref(obdb), 1)
    prm = O.params()
    pa, na, _ = O.sort_kmers(oadb, prm)
    pb, nb, _ = O.sort_kmers(obdb, prm)
    seeds = O.seed_pairs(oadb, obdb, pa, na, pb, nb, 0, comp, prm)
    rng = random.Random(5 + comp)
    pick = sorted(rng.sample(range(len(seeds)), 1500))
    tasks = []
    for i in pick:
        s = seeds[i]
        if s["aread"] == s["bread"] and not comp:
            continue                                   # selfie seeds need aseq == bseq pointers
        tasks += [int(s["aread"]), int(s["bread"]), int(s["diag"]), int(2 * s["apos"] - s["diag"])]
    nt = len(tasks) // 4
    ospec = O.lib().New_Align_Spec(.70, 100, oadb.freq, 1, 1, 0, 0, 1)
    spec = L.New_Align_Spec(.70, 100, adb.freq, 1, 1, 0, 0, 1)
    ablk, bblk = L.damar_block_upload(C.byref(adb)), L.damar_block_upload(C.byref(bdb))
    paths = (C.c_int * (12 * nt))()
    toff = (api.c_int64 * (2 * nt))()
    cap = nt * 1200
    traces = (C.c_uint16 * cap)()
    assert L.damar_local_alignment_batch(ablk, bblk, comp, spec, (C.c_int * len(tasks))(*tasks), nt,
                                         paths, toff, traces, cap) == 0
    maxtp = 4 * (max(adb.maxlen, bdb.maxlen) // 100 + 4)
    bad = 0
    for t in range(nt):
        ar, br, dg, anti = tasks[4 * t:4 * t + 4]
        want, wat, wbt = O.local_alignment(oadb, obdb, ar, br, comp, dg, anti, ospec, maxtp)
        got = list(paths[12 * t:12 * t + 12])
        gat = list(traces[toff[2 * t]:toff[2 * t] + got[5]])
        gbt = list(traces[toff[2 * t + 1]:toff[2 * t + 1] + got[11]])
        if got != want or gat != wat or gbt != wbt:
            bad += 1
    assert nt > (1000 if comp else 200) and bad == 0
    L.damar_block_free(ablk)
    L.damar_block_free(bblk)


def test_gpu_local_alignment_c_abi_single_call(gpu):
    """align.h's own entry points (New_Work_Data / Local_Alignment / Free_Work_Data) for the call
    shape of filter.c:2316: A-view path through align->path, B-view path returned, vs the oracle."""
    import oracle_api as O
    from damar_amd import api
    L = gpu

    class Alignment(C.Structure):
        _fields_ = [("path", C.POINTER(O.Path)), ("flags", C.c_uint32), ("aseq", C.c_void_p), ("bseq", C.c_void_p),
                    ("alen", C.c_int), ("blen", C.c_int)]
    L.New_Work_Data.restype = C.c_void_p
    L.Free_Work_Data.argtypes = [C.c_void_p]
    L.Local_Alignment.restype = C.POINTER(O.Path)
    L.Local_Alignment.argtypes = [C.POINTER(Alignment), C.c_void_p, C.c_void_p] + [C.c_int] * 5
    L.Set_Filter_Params(14, 6, 0, 35, 4)
    an = os.path.join(GOLDEN, "indel", "G.1")
    adb, oadb = api.read_block(an), O.read_block(an)
    prm = O.params()
    pa, na, _ = O.sort_kmers(oadb, prm)
    seeds = O.seed_pairs(oadb, oadb, pa, na, pa, na, 1, 0, prm)
    rng = random.Random(11)
    ospec = O.lib().New_Align_Spec(.70, 100, oadb.freq, 1, 1, 0, 0, 1)
    spec = L.New_Align_Spec(.70, 100, adb.freq, 1, 1, 0, 0, 1)
    work = L.New_Work_Data()
    maxtp = 4 * (adb.maxlen // 100 + 4)
    done = 0
    for i in rng.sample(range(len(seeds)), 40):
        s = seeds[i]
        ar, br, dg = int(s["aread"]), int(s["bread"]), int(s["diag"])
        if ar == br:
            continue
        anti = int(2 * s["apos"] - s["diag"])
        ap = O.Path()
        al = Alignment()
        al.path = C.pointer(ap)
        al.flags = 0
        al.aseq = adb.bases + adb.reads[ar].boff
        al.bseq = adb.bases + adb.reads[br].boff
        al.alen, al.blen = adb.reads[ar].rlen, adb.reads[br].rlen
        bp = L.Local_Alignment(C.byref(al), work, spec, dg, dg, anti, -1, -1).contents
        want, wat, wbt = O.local_alignment(oadb, oadb, ar, br, 0, dg, anti, ospec, maxtp)
        got = [ap.abpos, ap.bbpos, ap.aepos, ap.bepos, ap.diffs, ap.tlen,
               bp.abpos, bp.bbpos, bp.aepos, bp.bepos, bp.diffs, bp.tlen]
        assert got == want
        assert list(C.cast(ap.trace, C.POINTER(C.c_uint16))[:ap.tlen]) == wat
        assert list(C.cast(bp.trace, C.POINTER(C.c_uint16))[:bp.tlen]) == wbt
        done += 1
    assert done >= 20
    L.Free_Work_Data(work)


def test_gpu_config1_known_answer(gpu, tmp_path):
    """BASELINE config 1 (simulator 0.5 -c20 -r1 -e.15, 930 reads, 10 Mbp): the reference's
    .las md5 08bcb3ac... (reproduced in this repo's build container)."""
    from damar_amd import api, driver
    d = str(tmp_path)
    assert api.sim_write_db(d, "SIM", 0.5, coverage=20., seed=1, block_mbp=200) == 1
    blk = driver.Block(os.path.join(d, "SIM.1"))
    plan = driver.Plan(j=4)
    plan.run_line(blk, [blk], d)
    plan.finish()
    las = os.path.join(d, "d001_00001", "SIM.1.SIM.1.las")
    assert os.path.getsize(las) == 4982214
    assert hashlib.md5(open(las, "rb").read()).hexdigest() == "08bcb3acacfb24a16fb3d18daeab5ff0"
    assert plan.counts == [1475304 + 1473761, 8314 + 8377, 15632 + 15630]
    n, bp = driver.las_stats(las)
    assert (n, bp) == (31262, 184279123)


def _check_las_invariants(path, tspace=100):
    """LAcheck-style properties (utils/LAcheck.c -p -s): sorted by the 7 keys, trace sums
    consistent with the B interval, diffs = sum of per-segment diffs."""
    raw = open(path, "rb").read()
    novl = int(np.frombuffer(raw, "<i8", 1)[0])
    off, prev = 12, None
    for _ in range(novl):
        r = np.frombuffer(raw, "<i4", 10, off)
        tlen, diffs, ab, bb, ae, be, flags, ar, br = [int(x) for x in r[:9]]
        tr = np.frombuffer(raw, np.uint8, tlen, off + 40)
        off += 40 + tlen
        assert tlen % 2 == 0 and ae > ab and be > bb
        assert int(tr[1::2].sum()) == be - bb
        assert int(tr[0::2].sum()) == diffs
        assert tlen // 2 == (ae - 1) // tspace - ab // tspace + 1 or tlen // 2 == ae // tspace - ab // tspace + (ae % tspace != 0)
        key = (ar, br, flags & 1, ab, ae, bb, be)
        assert prev is None or prev <= key
        prev = key
    assert off == len(raw)
    return novl


def test_gpu_config2_full_plan_known_answer(gpu, tmp_path):
    """BASELINE config 2 at full size (540 Mbp, 4 blocks of 135 Mbp, all 10 block pairs, 16 .las
    files): every file against the md5 of what the reference daligner wrote (tests/golden/
    config2_ref_md5.txt), the LAcheck-style invariants on three of them, and the aligned-bp total
    that bench.py's metric is built from."""
    from damar_amd import api, driver
    d = str(tmp_path)
    assert api.sim_write_db(d, "SIM", 27., coverage=20., seed=2, block_mbp=135) == 4
    want = {}
    for ln in open(os.path.join(GOLDEN, "config2_ref_md5.txt")):
        if ln.startswith("#"):
            continue
        m, f = ln.split()
        want[f] = m
    assert len(want) == 16
    blocks = {i: driver.Block(os.path.join(d, "SIM.%d" % i)) for i in range(1, 5)}
    plan = driver.Plan(j=16)
    for a, bs in driver.hpc_plan(4):
        plan.run_line(blocks[a], [blocks[b] for b in bs], d)
    plan.finish()
    assert plan.index_builds == 8
    nrec = bp = 0
    for f, m in sorted(want.items()):
        assert hashlib.md5(open(os.path.join(d, f), "rb").read()).hexdigest() == m, f
        n, b = driver.las_stats(os.path.join(d, f))
        nrec += n
        bp += b
    assert (nrec, bp) == (1676758, 9893531852)
    for f in ("d001_00002/SIM.2.SIM.2.las", "d001_00002/SIM.2.SIM.1.las", "d001_00001/SIM.1.SIM.2.las"):
        assert _check_las_invariants(os.path.join(d, f)) > 50000


@pytest.mark.parametrize("name", ["tiny2", "mask_two"])
def test_reference_driver_linked_against_hip_library(gpu, tmp_path, name):
    """Drop-in proof: the reference's OWN dalign/daligner.c, compiled from /root/reference and
    linked against libdamar_hip.so instead of filter.c + align.c (oracle/Makefile.ref target
    `dropin`), writes the golden .las files."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "daligner_on_damar")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/daligner_on_damar not built (needs /root/reference at build time)")
    case = read_case(name)          # mask_two: the reference's own read_DB loads and merges the tracks
    link_db(case["dbdir"], str(tmp_path))
    for a, bs in case["lines"]:
        subprocess.run([exe] + case["opts"] + ["G." + a] + ["G." + b for b in bs], cwd=str(tmp_path), check=True,
                       stdout=subprocess.DEVNULL)
    assert compare_las(case, str(tmp_path)) == []


@pytest.mark.parametrize("name", ["tan_tandem", "tan_k18", "tan_plain", "tan_wide"])
def test_reference_datander_linked_against_tandem_library(gpu, tmp_path, name):
    """Drop-in proof of the SECOND boundary (scrub/tandem.h:54-60): the reference's OWN scrub/datander.c, compiled from
    /root/reference and linked on libdamar_tandem.so + libdamar_hip.so instead of scrub/tandem.c + dalign/align.c
    (oracle/Makefile.ref target `dropin`), writes the golden tan/*.las files: its main calls
    the 4-argument Set_Filter_Params, New_Align_Spec, Match_Self, Write_Overlap_Buffer, Reset_Overlap_Buffer and
    Free_Align_Spec of the libraries (scrub/datander.c:226-258)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "datander_on_damar")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/datander_on_damar not built (needs /root/reference at build time)")
    ldd = subprocess.run(["ldd", exe], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "libdamar_tandem.so" in ldd and "libdamar_hip.so" in ldd
    case = read_case(name)
    assert case["tool"] == "datander"
    link_db(case["dbdir"], str(tmp_path))
    for a, _ in case["lines"]:
        subprocess.run([exe] + case["opts"] + ["G." + a], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL)
    assert compare_las(case, str(tmp_path)) == []


def test_gpu_sparse_coverage_equals_oracle(gpu, tmp_path):
    """Hardly any true overlaps (0.4x coverage): almost every read pair dies in the screen or in the
    band filter, some block pairs produce no record at all; the .las files (header-only ones
    included) must equal the oracle's."""
    import subprocess
    from damar_amd import api, driver
    d = str(tmp_path)
    nb = api.sim_write_db(d, "S", 3.0, coverage=.4, seed=9, block_mbp=1)
    assert nb == 2
    blocks = {i: driver.Block(os.path.join(d, "S.%d" % i)) for i in (1, 2)}
    plan = driver.Plan(j=4)
    out = os.path.join(d, "gpu")
    for a, bs in driver.hpc_plan(2):
        plan.run_line(blocks[a], [blocks[b] for b in bs], out)
    plan.finish()
    orc = os.path.join(d, "orc")
    os.makedirs(orc)
    for f in ("S.db", ".S.idx", ".S.bps"):
        os.symlink(os.path.join(d, f), os.path.join(orc, f))
    for a, bs in driver.hpc_plan(2):
        subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner"), "-k14", "-j4", "S.%d" % a] + ["S.%d" % b for b in bs],
                       cwd=orc, check=True, stdout=subprocess.DEVNULL)
    n = 0
    for dp, _, fs in os.walk(orc):
        for f in fs:
            if f.endswith(".las"):
                rel = os.path.relpath(os.path.join(dp, f), orc)
                assert open(os.path.join(dp, f), "rb").read() == open(os.path.join(out, rel), "rb").read(), rel
                n += 1
    assert n == 4


def test_gpu_random_option_combinations_equal_oracle(gpu, tmp_path):
    """Differential sweep of the option space on a small two-block database: every combination is
    run through the in-process driver on the GPU and through oracle_daligner on the CPU (the oracle
    itself is pinned on each option family by the reference goldens); all .las must be identical."""
    import itertools
    import subprocess
    from damar_amd import driver
    rng = random.Random(2024)
    dbdir = os.path.join(GOLDEN, "mask_dust")
    combos = []
    for _ in range(10):
        o = dict(k=rng.choice([10, 12, 14, 16, 18, 21]), w=rng.choice([4, 5, 6, 7]), h=rng.choice([25, 35, 50]),
                 e=rng.choice([.65, .7, .8]), l=rng.choice([500, 1000, 2000]), s=rng.choice([50, 100, 126, 200]),
                 t=rng.choice([0, 0, 8, 20]), j=rng.choice([1, 2, 4, 8]))
        o["identity"] = rng.choice([0, 1])
        o["symmetric"] = rng.choice([1, 1, 0])
        o["masks"] = rng.choice([[], ["dust"], ["dust", "rnd"]])
        o["biased"] = rng.choice([0, 0, 1])
        combos.append(o)
    combos[0]["k"], combos[1]["k"] = 20, 27            # wide (64-bit) codes in any case
    for n, o in enumerate(combos):
        gdir, odir = os.path.join(str(tmp_path), "g%d" % n), os.path.join(str(tmp_path), "o%d" % n)
        link_db(dbdir, gdir)
        link_db(dbdir, odir)
        blocks = {i: driver.Block(os.path.join(gdir, "G.%d" % i)) for i in (1, 2)}
        plan = driver.Plan(**o)
        for a, bs in driver.hpc_plan(2):
            plan.run_line(blocks[a], [blocks[b] for b in bs], gdir)
        plan.finish()
        opts = ["-k%d" % o["k"], "-w%d" % o["w"], "-h%d" % o["h"], "-e%g" % o["e"], "-l%d" % o["l"], "-s%d" % o["s"],
                "-j%d" % o["j"]] + (["-t%d" % o["t"]] if o["t"] else []) + (["-I"] if o["identity"] else []) + \
               ([] if o["symmetric"] else ["-A"]) + ["-m" + m for m in o["masks"]] + (["-b"] if o["biased"] else [])
        for a, bs in driver.hpc_plan(2):
            subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner")] + opts + ["G.%d" % a] + ["G.%d" % b for b in bs],
                           cwd=odir, check=True, stdout=subprocess.DEVNULL)
        nlas = 0
        for dp, _, fs in os.walk(odir):
            for f in fs:
                if f.endswith(".las"):
                    rel = os.path.relpath(os.path.join(dp, f), odir)
                    assert open(os.path.join(dp, f), "rb").read() == open(os.path.join(gdir, rel), "rb").read(), (opts, rel)
                    nlas += 1
        assert nlas >= 3, opts


@pytest.mark.parametrize("name", ["tandem", "fusion", "noisy"])
def test_gpu_threaded_host_tail_with_redundancies(gpu, tmp_path, name):
    """The host tail splits the read pairs of one launch over several threads; force that path
    (DAMAR_TAIL_MIN=1, 7 threads) on the cases that exercise Handle_Redundancies, Fusion and the
    Bridge realignment (per-thread work buffers), through the CLI binary."""
    import subprocess
    case = read_case(name)
    link_db(case["dbdir"], str(tmp_path))
    env = dict(os.environ, DAMAR_TAIL_MIN="1", DAMAR_TAIL_THREADS="7")
    for a, bs in case["lines"]:
        subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "daligner")] + case["opts"] + ["G." + a] + ["G." + b for b in bs],
                       cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL, env=env)
    assert compare_las(case, str(tmp_path)) == []


def test_gpu_cli_plan_line_with_several_b_blocks_equals_reference(gpu, tmp_path):
    """One plan line against three B blocks through the C driver, with the host tail slowed to a
    single thread so that it lags behind the GPU: the driver reuses its block records for the next
    B block while tails are still pending (they must work on their own copies).  Checked against
    the real reference binary where it exists (GPU box and build container)."""
    import subprocess
    ref = os.path.join(ROOT, "oracle", "_ref", "daligner")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/daligner not built")
    d = str(tmp_path)
    subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "simdb"), d, "S", "3.0", "-c20", "-r13", "-e.15", "-S15"],
                   check=True, stdout=subprocess.DEVNULL)
    line = ["S.4", "S.4", "S.3", "S.2", "S.1"]
    for sub, exe, env in (("ref", ref, None),
                          ("gpu", os.path.join(ROOT, "damar_amd", "bin", "daligner"), dict(os.environ, DAMAR_TAIL_THREADS="1"))):
        w = os.path.join(d, sub)
        os.makedirs(w)
        for f in ("S.db", ".S.idx", ".S.bps"):
            os.symlink(os.path.join(d, f), os.path.join(w, f))
        subprocess.run([exe, "-k14", "-j8"] + line, cwd=w, check=True, stdout=subprocess.DEVNULL, env=env, timeout=300)
    n = 0
    for dp, _, fs in os.walk(os.path.join(d, "ref")):
        for f in fs:
            if f.endswith(".las"):
                rel = os.path.relpath(os.path.join(dp, f), os.path.join(d, "ref"))
                assert open(os.path.join(dp, f), "rb").read() == open(os.path.join(d, "gpu", rel), "rb").read(), rel
                n += 1
    assert n == 7


def test_gpu_multi_module_one_rank_rccl(gpu, tmp_path):
    """python -m damar_amd.multi under torchrun with one rank on the GPU (RCCL process group, plan,
    LAmerge step): per-pair files equal the golden ones, merged block files equal the reference
    LAmerge's."""
    import subprocess
    import sys
    case = read_case("tiny2")
    w = str(tmp_path)
    link_db(case["dbdir"], w)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29578", "-m", "damar_amd.multi",
                        os.path.join(w, "G"), "2", w], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert compare_las(case, w) == []
    want = {ln.split()[2]: ln.split()[0] for ln in open(os.path.join(GOLDEN, "lamerge_ref_md5.txt"))
            if ln.split()[1] == "tiny2" and ln.split()[3] == "-"}
    for b in (1, 2):
        assert hashlib.md5(open(os.path.join(w, "G.%d.las" % b), "rb").read()).hexdigest() == want["d001_%05d" % b]


@pytest.mark.parametrize("name", ["tiny2", "tandem"])
def test_gpu_buffer_overflow_relaunch(gpu, tmp_path, name):
    """The report kernel never writes past its record, trace or pebble buffers: it raises a flag
    and the host launches it again with larger ones.  Start from absurdly small buffers
    (DAMAR_TEST_SMALL_CAPS) and require the usual golden output."""
    import subprocess
    case = read_case(name)
    link_db(case["dbdir"], str(tmp_path))
    env = dict(os.environ, DAMAR_TEST_SMALL_CAPS="1")
    for a, bs in case["lines"]:
        r = subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "daligner"), "-v"] + case["opts"] + ["G." + a] + ["G." + b for b in bs],
                           cwd=str(tmp_path), check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        assert "retrying with larger buffers" in r.stdout
    assert compare_las(case, str(tmp_path)) == []


def _write_db(directory, root, reads):
    """A one-block DB (.db stub, .idx, .bps in the layout of db/DB.h) from Python sequences over 0..3."""
    import struct
    from damar_amd import api
    os.makedirs(directory, exist_ok=True)
    tot = sum(len(r) for r in reads)
    cnt = [sum(r.count(c) for r in reads) for c in range(4)]
    hdr = api.HITS_DB()
    hdr.ureads = len(reads)
    for c in range(4):
        hdr.freq[c] = cnt[c] / tot
    hdr.maxlen = max(len(r) for r in reads)
    hdr.totlen = tot
    with open(os.path.join(directory, ".%s.idx" % root), "wb") as idx, open(os.path.join(directory, ".%s.bps" % root), "wb") as bps:
        idx.write(bytes(hdr))
        off = 0
        for r in reads:
            rec = api.HITS_READ()
            rec.rlen, rec.boff, rec.coff, rec.flags = len(r), off, -1, 0x800
            idx.write(bytes(rec))
            packed = bytearray((len(r) + 3) // 4)
            for i, c in enumerate(r):
                packed[i >> 2] |= c << (6 - 2 * (i & 3))
            bps.write(bytes(packed))
            off += len(packed)
    with open(os.path.join(directory, "%s.db" % root), "w") as f:
        f.write("files = %9d\n  %9d %s %s\n" % (1, len(reads), "syn", "Syn"))
        f.write("blocks = %9d\nsize = %9d\n %9d\n %9d\n" % (1, 200, 0, len(reads)))


def test_gpu_band_ring_overflow_relaunch(gpu, tmp_path):
    """Bands wider than the wavefront live in per-slot rings of diagonals (4096 by default); a band that
    outgrows its ring raises a flag and the launch is repeated with larger rings.  Reads made of noisy
    short-period tandem arrays at -e.55 keep up to 240 diagonals alive: with rings of 128 diagonals (DAMAR_RING)
    the relaunch must happen and the .las must equal the oracle's, as it must with the default rings."""
    import subprocess
    rng = random.Random(1)
    unit = [rng.randrange(4) for _ in range(3)]
    core = [rng.randrange(4) for _ in range(900)] + unit * 1500 + [rng.randrange(4) for _ in range(900)]

    def noisy(seq, rate):
        out = []
        for c in seq:
            x = rng.random()
            if x < rate / 3:
                continue
            if x < 2 * rate / 3:
                out += [rng.randrange(4), c]
            elif x < rate:
                out.append((c + 1 + rng.randrange(3)) % 4)
            else:
                out.append(c)
        return out
    reads = [noisy(core, .15) for _ in range(8)]       # the oracle reports bands up to 240 diagonals at -e.55
    odir = str(tmp_path / "o")
    _write_db(odir, "G", reads)
    opts = ["-k14", "-j4", "-l800", "-e.55"]
    subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner")] + opts + ["G.1", "G.1"], cwd=odir, check=True, stdout=subprocess.DEVNULL)
    want = open(os.path.join(odir, "d001_00001", "G.1.G.1.las"), "rb").read()
    assert len(want) > 10000
    for ring, expect in (("128", True), ("4096", False)):
        gdir = str(tmp_path / ("g" + ring))
        _write_db(gdir, "G", reads)
        r = subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "daligner"), "-v"] + opts + ["G.1", "G.1"], cwd=gdir, check=True,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, DAMAR_RING=ring))
        assert open(os.path.join(gdir, "d001_00001", "G.1.G.1.las"), "rb").read() == want, ring
        assert ("retrying with larger buffers" in r.stdout) == expect, (ring, r.stdout[-600:])


def test_gpu_successive_jobs_with_different_correlation(gpu, tmp_path):
    """Several jobs in one process with different -e (so different SCORE/TABLE): every job must use
    its own tables even when the allocator hands a new Align_Spec the address of a freed one."""
    import subprocess
    from damar_amd import driver
    dbdir = os.path.join(GOLDEN, "noisy")
    for n, e in enumerate([.8, .65, .75, .65, .8]):
        gdir, odir = os.path.join(str(tmp_path), "g%d" % n), os.path.join(str(tmp_path), "o%d" % n)
        link_db(dbdir, gdir)
        link_db(dbdir, odir)
        blk = driver.Block(os.path.join(gdir, "G.1"))
        plan = driver.Plan(e=e, l=500)
        plan.run_line(blk, [blk], gdir)
        plan.finish()
        subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner"), "-k14", "-j4", "-e%g" % e, "-l500", "G.1", "G.1"],
                       cwd=odir, check=True, stdout=subprocess.DEVNULL)
        rel = os.path.join("d001_00001", "G.1.G.1.las")
        assert open(os.path.join(odir, rel), "rb").read() == open(os.path.join(gdir, rel), "rb").read(), e


def test_gpu_pipeline_from_fasta_with_the_repository_tools(gpu, tmp_path):
    """The whole tool chain a user runs, with this repository's binaries only: the reads of the tiny2 fixture as
    FASTA -> bin/FA2db -> bin/DBsplit -s1 -> bin/daligner (HPCdaligner plan of 2 blocks) -> bin/LAmerge ->
    bin/lastrace.  The database must equal the fixture's (so every .las equals the reference's golden files),
    the merged block files the reference LAmerge's, the edit scripts the reference Compute_Trace_PTS's."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(GOLDEN))
    import make_golden as MG
    d = str(tmp_path)
    reads = MG.unpack_reads(os.path.join(GOLDEN, "tiny2"), "G")
    MG.write_fasta(os.path.join(d, "sim.fasta"), reads)
    tools = os.path.join(ROOT, "damar_amd", "bin")
    subprocess.run([os.path.join(tools, "FA2db"), "G", "sim.fasta"], cwd=d, check=True)
    subprocess.run([os.path.join(tools, "DBsplit"), "-s1", "G"], cwd=d, check=True)
    assert open(os.path.join(d, ".G.bps"), "rb").read() == open(os.path.join(GOLDEN, "tiny2", ".G.bps"), "rb").read()
    assert open(os.path.join(d, "G.db")).read() == open(os.path.join(GOLDEN, "tiny2", "G.db")).read()
    case = read_case("tiny2")
    for a, bs in case["lines"]:
        subprocess.run([os.path.join(tools, "daligner")] + case["opts"] + ["G." + a] + ["G." + b for b in bs], cwd=d, check=True,
                       stdout=subprocess.DEVNULL)
    assert compare_las(case, d) == []
    want = {ln.split()[2]: ln.split()[0] for ln in open(os.path.join(GOLDEN, "lamerge_ref_md5.txt"))
            if ln.split()[1] == "tiny2" and ln.split()[3] == "-"}
    for b in (1, 2):
        subprocess.run([os.path.join(tools, "LAmerge"), "-n", "8", "G", "G.%d.las" % b, "d001_%05d" % b], cwd=d, check=True,
                       stdout=subprocess.DEVNULL)
        assert hashlib.md5(open(os.path.join(d, "G.%d.las" % b), "rb").read()).hexdigest() == want["d001_%05d" % b]
    ref = {(n, l, m): h for h, n, l, m in (ln.split() for ln in open(os.path.join(GOLDEN, "trace_ref_md5.txt")))}
    las = case["las"][0]
    subprocess.run([os.path.join(tools, "lastrace"), os.path.join(d, "G"), os.path.join(d, "G"), os.path.join(d, las),
                    os.path.join(d, "t.bin")], check=True)
    assert hashlib.md5(open(os.path.join(d, "t.bin"), "rb").read()).hexdigest() == ref[("tiny2", las, "0")]

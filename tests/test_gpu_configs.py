"""BASELINE.json configs 3, 4 and 5 on the GPU, against md5s of what the REAL reference wrote for the
same inputs (tests/golden/config{3,4,5}_ref_md5.txt, made in the build container by
tests/golden/make_golden.py from oracle/_ref; the inputs are regenerated here from their seeds by
the repository's own simulator restatement, which is byte-identical to `simulator | FA2db | DBsplit`).

  config 3: simulator 4.6 -c87 -r3, DBsplit -s25 -> 17 blocks, all 153 block pairs, 289 .las files.
  config 4: simulator 248 -c80 -m15000 -s3000 -r4, DBsplit -s78 -> 255 blocks of 78 Mbp (19.8 Gbp).
            A seeded random sample of 8 block pairs (2 self, 6 cross) among the first 24 blocks, which
            `simdb -N24` reproduces in 15 s instead of 2.7 min for the whole database (the generator is
            sequential: they are the same blocks); with DAMAR_C4_FULL=1 also 8 pairs drawn from the
            whole 255 x 255 triangle (scripts/gpu_c4_full.sh runs that once per round).
  config 5: datander on blocks 1 and 4 of the config-2 database, plain (no tandem seeds) and with tandem
            arrays implanted into 30 % of the reads.
"""
import hashlib
import os

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


@pytest.fixture(scope="module")
def gpu(built):
    from damar_amd import api
    L = api.lib()
    if L.damar_hip_init(0) < 1:
        pytest.fail("no HIP device: the product path has no CPU fallback")
    return L


def test_gpu_config3_all_block_pairs_equal_reference(gpu, tmp_path):
    """dalign/daligner.c:958-1056 over the whole HPCdaligner plan of config 3 (17 lines, 153 block pairs)."""
    from damar_amd import api, driver
    d = str(tmp_path)
    nb = api.sim_write_db(d, "SIM", 4.6, coverage=87., seed=3, block_mbp=25)
    assert nb == 17
    want = {}
    for ln in open(os.path.join(GOLDEN, "config3_ref_md5.txt")):
        m, f = ln.split()
        want[f] = m
    assert len(want) == 17 * 17
    blocks = {i: driver.Block(os.path.join(d, "SIM.%d" % i)) for i in range(1, nb + 1)}
    plan = driver.Plan(j=16)
    for a, bs in driver.hpc_plan(nb):
        plan.run_line(blocks[a], [blocks[b] for b in bs], d)
    plan.finish()
    assert plan.index_builds == 2 * nb
    bad = [f for f, m in sorted(want.items()) if _md5(os.path.join(d, f)) != m]
    assert not bad, bad[:5]
    for b in blocks.values():
        b.close()


def _c4_want(sample):
    pairs, files = [], {}
    for ln in open(os.path.join(GOLDEN, "config4_ref_md5.txt")):
        if ln.startswith("#"):
            continue
        m, tag, a, b, rel = ln.split()
        if tag == sample:
            if (int(a), int(b)) not in pairs:
                pairs.append((int(a), int(b)))
            files[rel] = m
    return pairs, files


def _run_c4_sample(d, pairs, files):
    from damar_amd import driver
    cache = {}
    plan = driver.Plan(j=8, index_cache_bytes=40 << 30)
    for a, b in pairs:
        for x in (a, b):
            if x not in cache:
                cache[x] = driver.Block(os.path.join(d, "SIM.%d" % x))
        plan.run_line(cache[a], [cache[b]], d)
    plan.finish()
    bad = [f for f, m in sorted(files.items()) if _md5(os.path.join(d, f)) != m]
    assert not bad, bad
    for blk in cache.values():
        blk.close()
    return plan


def test_gpu_config4_sample_of_8_block_pairs_among_first_24_blocks_equal_reference(gpu, tmp_path):
    from damar_amd import api
    d = str(tmp_path)
    nb = api.sim_write_db(d, "SIM", 248., coverage=80., seed=4, rmean=15000, rsdev=3000, block_mbp=78, max_blocks=24)
    assert nb == 24
    pairs, files = _c4_want("lead")
    assert len(pairs) == 8 and sum(a == b for a, b in pairs) == 2 and len(files) == 14
    plan = _run_c4_sample(d, pairs, files)
    assert plan.counts[2] > 10000          # (0.3x coverage per block: few overlaps per block pair)


@pytest.mark.skipif(bool(os.environ.get("DAMAR_C4_SKIP")), reason="DAMAR_C4_SKIP set (the whole 19.8 Gbp database: about 100 s)")
def test_gpu_config4_sample_of_8_block_pairs_of_all_255_blocks_equal_reference(gpu):
    import shutil
    import tempfile
    from damar_amd import api
    d = tempfile.mkdtemp(prefix="damar_c4_", dir="/dev/shm")
    try:
        nb = api.sim_write_db(d, "SIM", 248., coverage=80., seed=4, rmean=15000, rsdev=3000, block_mbp=78)
        assert nb == 255
        pairs, files = _c4_want("full")
        assert len(pairs) == 8
        _run_c4_sample(d, pairs, files)
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("variant", ["plain", "tandem"])
def test_gpu_config5_datander_on_config2_blocks_equals_reference(gpu, tmp_path, variant):
    """scrub/datander.c:226-258 on 135 Mbp blocks; the tandem variant gives Match_Self real work."""
    from damar_amd import api, driver
    d = str(tmp_path)
    assert api.sim_write_db(d, "SIM", 27., coverage=20., seed=2, block_mbp=135,
                            tandem_frac=.3 if variant == "tandem" else 0.) == 4
    want = {}
    for ln in open(os.path.join(GOLDEN, "config5_ref_md5.txt")):
        if ln.startswith("#"):
            continue
        m, tag, blk, novl = ln.split()
        if tag == variant:
            want[int(blk)] = (m, int(novl))
    assert sorted(want) == [1, 2, 3, 4]
    for blk, (m, novl) in sorted(want.items()):
        if blk in (2, 3):
            continue                                   # (through the command below)
        b = driver.Block(os.path.join(d, "SIM.%d" % blk))
        driver.run_datander(b, d, j=8)
        las = os.path.join(d, "tan", "SIM.%d.SIM.%d.las" % (blk, blk))
        n, _ = driver.las_stats(las)
        assert n == novl
        assert _md5(las) == m
        if variant == "tandem":
            assert n > 1000
        b.close()
    # the command over all four blocks (one process: a reader thread ahead of the GPU, host/datander.c)
    import subprocess
    cli = os.path.join(d, "cli")
    os.makedirs(cli)
    for f in ("SIM.db", ".SIM.idx", ".SIM.bps"):
        os.symlink(os.path.join(d, f), os.path.join(cli, f))
    subprocess.run([os.path.join(os.path.dirname(api.daligner_binary()), "datander"), "-j16"] + ["SIM.%d" % b for b in sorted(want)],
                   cwd=cli, check=True, stdout=subprocess.DEVNULL)
    for blk, (m, novl) in sorted(want.items()):
        assert _md5(os.path.join(cli, "tan", "SIM.%d.SIM.%d.las" % (blk, blk))) == m, blk


@pytest.mark.parametrize("name,upr", [("tiny2", 1), ("tiny2", 5), ("tandem", 3)])
def test_gpu_work_queue_and_split_pairs_equal_reference_golden(gpu, tmp_path, name, upr):
    """damar_amd.multi's queue on one rank: the units of the plan (with `upr` > 1 every pair split by
    B-read range, damar_set_bread_range) through GpuRunner, parts merged -> the reference's golden files."""
    from conftest import read_case, link_db, compare_las
    from damar_amd import multi
    case = read_case(name)
    work = str(tmp_path)
    link_db(case["dbdir"], work)
    nblocks = int(open(os.path.join(work, "G.db")).read().split("blocks =")[1].split()[0])
    world = 4 if upr > 1 else 1                      # pretend ranks: only the granularity matters here
    units = multi.work_units(nblocks, world, units_per_rank=upr)
    if upr > 1:
        assert max(n for _, _, _, n in units) > 1
    runner = multi.GpuRunner(dict(j=4), max_blocks=1)
    mine = multi.run_queue(os.path.join(work, "G"), units, work, multi.LocalQueue(len(units)), runner)
    runner.finish()
    assert len(mine) == len(units)
    for r in range(world):
        multi.merge_parts(os.path.join(work, "G"), units, work, r, world)
    runner.close()
    assert compare_las(case, work) == []


@pytest.mark.parametrize("name,front,cap", [("tiny2", False, None), ("mask_two", True, None), ("tiny_s", False, "2"),
                                            ("prod", True, "2"), ("tandem", False, None), ("fusion", False, None)])
def test_gpu_cli_plan_mode_equals_reference_golden(gpu, tmp_path, name, front, cap):
    """`daligner -P <plan>`: all lines of an HPCdaligner-style plan (comment and LAmerge lines included) in one
    process with blocks and indexes resident; `front` puts the options before -P instead of into the lines,
    `cap` squeezes the block table (DAMAR_PLAN_BLOCKS) so that blocks are evicted and read again."""
    import subprocess
    from conftest import read_case, link_db, compare_las
    from damar_amd import api
    case = read_case(name)
    work = str(tmp_path)
    link_db(case["dbdir"], work)
    opts = " ".join(case["opts"])
    with open(os.path.join(work, "plan.txt"), "w") as f:
        f.write("# Daligner jobs (%d)\n" % len(case["lines"]))
        for a, bs in case["lines"]:
            f.write("daligner %s G.%s %s\n" % ("" if front else opts, a, " ".join("G." + b for b in bs)))
        f.write("# merge jobs\nLAmerge -n 8 G.db G.1.las d001_00001\n")
    env = dict(os.environ)
    if cap:
        env["DAMAR_PLAN_BLOCKS"] = cap
    cmd = [api.daligner_binary()] + (case["opts"] if front else []) + ["-P", "plan.txt"]
    subprocess.run(cmd, cwd=work, check=True, env=env, stdout=subprocess.DEVNULL)
    assert compare_las(case, work) == []
    # the same plan on stdin
    for rel in case["las"]:
        os.unlink(os.path.join(work, rel))
    with open(os.path.join(work, "plan.txt")) as f:
        subprocess.run(cmd[:-1] + ["-"], cwd=work, check=True, env=env, stdin=f, stdout=subprocess.DEVNULL)
    assert compare_las(case, work) == []


@pytest.mark.parametrize("env_extra", [{"DAMAR_EARLY_CUT": "1"}, {"DAMAR_BATCH": "1"}, {"DAMAR_BATCH": "16"},
                                       {"DAMAR_PACKED": "0", "DAMAR_BATCH": "2"}, {"DAMAR_EARLY_CUT": "1", "DAMAR_TEST_SMALL_CAPS": "1"},
                                       {"DAMAR_OVERLAP": "0"}, {"DAMAR_OVERLAP": "1", "DAMAR_TEST_SMALL_CAPS": "1", "DAMAR_BATCH": "8"},
                                       {"DAMAR_OVERLAP": "2"}, {"DAMAR_LAUNCH_QUEUE": "2", "DAMAR_BATCH": "1"},
                                       {"DAMAR_LAUNCH_QUEUE": "2", "DAMAR_TEST_SMALL_CAPS": "1", "DAMAR_BATCH": "2"},
                                       {"DAMAR_SEED_PRIO": "0"}, {"DAMAR_SEED_PRIO": "7", "DAMAR_BATCH_WORK": "1"},
                                       {"DAMAR_DB_UNPACKED": "1"}, {"DAMAR_PLAN_TIDY": "1", "DAMAR_PLAN_RELEASE": "1"},
                                       {"DAMAR_DEVICE_T8": "0"}, {"DAMAR_TEST_T8_LIMIT": "40", "DAMAR_BATCH": "2"},
                                       {"DAMAR_TEST_MAX_CELLS": "512"}, {"DAMAR_TEST_MAX_CELLS": "1024", "DAMAR_BATCH": "1", "DAMAR_DEVICE_T8": "0"}])
@pytest.mark.parametrize("name", ["tiny2", "prod"])
def test_gpu_cli_plan_mode_other_launch_shapes_equal_reference_golden(gpu, tmp_path, name, env_extra):
    """The switches that change how the work reaches the report kernel -- the early cut of the seed pairs, the number of
    comparisons per launch, one read pair per wavefront, report launches in flight beside the next seed stages or not
    (with re-launches after buffer overflows), kernels in order on the device with the host pipelined, two launches in
    flight (the second queued behind the first, also with re-launches), the wave priority of the seed kernels, blocks
    unpacked and complemented on the host instead of kept packed and unpacked by the GPU (the default since round 5),
    one process that releases everything itself, trace values compressed to bytes by the host instead of by the report
    kernel, and a byte limit so low that every launch is repeated with 16-bit values (what a value above 255 does) --
    a pebble pool so small that most read pairs overflow it and are done again by the wide kernel (16-byte pebbles; what a
    pair beyond 2^18 pebbles a direction does) -- must not change a byte of the output."""
    import subprocess
    from conftest import read_case, link_db, compare_las
    from damar_amd import api
    case = read_case(name)
    work = str(tmp_path)
    link_db(case["dbdir"], work)
    with open(os.path.join(work, "plan.txt"), "w") as f:
        for a, bs in case["lines"]:
            f.write("daligner %s G.%s %s\n" % (" ".join(case["opts"]), a, " ".join("G." + b for b in bs)))
    r = subprocess.run([api.daligner_binary(), "-v", "-P", "plan.txt"], cwd=work, check=True, env=dict(os.environ, **env_extra),
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert compare_las(case, work) == []
    if "DAMAR_TEST_MAX_CELLS" in env_extra and name == "tiny2":
        assert "wide kernel" in r.stderr                 # (the hook really sent pairs that way: 10 kb reads drop ~2 500 pebbles a direction)


@pytest.mark.parametrize("async_tail", [False, True])
def test_gpu_match_batch_with_different_a_blocks_in_one_call(gpu, tmp_path, async_tail):
    """damar_match_batch takes arbitrary comparisons: here ALL six of the two-block fixture -- two A blocks, self and cross,
    both orientations -- in one call (DAMAR_BATCH then cuts the launches), one Align_Spec per block pair, the files
    written afterwards.  Synchronous (counts filled, launches completed inside the call) and asynchronous (the last launch in
    flight at return, the write requests queued behind its tails)."""
    import ctypes as C
    from conftest import read_case, link_db, compare_las
    from damar_amd import api, driver
    from damar_amd.driver import _cwd
    case = read_case("tiny2")
    work = str(tmp_path)
    link_db(case["dbdir"], work)
    L = api.lib()
    from conftest import opts_to_plan_kwargs
    plan = driver.Plan(async_tail=async_tail, **opts_to_plan_kwargs(case["opts"]))
    b1, b2 = driver.Block(os.path.join(work, "G.1")), driver.Block(os.path.join(work, "G.2"))
    with _cwd(work):
        for part in (1, 2):
            os.makedirs(api.get_dir(1, part), exist_ok=True)
        pairs = [(b1, b1), (b2, b2), (b2, b1)]
        jobs, specs = [], []
        for q, (a, b) in enumerate(pairs):
            spec = plan._spec(a, q)
            specs.append(spec)
            aidx = plan._index(a, 0)
            bidx, cidx = (aidx if b is a else plan._index(b, 0)), plan._index(b, 1)
            same = 1 if b is a else 0
            jobs += [(a.db, b.db, aidx, bidx, same, 0, spec), (a.db, b.cdb, aidx, cidx, same, 1, spec)]
        plan._match_batch(jobs)
        outabs = os.path.abspath(work)
        for (a, b), spec in zip(pairs, specs):
            d1 = os.path.join(outabs, api.get_dir(1, a.db.part)).encode()
            if b is a:
                L.damar_write_overlaps(spec, d1, None, a.root.encode(), a.root.encode(), a.last_read())
            else:
                d2 = os.path.join(outabs, api.get_dir(1, b.db.part)).encode()
                last = b.last_read() if b.db.part < a.db.part else a.last_read()
                L.damar_write_overlaps(spec, d1, d2, a.root.encode(), b.root.encode(), last)
        plan.finish()
    assert plan.matches == 6 and plan.counts[0] > 0 and plan.counts[1] > 0 and plan.counts[2] > 0
    assert compare_las(case, work) == []
    plan2 = driver.Plan(async_tail=False)       # (leave the library in synchronous mode for the tests that follow)
    plan2.finish()


@pytest.mark.parametrize("cap", [{"DAMAR_PLAN_BLOCKS": "2"}, {"DAMAR_PLAN_GB": "0.02"}])
def test_gpu_cli_plan_mode_evicts_and_rereads_blocks(gpu, tmp_path, cap):
    """A 4-block plan through `daligner -P` with room for 2 blocks only (DAMAR_PLAN_BLOCKS=2), or with a byte budget that
    one block's strands and indexes fill (DAMAR_PLAN_GB: the idle blocks' device copies and indexes are released, the host
    copies stay): every line's B blocks push each other out and are uploaded / indexed again; files equal the CPU oracle's."""
    import filecmp
    import subprocess
    from conftest import ROOT
    from damar_amd import api
    g, o = os.path.join(str(tmp_path), "gpu"), os.path.join(str(tmp_path), "cpu")
    os.makedirs(g)
    os.makedirs(o)
    nb = api.sim_write_db(g, "S", 0.3, coverage=12., seed=23, block_mbp=1)
    assert nb == 4
    for f in ("S.db", ".S.idx", ".S.bps"):
        os.symlink(os.path.join(g, f), os.path.join(o, f))
    lines = [(a, list(range(a, 0, -1))) for a in range(1, nb + 1)]
    with open(os.path.join(g, "plan.txt"), "w") as f:
        for a, bs in lines:
            f.write("daligner -k14 -j4 S.%d %s\n" % (a, " ".join("S.%d" % b for b in bs)))
    r = subprocess.run([api.daligner_binary(), "-P", "plan.txt"], cwd=g, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                       text=True, env=dict(os.environ, DAMAR_CLIPROF="1", **cap))
    builds = int(r.stderr.split(" index builds")[0].split()[-1])
    assert builds > 2 * nb                         # (more than once per block and strand: blocks came back)
    n = 0
    for a, bs in lines:
        subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner"), "-k14", "-j4", "S.%d" % a] + ["S.%d" % b for b in bs],
                       cwd=o, check=True, stdout=subprocess.DEVNULL)
    for dp, _, fs in os.walk(o):
        for f in fs:
            if f.endswith(".las"):
                rel = os.path.relpath(os.path.join(dp, f), o)
                assert filecmp.cmp(os.path.join(dp, f), os.path.join(g, rel), shallow=False), rel
                n += 1
    assert n == nb * nb


@pytest.mark.parametrize("name,ngpu", [("tiny2", 2), ("prod", 2), ("tiny2", 5)])
def test_gpu_cli_node_mode_two_workers_sharing_the_gpu_equals_reference_golden(gpu, tmp_path, name, ngpu):
    """`daligner -P <plan> -G<n>` (dalign/daligner.c:958 + the work list of HPCdaligner.c:628-788, for one node): the parent
    forks n workers before any HIP call, here all on GPU 0 (DAMAR_SHARE_GPU=1); with more workers than half the block
    pairs (tiny2 has 3 pairs: -G2 and -G5 both) the pairs are split by B-read range and the parent merges the parts; -L
    also runs the plan's LAmerge line.  Files equal the reference's golden files, part directories are gone."""
    import subprocess
    from conftest import read_case, link_db, compare_las
    from damar_amd import api
    case = read_case(name)
    work = str(tmp_path)
    link_db(case["dbdir"], work)
    dir1 = [os.path.dirname(rel) for rel in case["las"] if os.path.basename(rel).startswith("G.1.")][0]     # (-r<run> names it)
    with open(os.path.join(work, "plan.txt"), "w") as f:
        f.write("# Daligner jobs (%d)\n" % len(case["lines"]))
        for a, bs in case["lines"]:
            f.write("daligner %s G.%s %s\n" % (" ".join(case["opts"]), a, " ".join("G." + b for b in bs)))
        f.write("# merge jobs\nLAmerge -n 8 G.db G.1.las %s\n" % dir1)
    env = dict(os.environ, DAMAR_SHARE_GPU="1")
    subprocess.run([api.daligner_binary(), "-P", "plan.txt", "-G%d" % ngpu, "-L"], cwd=work, check=True, env=env,
                   stdout=subprocess.DEVNULL)
    assert compare_las(case, work) == []
    assert not os.path.exists(os.path.join(work, "_parts"))
    merged = os.path.join(work, "G.1.las")
    assert os.path.getsize(merged) > 12
    # the merged block file holds exactly the records of the block's directory
    from damar_amd import driver
    n_dir = sum(driver.las_stats(os.path.join(work, dir1, f))[0] for f in os.listdir(os.path.join(work, dir1)))
    assert driver.las_stats(merged)[0] == n_dir


def test_gpu_cli_node_mode_config3_regions_and_stealing_equal_reference(gpu, tmp_path):
    """Config 3 (17 blocks, 153 block pairs) through `daligner -P plan -G5` with the five workers sharing GPU 0 (a rehearsal
    of a node of GPUs within the box's limit of six processes on its card): one region of the plan's triangle per worker,
    cursors in the shared page, leftovers stolen; all 289 files against the reference's md5s; the workers' busy times (start
    to last file closed) within 10 % of their mean; every worker placed (NUMA node or -1, its share of the cores, its thread
    counts) and logged with its units and index builds."""
    import re
    import subprocess
    from damar_amd import api
    d = str(tmp_path)
    nb = api.sim_write_db(d, "SIM", 4.6, coverage=87., seed=3, block_mbp=25)
    assert nb == 17
    with open(os.path.join(d, "plan.txt"), "w") as f:
        for a in range(1, nb + 1):
            f.write("daligner -k14 -j16 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))))
    r = subprocess.run([api.daligner_binary(), "-v", "-P", "plan.txt", "-G5"], cwd=d,
                       env=dict(os.environ, DAMAR_SHARE_GPU="1", DAMAR_PLAN_STATS=os.path.join(d, "node.json")),
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "153 block pairs" in r.stderr
    # the machine-readable line of the run (DAMAR_PLAN_STATS): every worker's units, stolen units, index builds, busy time,
    # phases and output volume -- what a scaling record needs to explain itself
    import json
    st = json.loads(open(os.path.join(d, "node.json")).read())
    assert st["workers"] == 5 and st["block_pairs"] == 153 and st["ok"] == 1 and len(st["worker"]) == 5
    assert sum(w["block_pairs"] for w in st["worker"]) == 153 and sum(w["units"] for w in st["worker"]) == st["units"]
    assert all(w["busy_ms"] > 0 and w["index_builds"] > 0 and w["phase_ms"]["report"] > 0 and w["las_bytes"] > 0 for w in st["worker"])
    assert 1.0 <= st["busy_max_over_mean"] <= 1.5
    line = [ln for ln in r.stderr.splitlines() if "block pairs," in ln and "GPU worker" in ln][-1]
    print(line)
    workers = re.findall(r"\[gpu (\d+): (\d+) units \((\d+) stolen\), (\d+) index builds, ([\d.]+) s busy, numa (-?\d+), (\d+) cpus, "
                         r"(\d+)\+(\d+) tail/write threads\]", line)
    assert len(workers) == 5 and all(int(w[1]) > 0 and int(w[3]) > 0 for w in workers), line
    assert all(int(w[6]) >= 1 and 1 <= int(w[7]) <= 4 and 1 <= int(w[8]) <= 2 for w in workers), line
    # five processes on ONE GPU: wall-clock balance is a logged metric (1.03-1.08 measured), asserted loosely so that a
    # loaded box does not fail the parity test (DAMAR_TEST_BALANCE tightens it)
    balance = float(re.search(r"busy max/mean ([\d.]+)", line).group(1))
    print("node mode -G5 on one GPU: busy max/mean %.3f" % balance)
    assert balance <= float(os.environ.get("DAMAR_TEST_BALANCE", "1.5")), line
    bad = []
    for ln in open(os.path.join(GOLDEN, "config3_ref_md5.txt")):
        if ln.startswith("#"):
            continue
        m, f = ln.split()
        if _md5(os.path.join(d, f)) != m:
            bad.append(f)
    assert not bad, bad[:5]


def test_gpu_cli_node_mode_reports_a_failed_worker(gpu, tmp_path):
    """A worker that cannot do its units (a block of the plan does not exist) exits non-zero; the parent says so and fails."""
    import subprocess
    from conftest import read_case, link_db
    from damar_amd import api
    case = read_case("tiny2")
    work = str(tmp_path)
    link_db(case["dbdir"], work)
    with open(os.path.join(work, "plan.txt"), "w") as f:
        f.write("daligner -k14 G.1 G.1\ndaligner -k14 G.2 G.2 G.1\ndaligner -k14 G.7 G.7 G.2 G.1\n")
    r = subprocess.run([api.daligner_binary(), "-P", "plan.txt", "-G2"], cwd=work, env=dict(os.environ, DAMAR_SHARE_GPU="1"),
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0
    assert "failed" in r.stderr


def test_gpu_two_processes_on_one_gpu_region_queue_and_split_pair(gpu, tmp_path):
    """Two ranks (gloo, both on GPU 0) on a config-3-style database cut to 6 blocks: the plan's 21 pairs from the region
    cursors in the job's store (damar_amd/multi.py), then ONE pair split by B-read range between the two ranks and merged;
    the union of the files equals the CPU oracle's, file by file."""
    import filecmp
    import subprocess
    import sys
    from conftest import ROOT
    from damar_amd import api
    g, o = os.path.join(str(tmp_path), "gpu"), os.path.join(str(tmp_path), "cpu")
    os.makedirs(g)
    os.makedirs(o)
    nb = api.sim_write_db(g, "S", 0.7, coverage=30., seed=31, block_mbp=3, max_blocks=6)
    assert nb == 6
    for f in ("S.db", ".S.idx", ".S.bps"):
        os.symlink(os.path.join(g, f), os.path.join(o, f))
    script = os.path.join(str(tmp_path), "rank.py")
    with open(script, "w") as f:
        f.write('''
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
from damar_amd import api, multi
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
api.lib().damar_hip_init(0)
db, nb, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
store = multi.default_store()
runner = multi.GpuRunner(dict(j=4))
units = multi.work_units(nb, world)                       # regions: one per rank
assert units.parts is not None and len(units.parts) == world
# keep the last cross pair of the plan out of the regions: it is run split by B-read range below
split = (nb, 1)
keep = multi.Units(u for u in units if not (u[0] == split[0] and split[1] in u[1]))
cut = [u for u in units if u[0] == split[0] and split[1] in u[1]]
keep.parts, at = [], 0
for first, end in units.parts:
    n = sum(1 for u in units[first:end] if not (u[0] == split[0] and split[1] in u[1]))
    keep.parts.append((at, at + n)); at += n
rest = multi.Units((a, tuple(b for b in bs if b != split[1]), 0, 1) for a, bs, _, _ in cut if len(bs) > 1)
mine = multi.run_queue(db, keep, out, multi.make_queue(store, "main", keep, rank), runner)
more = multi.Units(list(rest) + [(split[0], split[1], p, world) for p in range(world)])
mine += multi.run_queue(db, more, out, multi.make_queue(store, "more", more, rank), runner)
runner.finish()
dist.barrier()
multi.merge_parts(db, more, out, rank, world)
dist.barrier()
sys.stdout.write("rank " + str(rank) + " ran " + str(len(mine)) + " units\\n"); sys.stdout.flush()      # one write: the ranks share the pipe
assert len(mine) > 0
runner.close()
dist.destroy_process_group()
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", script, os.path.join(g, "S"), str(nb), g], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "rank 0 ran" in r.stdout and "rank 1 ran" in r.stdout
    for a in range(1, nb + 1):
        subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner"), "-k14", "-j4", "S.%d" % a] + ["S.%d" % b for b in range(a, 0, -1)],
                       cwd=o, check=True, stdout=subprocess.DEVNULL)
    n = 0
    for dp, _, fs in os.walk(o):
        for f in fs:
            if f.endswith(".las"):
                rel = os.path.relpath(os.path.join(dp, f), o)
                assert filecmp.cmp(os.path.join(dp, f), os.path.join(g, rel), shallow=False), rel
                n += 1
    assert n == nb * nb

"""N > 1 path on CPU: two gloo ranks shard the block pairs of the 2-block fixture, each
rank computes its share (the CPU oracle stands in for the GPU here -- test only), and the
union of the .las files must equal the reference's golden output; the rank-0 reduction of
time and counters goes through torch.distributed."""
import os
import subprocess
import sys

from conftest import ROOT, GOLDEN, read_case, compare_las

WORKER = r'''
import os, sys, subprocess, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from damar_amd import multi
root, work = sys.argv[1], sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
def oracle_runner(a, bs, outdir):
    subprocess.run([os.path.join(root, "oracle", "oracle_daligner"), "-k14", "-j4", os.path.basename(a)] +
                   [os.path.basename(b) for b in bs], cwd=outdir, check=True, stdout=subprocess.DEVNULL)
dist.barrier()
t0 = time.time()
mine = multi.run_rank(os.path.join(work, "G"), 2, work, rank, world, oracle_runner)
npairs = sum(len(v) for v in mine.values())
dist.barrier()
merged = multi.merge_blocks(os.path.join(work, "G"), 2, work, rank, world)
assert len(merged) == 1 and os.path.exists(merged[0])
dist.barrier()
el, tot = multi.reduce_stats(dist, torch.device("cpu"), time.time() - t0, [npairs, rank + 1])
if rank == 0:
    assert tot == [3.0, 3.0], tot
    assert el > 0
    print("OK", el)
dist.destroy_process_group()
'''


QUEUE_WORKER = r'''
import os, sys, subprocess, time, shutil, struct, tempfile
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from damar_amd import multi
root, work, nblocks, upr = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
firsts = [int(x) for x in open(os.path.join(work, "G.db")).read().split("size =")[1].split()[1:]]

def filter_las(src, dst, keep):
    raw = open(src, "rb").read()
    novl, tspace = struct.unpack("<qi", raw[:12])
    tb = 1 if tspace <= 125 else 2
    off, out, n = 12, [], 0
    for _ in range(novl):
        rec = struct.unpack("<10i", raw[off:off + 40])
        size = 40 + tb * rec[0]
        if keep(rec[7], rec[8]):
            out.append(raw[off:off + size]); n += 1
        off += size
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "wb").write(struct.pack("<qi", n, tspace) + b"".join(out))

def oracle_runner(a, b, outdir, part, nparts):
    # test stand-in for the GPU: the CPU oracle computes the whole pair, the part keeps its B-read range
    if isinstance(b, (list, tuple)):       # a group of subject blocks (one report launch on the GPU)
        for x in b:
            oracle_runner(a, x, outdir, part, nparts)
        return
    ia, ib = int(a.rsplit(".", 1)[1]), int(b.rsplit(".", 1)[1])
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("G.db", ".G.idx", ".G.bps"):
            os.symlink(os.path.join(work, f), os.path.join(tmp, f))
        subprocess.run([os.path.join(root, "oracle", "oracle_daligner"), "-k14", "-j4", "G.%d" % ia, "G.%d" % ib],
                       cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        nb = firsts[ib] - firsts[ib - 1]
        lo, hi = firsts[ib - 1] + nb * part // nparts, firsts[ib - 1] + nb * (part + 1) // nparts
        for dp, _, fs in os.walk(tmp):
            for f in fs:
                if not f.endswith(".las"):
                    continue
                rel = os.path.relpath(os.path.join(dp, f), tmp)
                if ia == ib:
                    keep = lambda ar, br: lo <= min(ar, br) < hi
                elif f == "G.%d.G.%d.las" % (ia, ib):
                    keep = lambda ar, br: lo <= br < hi
                else:
                    keep = lambda ar, br: lo <= ar < hi
                filter_las(os.path.join(dp, f), os.path.join(outdir, rel), keep)

units = multi.work_units(nblocks, world, units_per_rank=upr)
queue = multi.make_queue(multi.default_store(), "t", units, rank)
dist.barrier()
mine = multi.run_queue(os.path.join(work, "G"), units, work, queue, oracle_runner)
dist.barrier()
merged = multi.merge_parts(os.path.join(work, "G"), units, work, rank, world)
dist.barrier()
el, tot = multi.reduce_stats(dist, torch.device("cpu"), 1.0, [len(mine), len(merged)])
if rank == 0:
    assert tot[0] == len(units), (tot, len(units))
    print("OK units=%d split=%d merged=%d" % (len(units), sum(1 for u in units if u[3] > 1), int(tot[1])))
dist.destroy_process_group()
'''


def test_work_units_cover_every_pair_once():
    from damar_amd import multi
    for nb, world, upr in [(4, 1, 2), (4, 8, 2), (17, 8, 2), (2, 2, 4), (3, 4, 3)]:
        units = multi.work_units(nb, world, upr)
        pairs = {}
        for a, bs, i, n in units:
            if isinstance(bs, tuple):          # a group: one A block, up to GROUP subject blocks, never split
                assert 1 <= len(bs) <= multi.GROUP and (i, n) == (0, 1)
            for b in (bs if isinstance(bs, tuple) else (bs,)):
                assert 1 <= b <= a <= nb and 0 <= i < n
                pairs.setdefault((a, b), []).append((i, n))
        assert sorted(pairs) == sorted((a, b) for a in range(1, nb + 1) for b in range(1, a + 1))
        for parts in pairs.values():
            n = parts[0][1]
            assert sorted(parts) == [(i, n) for i in range(n)]
        if world > 1:
            assert len(units) >= min(upr * world, 2 * len(pairs)) or all(n > 1 for _, _, _, n in units if _ != 0)
        if all(not isinstance(b, tuple) for _, b, _, _ in units):
            kinds = [a == b for a, b, _, _ in units]      # split pairs: cross pairs come first, self pairs last
            assert kinds == sorted(kinds)
        elif world == 1:
            cost = [sum(1 if b == a else 2 for b in bs) for a, bs, _, _ in units]
            assert cost == sorted(cost, reverse=True)     # one GPU: groups, most expensive first
        else:
            assert units.parts is not None                # several ranks: groups dealt by region
        q = multi.LocalQueue(len(units))
        got = []
        while True:
            i = q.next()
            if i is None:
                break
            got.append(i)
        assert got == list(range(len(units)))


def test_region_queue_hands_out_every_unit_once_own_region_first():
    from damar_amd import multi

    class FakeStore:                      # the store's atomic add
        def __init__(self):
            self.v = {}

        def add(self, key, n):
            self.v[key] = self.v.get(key, 0) + n
            return self.v[key]

    for nb, world in [(17, 8), (6, 3), (16, 2), (255, 8)]:
        units = multi.work_units(nb, world)
        assert units.parts is not None and len(units.parts) == world
        assert units.parts[0][0] == 0 and units.parts[-1][1] == len(units)
        assert all(units.parts[i][1] == units.parts[i + 1][0] for i in range(world - 1))
        cost = [sum(sum(1 if b == a else 2 for b in bs) for a, bs, _, _ in units[f:e]) for f, e in units.parts]
        assert max(cost) <= 1.5 * (sum(cost) / world) + 4                 # regions of about equal cost
        # a region needs far fewer k-mer indexes than the whole database has (nb forward + nb complement)
        if nb >= 16:
            for f, e in units.parts:
                blocks_a = {a for a, _, _, _ in units[f:e]}
                blocks_b = {b for _, bs, _, _ in units[f:e] for b in bs}
                assert len(blocks_a | blocks_b) + len(blocks_b) <= 0.7 * 2 * nb
        store = FakeStore()
        queues = [multi.RegionQueue(store, "t", units.parts, r) for r in range(world)]
        got, first = [], {}
        live = list(range(world))
        turn = 0
        while live:                        # ranks pull in turn; rank 0 stops early to be stolen from
            r = live[turn % len(live)]
            turn += 1
            i = queues[r].next()
            if i is None:
                live.remove(r)
                continue
            first.setdefault(r, i)
            got.append(i)
        assert sorted(got) == list(range(len(units)))
        for r, i in first.items():
            f, e = units.parts[r]
            assert f <= i < e or f == e    # a rank starts in its own region


def _run_workers(script_text, work, args, port):
    script = os.path.join(work, "worker.py")
    open(script, "w").write(script_text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), script, ROOT, work] + args,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_two_rank_gloo_dynamic_queue_equals_golden(built, tmp_path):
    """The shared cursor hands the 3 block pairs of the 2-block fixture to 2 ranks; the union of their files
    is the reference's golden output."""
    case = read_case("tiny2")
    work = str(tmp_path)
    for f in ("G.db", ".G.idx", ".G.bps"):
        os.symlink(os.path.join(case["dbdir"], f), os.path.join(work, f))
    out = _run_workers(QUEUE_WORKER, work, ["2", "1"], 29541)
    assert "OK units=3 split=0" in out, out[-2000:]
    assert compare_las(case, work) == []


def test_two_rank_gloo_split_pairs_merge_to_golden(built, tmp_path):
    """Too few pairs for the ranks: every pair is split by B-read range, the parts are computed by whichever
    rank pulls them, and the merged part files are byte-identical to the reference's files."""
    case = read_case("tiny2")
    work = str(tmp_path)
    for f in ("G.db", ".G.idx", ".G.bps"):
        os.symlink(os.path.join(case["dbdir"], f), os.path.join(work, f))
    out = _run_workers(QUEUE_WORKER, work, ["2", "4"], 29545)
    assert "OK units=" in out and "split=0" not in out, out[-2000:]
    assert compare_las(case, work) == []


def test_shard_pairs_cover_and_balance():
    from damar_amd import multi
    for nb, world in [(4, 1), (4, 2), (16, 8), (5, 3)]:
        shards, load = multi.shard_pairs(nb, world)
        seen = sorted((a, b) for d in shards for a, bs in d.items() for b in bs)
        assert seen == sorted((a, b) for a in range(1, nb + 1) for b in range(1, a + 1))
        assert max(load) - min(load) <= 2.0
        for d in shards:
            for a, bs in d.items():
                assert bs == sorted(bs, reverse=True) and all(b <= a for b in bs)


def test_two_rank_gloo_plan_equals_golden(built, tmp_path):
    case = read_case("tiny2")
    work = str(tmp_path)
    for f in ("G.db", ".G.idx", ".G.bps"):
        os.symlink(os.path.join(case["dbdir"], f), os.path.join(work, f))
    script = os.path.join(work, "worker.py")
    open(script, "w").write(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", script, ROOT, work],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "OK" in r.stdout
    assert compare_las(case, work) == []
    # the merged block files equal what the reference's LAmerge wrote for the same directories
    import hashlib
    want = {ln.split()[2]: ln.split()[0] for ln in open(os.path.join(GOLDEN, "lamerge_ref_md5.txt"))
            if ln.split()[1] == "tiny2" and ln.split()[3] == "-"}
    for b in (1, 2):
        got = hashlib.md5(open(os.path.join(work, "G.%d.las" % b), "rb").read()).hexdigest()
        assert got == want["d001_%05d" % b]


def test_bench_gpus_argument_starts_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher (the form the driver may use): the N ranks are started as children
    under torch.distributed.run before the parent touches the GPU; fewer visible GPUs than N is an error unless the
    one-GPU rehearsal is asked for; a launcher whose world size contradicts --gpus is an error too."""
    import importlib
    import pytest
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("DAMAR_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    assert bench.relaunch_under_torchrun(4, argv=["--gpus", "4", "--steps", "3"], run=fake_run) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert "DAMAR_BENCH_BACKEND" not in seen["env"] or seen["env"]["DAMAR_BENCH_BACKEND"] != "gloo"

    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.relaunch_under_torchrun(2, argv=["--gpus", "2"], run=fake_run)
    assert "only 1 GPU" in str(e.value)
    # under a preloaded profiler the parent's runtime is up: it must not start GPU ranks
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    for var, val in (("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so"), ("HSA_TOOLS_LIB", "libroctracer64.so"),
                     ("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk.so")):
        monkeypatch.setenv(var, val)
        seen.clear()
        with pytest.raises(SystemExit) as e:
            bench.relaunch_under_torchrun(2, argv=["--gpus", "2"], run=fake_run)
        assert "profiler" in str(e.value) and not seen
        monkeypatch.delenv(var)
    # the count of GPUs comes from the kfd topology (no runtime call) and honours the *_VISIBLE_DEVICES lists
    monkeypatch.undo()
    monkeypatch.setattr(bench.os, "environ", dict(os.environ, HIP_VISIBLE_DEVICES="0,1", ROCR_VISIBLE_DEVICES="3"))
    import glob as _glob
    monkeypatch.setattr(_glob, "glob", lambda pat: [])
    assert bench.visible_gpus() == 0
    monkeypatch.undo()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("DAMAR_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    monkeypatch.setenv("DAMAR_BENCH_SHARE_GPU", "1")
    assert bench.relaunch_under_torchrun(2, argv=["--gpus", "2"], run=fake_run) == 0
    assert seen["env"]["DAMAR_BENCH_BACKEND"] == "gloo"

    # main(): plain start with --gpus 2 goes through the relaunch and exits with the child's code ...
    monkeypatch.setattr(bench, "relaunch_under_torchrun", lambda n: 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    # ... a launcher without --gpus says how many ranks there are (that form worked before --gpus was read) ...
    # ... and a launcher with another world size than --gpus is refused before anything is set up
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "--gpus 4" in str(e.value)

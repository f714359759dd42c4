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

"""Block-pair scheduling over the GPUs of one node.

Block pairs of the HPCdaligner plan are independent (each writes its own .las files), so
they are sharded over ranks with no collective in the data path: one process per GPU
(torchrun / RCCL is used only for the start/end barriers and for summing the counters).
Cross-block pairs cost about twice a self pair (two B index builds, twice the seeds), so
pairs are dealt longest-first to the least-loaded rank; a rank then runs its pairs grouped
by A block so that the A index is built once per group, like one daligner plan line.

    torchrun --nproc-per-node 8 -m damar_amd.multi <dbdir>/<root> <nblocks> <outdir>

When all pairs are done the per-pair files of every block directory are merged into one sorted
<root>.<b>.las (LAmerge, the next step of every HPCdaligner plan), block directories dealt over
the ranks.
"""
import os
import sys
import time


def pair_cost(a, b):
    return 1.0 if a == b else 2.0


def shard_pairs(nblocks, world):
    """[(rank -> {a: [b, ...]})]: every pair (a, b<=a) exactly once, LPT-balanced."""
    pairs = [(a, b) for a in range(1, nblocks + 1) for b in range(a, 0, -1)]
    pairs.sort(key=lambda p: (-pair_cost(*p), p))
    load = [0.0] * world
    out = [dict() for _ in range(world)]
    for a, b in pairs:
        r = min(range(world), key=lambda i: (load[i], i))
        load[r] += pair_cost(a, b)
        out[r].setdefault(a, []).append(b)
    for d in out:
        for a in d:
            d[a].sort(reverse=True)
    return out, load


def reduce_stats(dist, device, elapsed, values):
    """max of the elapsed time and sum of the counters over all ranks (rank-0 report)."""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    v = torch.tensor([float(x) for x in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in v.tolist()]


def run_rank(dbprefix, nblocks, outdir, rank, world, runner):
    """runner(a_block_name, [b_block_names], outdir) performs one group of pairs."""
    shards, _ = shard_pairs(nblocks, world)
    mine = shards[rank]
    for a in sorted(mine):
        runner("%s.%d" % (dbprefix, a), ["%s.%d" % (dbprefix, b) for b in mine[a]], outdir)
    return mine


def merge_blocks(dbprefix, nblocks, outdir, rank, world, run=1):
    """After every rank has finished its pairs (barrier!): the LAmerge step of the plan
    (HPCdaligner.c:790-808), one block directory per call, dealt round-robin over the ranks.
    Writes <root>.<b>.las next to the block directories; returns the files this rank wrote."""
    import subprocess
    from . import api, lib
    root = os.path.basename(dbprefix)
    done = []
    for b in range(1, nblocks + 1):
        if (b - 1) % world != rank:
            continue
        out = "%s.%d.las" % (root, b)
        subprocess.run([lib.bin_path("LAmerge"), "-n", "8", dbprefix, out, api.get_dir(run, b)], cwd=outdir, check=True,
                       stdout=subprocess.DEVNULL)
        done.append(os.path.join(outdir, out))
    return done


def gpu_runner(plan_kwargs=None):
    from . import driver
    cache = {}
    plan = driver.Plan(**(plan_kwargs or {}))

    def run(a, bs, outdir):
        for n in [a] + bs:
            if n not in cache:
                cache[n] = driver.Block(n)
        plan.run_line(cache[a], [cache[b] for b in bs], outdir)
    run.plan = plan
    return run


def main():
    import torch
    import torch.distributed as dist
    dbprefix, nblocks, outdir = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from . import api
    api.lib().damar_hip_init(local)
    runner = gpu_runner()
    dist.barrier()
    t0 = time.time()
    run_rank(dbprefix, nblocks, outdir, rank, world, runner)
    runner.plan.finish()
    torch.cuda.synchronize()
    dist.barrier()
    merge_blocks(dbprefix, nblocks, outdir, rank, world)
    dist.barrier()
    el, cnt = reduce_stats(dist, torch.device("cuda", local), time.time() - t0, runner.plan.counts)
    if rank == 0:
        print("pairs done in %.2f s over %d GPUs: %d seed pairs, %d alignments, %d records" % (el, world, *cnt))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Block-pair scheduling over the GPUs of one node.

The work list is the HPCdaligner plan (dalign/HPCdaligner.c:628-788: block a against blocks a, a-1, ... 1);
the loop it replaces is the `for B` loop of dalign/daligner.c:958 plus the cluster scheduler above it.
Block pairs are independent (each writes its own .las files), so there is no collective in the data
path: one process per GPU, every rank holds the read blocks, and the pairs are handed out DYNAMICALLY
from one shared cursor -- an atomic counter in the job's torch.distributed store (rank 0's TCP store; the
"trivial work queue" of BASELINE.json's north_star).  A rank that finishes early simply pulls the next
unit, so a slow pair, a slow GPU or a busy host core delays nobody else (work stealing without victims).
RCCL is used only for the barriers around the job and the final reduction of the counters.

A unit of the queue is one A block against up to GROUP subject blocks, both orientations each (the plan's
lines cut into groups, most expensive first): the comparisons of a unit share launches of the report kernel
(damar_match_batch), and the A block's k-mer index is built once per rank and stays in its LRU index cache.
The group size is halved until every rank can expect about six units.  When the plan has fewer than two
block pairs per rank (config 2 on 8 GPUs: 10 pairs), the units are single pairs and cross pairs are split by B-read
range (damar_set_bread_range): every part runs the index merge and the seed sort of the whole pair and
then only its share of the read pairs -- the dominant 60 % -- and the parts' sorted files are merged
into exactly the files the unsplit pair writes.

    torchrun --nproc-per-node 8 -m damar_amd.multi <dbdir>/<root> <nblocks> <outdir>

When all pairs are done the per-pair files of every block directory are merged into one sorted
<root>.<b>.las (LAmerge, the next step of every HPCdaligner plan), block directories dealt over the ranks.
"""
import os
import subprocess
import sys
import time


def pair_cost(a, b):
    return 1.0 if a == b else 2.0


def shard_pairs(nblocks, world):
    """Static alternative to the queue: [(rank -> {a: [b, ...]})], every pair (a, b<=a) exactly once,
    longest-processing-time balanced.  Kept for callers without a shared store."""
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


GROUP = 8        # subject blocks behind one report launch (driver.Plan.GROUP)


def group_units(nblocks, g):
    """The plan's lines (block a against a, a-1, ..., 1) cut into groups of at most g subject blocks:
    (a, (b, ...), 0, 1), most expensive first (a cross pair costs 2, a self pair 1)."""
    units = []
    for a in range(1, nblocks + 1):
        bs = list(range(a, 0, -1))
        for k in range(0, len(bs), g):
            units.append((a, tuple(bs[k:k + g]), 0, 1))
    units.sort(key=lambda u: -sum(1 if b == u[0] else 2 for b in u[1]))
    return units


class Units(list):
    """The queue's contents; .parts = [(first, end), ...] index ranges, one per rank, when the units are dealt by region
    (RegionQueue), else None (one cursor for all)."""
    parts = None


_SAT = {}


def _sat(n):
    """Summed-area table of pair_cost over the plan's triangle (b <= a) of n blocks: S[a][b] = cost of {1..a} x {1..b}."""
    t = _SAT.get(n)
    if t is None:
        t = [[0] * (n + 1) for _ in range(n + 1)]
        for a in range(1, n + 1):
            row, up = t[a], t[a - 1]
            for b in range(1, n + 1):
                row[b] = up[b] + row[b - 1] - up[b - 1] + (pair_cost(a, b) if b <= a else 0)
        _SAT[n] = t
    return t


_SAT_N = [0]


def _region_cost(reg):
    """Cost of the pairs {alo <= a <= ahi, blo <= b <= min(a, bhi)}, O(1) (the loops cost 10 s at 255 blocks)."""
    alo, ahi, blo, bhi = reg
    if alo > ahi or blo > bhi:
        return 0
    n = max(_SAT_N[0], ahi, bhi)
    _SAT_N[0] = n
    t = _sat(n)
    return t[ahi][bhi] - t[alo - 1][bhi] - t[ahi][blo - 1] + t[alo - 1][blo - 1]


def _split_region(reg, k):
    """k regions of about equal cost out of the pairs {alo <= a <= ahi, blo <= b <= min(a, bhi)} by recursive bisection,
    each time across the dimension that is longer in index builds (an A block costs one k-mer index, a subject block
    two: both strands)."""
    if k <= 1:
        return [reg]
    alo, ahi, blo, bhi = reg
    alo = max(alo, blo)                                   # rows below blo are empty
    bhi = min(bhi, ahi)
    k1 = k // 2
    want = _region_cost((alo, ahi, blo, bhi)) * k1 / float(k)
    cuts = []
    if ahi > alo:
        cuts += [("a", t) for t in range(alo, ahi)]       # A <= t | A > t
    if bhi > blo:
        cuts += [("b", t) for t in range(blo, bhi)]       # B <= t | B > t
    if not cuts:
        return [(alo, ahi, blo, bhi)]
    prefer = "a" if (ahi - alo + 1) >= 2 * (bhi - blo + 1) else "b"
    best = None
    for dim, t in cuts:
        left = (alo, t, blo, bhi) if dim == "a" else (alo, ahi, blo, t)
        right = (t + 1, ahi, blo, bhi) if dim == "a" else (alo, ahi, t + 1, bhi)
        cl, cr = _region_cost(left), _region_cost(right)
        if cl == 0 or cr == 0:
            continue
        key = (dim != prefer, abs(cl - want))
        if best is None or key < best[0]:
            best = (key, left, right)
    if best is None:
        return [(alo, ahi, blo, bhi)]
    # the preferred dimension unless its best cut is badly off balance
    alt = min(((abs(_region_cost((alo, t, blo, bhi) if d == "a" else (alo, ahi, blo, t)) - want), d, t) for d, t in cuts
               if _region_cost((alo, t, blo, bhi) if d == "a" else (alo, ahi, blo, t)) > 0
               and _region_cost((t + 1, ahi, blo, bhi) if d == "a" else (alo, ahi, t + 1, bhi)) > 0), default=None)
    if alt is not None and best[0][1] > 0.15 * max(want, 1.) and alt[0] < best[0][1]:
        d, t = alt[1], alt[2]
        best = (None, (alo, t, blo, bhi) if d == "a" else (alo, ahi, blo, t),
                (t + 1, ahi, blo, bhi) if d == "a" else (alo, ahi, t + 1, bhi))
    return _split_region(best[1], k1) + _split_region(best[2], k - k1)


def region_units(nblocks, world, g):
    """The block pairs cut into `world` regions of about equal cost (A range x subject range of the plan's triangle), each
    region's pairs as units (a, (b, ...), 0, 1) of at most g subject blocks.  A rank that works through ONE region builds
    the k-mer indexes of that region's blocks only -- with pairs dealt from one list every rank ends up building nearly
    every index of the database, which at 8 GPUs costs a quarter of the step."""
    regs = _split_region((1, nblocks, 1, nblocks), world)
    units = Units()
    parts = []
    for alo, ahi, blo, bhi in regs:
        first = len(units)
        mine = []
        for a in range(alo, ahi + 1):
            bs = [b for b in range(min(a, bhi), blo - 1, -1)]
            for k in range(0, len(bs), g):
                mine.append((a, tuple(bs[k:k + g]), 0, 1))
        # most expensive first: what a rank does last (and whose host tail it then waits for) is small
        mine.sort(key=lambda u: -sum(pair_cost(u[0], b) for b in u[1]))
        units.extend(mine)
        parts.append((first, len(units)))
    while len(parts) < world:                              # fewer regions than ranks: the others only steal
        parts.append((len(units), len(units)))
    units.parts = parts
    return units


def work_units(nblocks, world, units_per_rank=2, fine=6):
    """The queue's contents.  Enough block pairs for the ranks: groups (a, (b, ...), 0, 1) of one A block against up to
    GROUP subject blocks -- one report launch each, the larger the group the smaller the share of the launch spent
    waiting for its longest alignment -- as large as still leaves `fine` units per rank (one GPU: whole plan lines), dealt
    by region for several ranks (region_units).
    Too few pairs for `world` ranks: (a, b, part, nparts) with b <= a, cross pairs first in plan order, then the self
    pairs, split by B-read range."""
    cross = [(a, b) for a in range(1, nblocks + 1) for b in range(a - 1, 0, -1)]
    selfs = [(a, a) for a in range(1, nblocks + 1)]
    npairs = len(cross) + len(selfs)
    if world == 1 or npairs >= units_per_rank * world:
        g = GROUP
        while g > 1 and world > 1 and len(group_units(nblocks, g)) < fine * world:
            g //= 2
        return Units(group_units(nblocks, g)) if world == 1 else region_units(nblocks, world, g)
    ncross = nself = 1
    if world > 1 and npairs < units_per_rank * world:
        # cost units (cross = 2, self = 1) per rank wanted: split so that every rank gets about units_per_rank pieces
        ncross = -(-units_per_rank * world // npairs)          # ceil
        nself = max(1, ncross // 2)
    units = Units((a, b, i, ncross) for a, b in cross for i in range(ncross))
    units += [(a, b, i, nself) for a, b in selfs for i in range(nself)]
    return units


class LocalQueue:
    """Single-process stand-in for the shared cursor."""

    def __init__(self, n):
        self.n, self.i = n, 0

    def next(self):
        i = self.i
        self.i += 1
        return i if i < self.n else None


class StoreQueue:
    """Shared cursor over `n` units: one atomic add per pull on the job's torch.distributed store (served by
    rank 0).  `name` must be new for every pass (e.g. "step3")."""

    def __init__(self, store, name, n):
        self.store, self.key, self.n = store, "damar/cursor/%s" % name, n

    def next(self):
        i = self.store.add(self.key, 1) - 1
        return i if i < self.n else None


class RegionQueue:
    """One cursor per region (`parts`: index ranges into the unit list, one per rank): a rank works through its own
    region and then takes what is left of the others', the next rank's first -- affinity without giving up the
    dynamic balance.  Cursors are atomic adds on the job's store like StoreQueue's."""

    def __init__(self, store, name, parts, rank):
        self.store, self.name, self.parts = store, name, list(parts)
        self.order = [(rank + i) % len(self.parts) for i in range(len(self.parts))]
        self.done = set()
        self.stolen = 0          # units this rank took out of another rank's region

    def next(self):
        for p in self.order:
            if p in self.done:
                continue
            first, end = self.parts[p]
            i = self.store.add("damar/cursor/%s/%d" % (self.name, p), 1) - 1 if end > first else end - first
            if i < end - first:
                if p != self.order[0]:
                    self.stolen += 1
                return first + i
            self.done.add(p)
        return None


def make_queue(store, name, units, rank=0):
    """The queue for one pass over `units` (name must be new for every pass)."""
    if store is None:
        return LocalQueue(len(units))
    if getattr(units, "parts", None):
        return RegionQueue(store, name, units.parts, rank)
    return StoreQueue(store, name, len(units))


def default_store():
    """The store torch.distributed's process group was initialised with (torchrun: TCP store on rank 0)."""
    import torch.distributed as dist
    from torch.distributed import distributed_c10d
    if not dist.is_initialized():
        raise RuntimeError("init_process_group first")
    return distributed_c10d._get_default_store()


def part_dir(outdir, a, b, part, nparts):
    return os.path.join(outdir, "_parts", "%d.%d" % (a, b), "p%dof%d" % (part, nparts))


def run_queue(dbprefix, units, outdir, queue, runner):
    """Pull units until the queue is empty.  runner(a_name, b_name or [b_names], outdir, part, nparts) computes one
    unit; a split unit is written under part_dir().  Returns the units this rank ran."""
    mine = []
    while True:
        i = queue.next()
        if i is None:
            break
        a, b, part, nparts = units[i]
        if isinstance(b, tuple):             # a group of subject blocks
            runner("%s.%d" % (dbprefix, a), ["%s.%d" % (dbprefix, x) for x in b], outdir, 0, 1)
        else:
            dst = outdir if nparts == 1 else part_dir(outdir, a, b, part, nparts)
            runner("%s.%d" % (dbprefix, a), "%s.%d" % (dbprefix, b), dst, part, nparts)
        mine.append(units[i])
    return mine


def run_rank(dbprefix, nblocks, outdir, rank, world, runner):
    """Static variant (shard_pairs): runner(a_block_name, [b_block_names], outdir) performs one group."""
    shards, _ = shard_pairs(nblocks, world)
    mine = shards[rank]
    for a in sorted(mine):
        runner("%s.%d" % (dbprefix, a), ["%s.%d" % (dbprefix, b) for b in mine[a]], outdir)
    return mine


def merge_parts(dbprefix, units, outdir, rank, world, run=1):
    """After every rank has finished (barrier!): the files of split pairs.  Every part holds the records of
    its B-read range, sorted; records of one (aread, bread) pair never span parts, so a merge on the record
    order (bin/LAmerge on the named files) restores exactly the file of the unsplit pair.  Split pairs are
    dealt round-robin over the ranks.  Returns the files written."""
    from . import api, lib
    root = os.path.basename(dbprefix)
    split = sorted({(a, b, n) for a, b, _, n in units if n > 1})       # (groups are never split)
    done = []
    for j, (a, b, n) in enumerate(split):
        if j % world != rank:
            continue
        rels = [os.path.join(api.get_dir(run, a), "%s.%d.%s.%d.las" % (root, a, root, b))]
        if a != b:
            rels.append(os.path.join(api.get_dir(run, b), "%s.%d.%s.%d.las" % (root, b, root, a)))
        for rel in rels:
            srcs = [os.path.join(part_dir(outdir, a, b, p, n), rel) for p in range(n)]
            srcs = [s for s in srcs if os.path.exists(s)]
            if not srcs:
                continue
            os.makedirs(os.path.dirname(os.path.join(outdir, rel)), exist_ok=True)
            subprocess.run([lib.bin_path("LAmerge"), dbprefix, os.path.join(outdir, rel)] + srcs, check=True,
                           stdout=subprocess.DEVNULL)
            done.append(os.path.join(outdir, rel))
    return done


def merge_blocks(dbprefix, nblocks, outdir, rank, world, run=1):
    """After every rank has finished its pairs (barrier!): the LAmerge step of the plan
    (HPCdaligner.c:790-808), one block directory per call, dealt round-robin over the ranks.
    Writes <root>.<b>.las next to the block directories; returns the files this rank wrote."""
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


def reduce_stats(dist, device, elapsed, values):
    """max of the elapsed time and sum of the counters over all ranks (rank-0 report)."""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    v = torch.tensor([float(x) for x in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in v.tolist()]


class GpuRunner:
    """One rank's executor: a driver.Plan (LRU index cache bounded by `index_cache_bytes`) and the read
    blocks it has opened.  `resident` blocks (uploaded before the timed region by the caller) are kept;
    blocks the runner opens itself are closed again once more than `max_blocks` are open (least recently
    used first), so neither HBM nor host memory grows with the number of blocks of the database."""

    def __init__(self, plan_kwargs=None, resident=None, max_blocks=8):
        from . import driver
        self.driver = driver
        self.plan = driver.Plan(**(plan_kwargs or {}))
        self.resident = dict(resident or {})
        self.cache = {}                  # name -> Block, LRU order
        self.max_blocks = max_blocks

    def block(self, name, keep=()):
        if name in self.resident:
            return self.resident[name]
        blk = self.cache.pop(name, None)
        if blk is None:
            blk = self.driver.Block(name)
        self.cache[name] = blk
        for old in list(self.cache):
            if len(self.cache) <= self.max_blocks:
                break
            if old == name or old in keep:
                continue
            victim = self.cache.pop(old)
            self.plan.drop_block(victim)
            victim.close()
        return blk

    def __call__(self, a, b, outdir, part=0, nparts=1):
        ba = self.block(a)
        names = list(b) if isinstance(b, (list, tuple)) else [b]
        bbs = [ba if x == a else self.block(x, keep=[a] + names) for x in names]
        self.plan.run_pairs(ba, bbs, outdir, part, nparts)

    def run_line(self, a, bs, outdir):
        for k in range(0, len(bs), GROUP):
            self(a, bs[k:k + GROUP], outdir)

    def end_pass(self):
        """End of one pass over the plan when another follows in the same job: the next pass builds its indexes again."""
        self.plan.release_indexes()

    def finish(self):
        self.plan.finish()

    def close(self):
        self.plan.finish()
        for blk in self.cache.values():
            blk.close()
        self.cache = {}


def gpu_runner(plan_kwargs=None):
    """runner(a, [b...], outdir) for run_rank (static sharding)."""
    r = GpuRunner(plan_kwargs)

    def run(a, bs, outdir):
        r.run_line(a, bs, outdir)
    run.plan = r.plan
    run.runner = r
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
    runner = GpuRunner()
    units = work_units(nblocks, world)
    queue = make_queue(default_store(), "main", units, rank)
    dist.barrier()
    t0 = time.time()
    mine = run_queue(dbprefix, units, outdir, queue, runner)
    runner.finish()
    torch.cuda.synchronize()
    dist.barrier()
    merge_parts(dbprefix, units, outdir, rank, world)
    dist.barrier()
    merge_blocks(dbprefix, nblocks, outdir, rank, world)
    dist.barrier()
    el, cnt = reduce_stats(dist, torch.device("cuda", local), time.time() - t0, runner.plan.counts + [len(mine)])
    if rank == 0:
        print("%d units done in %.2f s over %d GPUs: %d seed pairs, %d alignments, %d records"
              % (cnt[3], el, world, cnt[0], cnt[1], cnt[2]))
    runner.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

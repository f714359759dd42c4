"""In-process mirror of the block-pair loop of dalign/daligner.c:948-1074 on top of the
C-ABI library (ctypes), used by tests/ and bench.py so that the timed region can start
with the read blocks already resident in HBM.  The production host driver is the C binary
damar_amd/bin/daligner; both call the same library entry points in the same order.
"""
import ctypes as C
import os
import contextlib

from . import api


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


class Block:
    """A read block loaded on the host (HITS_DB) and, optionally, resident in HBM."""

    def __init__(self, name):
        self.name = name                       # e.g. "SIM.3"
        self.root = os.path.basename(name)
        if self.root.endswith(".db"):
            self.root = self.root[:-3]
        self.db = api.read_block(name)
        self.masks = None
        self.dev = None
        self._cdb = None
        self.cdev = None

    def load_masks(self, names):
        """-m tracks (daligner.c:442-497): must happen before the block goes to HBM."""
        names = list(names or [])
        if self.masks is None:
            if self.dev is not None or self.cdev is not None:
                if names:
                    raise RuntimeError("mask tracks must be loaded before the block is uploaded")
            elif names:
                arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
                if api.lib().damar_load_masks(C.byref(self.db), arr, len(names)) != 0:
                    raise RuntimeError("cannot load mask tracks %r of %s" % (names, self.name))
            self.masks = names
        elif self.masks != names:
            raise RuntimeError("block %s already carries masks %r" % (self.name, self.masks))

    def upload(self):
        L = api.lib()
        if self.dev is None:
            self.dev = L.damar_block_upload(C.byref(self.db))
        return self.dev

    def upload_complement(self):
        """Reverse-complemented copy (daligner.c:529-570) in a record this object owns, resident in HBM
        as well."""
        L = api.lib()
        if self.cdev is None:
            self._cdb = api.HITS_DB()
            L.damar_complement_copy(C.byref(self.db), C.byref(self._cdb))
            self.cdev = L.damar_block_upload(C.byref(self._cdb))
        return self.cdev

    def release_device(self):
        """Give the block's HBM back (bases of both strands); the host copy stays.  The caller must have
        released every index built on it first (Plan.drop_block does)."""
        L = api.lib()
        for d in (self.dev, self.cdev):
            if d is not None:
                L.damar_block_free(d)
        self.dev = self.cdev = None

    def close(self):
        """Release everything: device blocks, the complement's host bases and masks, the host block."""
        L = api.lib()
        L.damar_async_drain()                 # the host tail reads the blocks' bases (bridges)
        self.release_device()
        if self._cdb is not None:
            L.damar_free_complement(C.byref(self._cdb))
            self._cdb = None
        if self.db is not None:
            L.damar_close_block(C.byref(self.db))
            self.db = None

    @property
    def cdb(self):
        return self._cdb

    def last_read(self):
        return self.db.ufirst + self.db.nreads - 1


class Plan:
    """daligner <A> <B1> <B2> ... for resident blocks."""

    def __init__(self, k=14, w=6, h=35, t=0, e=.70, l=1000, s=100, j=4, run=1,
                 symmetric=1, identity=0, verbose=0, async_tail=True, masks=None, biased=0,
                 only_identity=0, no_trace=0, index_cache_bytes=None):
        self.k, self.w, self.h, self.t, self.e, self.l, self.s, self.j, self.run = k, w, h, t, e, l, s, j, run
        self.symmetric, self.identity, self.verbose = symmetric, identity, verbose
        self.only_identity, self.no_trace = only_identity, no_trace      # daligner -O, -T (daligner.c:731-739)
        self.masks = list(masks or [])
        L = api.lib()
        api.set_globals(verbose=verbose, minover=2 * l, symmetric=symmetric, identity=identity, biased=biased)
        L.damar_bias_reset()
        if L.Set_Filter_Params(k, w, t, h, j):
            raise ValueError("Illegal combination of filter parameters")
        self.timings = {}
        self.counts = [0, 0, 0]
        self.async_tail = async_tail
        self._specs = []
        self._spec_of = {}
        self._idx = {}           # (block name, comp) -> k-mer index resident in HBM for this job (LRU order)
        self._idx_bytes = 0
        # residency cap of the index cache (each index: 8 B per k-mer + a prefix table of up to 1 GB): least
        # recently used indexes are released beyond it.  Default: a third of a 288 GB MI355X.
        self.index_cache_bytes = (96 << 30) if index_cache_bytes is None else index_cache_bytes
        self._pinned = set()
        self._line_a = None
        self._group = []         # index keys the comparisons of the group being prepared refer to
        self.index_builds = 0
        self.matches = 0             # comparisons (block pair x orientation) run
        self.report_launches = 0     # launches of the report kernel they took
        self.wave = [0, 0, 0]        # band cells, wave steps per alignment pass, wave-loop iterations (damar_wave_totals)
        L.damar_set_async(1 if async_tail else 0)

    def finish(self):
        """Drain the asynchronous host tail, release the Align_Specs, fold in the counters."""
        L = api.lib()
        n, t, w = api.c_int64(0), C.c_double(0), C.c_double(0)
        L.damar_async_totals(C.byref(n), C.byref(t), C.byref(w))
        if self.async_tail:
            nf, rms, nl = api.c_int64(0), C.c_double(0), api.c_int64(0)
            L.damar_async_counts(C.byref(nf), C.byref(rms), C.byref(nl))
            self.counts[1] += nf.value
            self.report_launches += nl.value
            self.timings["report"] = self.timings.get("report", 0.) + rms.value
            self.counts[2] += n.value
            self.timings["tail"] = self.timings.get("tail", 0.) + t.value
            self.timings["write"] = self.timings.get("write", 0.) + w.value
            self.timings["d2h"] = self.timings.get("d2h", 0.) + L.damar_async_d2h_ms()
        wc, wh, wi = api.c_int64(0), api.c_int64(0), api.c_int64(0)
        L.damar_wave_totals(C.byref(wc), C.byref(wh), C.byref(wi))
        self.wave = [self.wave[0] + wc.value, self.wave[1] + wh.value, self.wave[2] + wi.value]
        for sp in self._specs:
            L.Free_Align_Spec(sp)
        self._specs = []
        self._spec_of = {}
        for idx in self._idx.values():
            L.damar_index_free(idx)
        self._idx = {}
        self._idx_bytes = 0

    def release_indexes(self):
        """Forget every k-mer index (a new pass over the plan builds them again, as a new job would) WITHOUT draining the
        pipeline: the report launch in flight and the host tail do not read indexes."""
        L = api.lib()
        for idx in self._idx.values():
            L.damar_index_free(idx)
        self._idx = {}
        self._idx_bytes = 0

    def drop_block(self, block):
        """Release the indexes of a block (both strands) and its HBM copy: for a scheduler that knows the
        block is not needed again soon."""
        L = api.lib()
        for comp in (0, 1):
            idx = self._idx.pop((block.name, comp), None)
            if idx is not None:
                self._idx_bytes -= L.damar_index_bytes(idx)
                L.damar_index_free(idx)
        block.release_device()

    def _evict(self):
        L = api.lib()
        for key in list(self._idx):
            if self._idx_bytes <= self.index_cache_bytes:
                break
            if key in self._pinned:
                continue
            idx = self._idx.pop(key)
            self._idx_bytes -= L.damar_index_bytes(idx)
            L.damar_index_free(idx)

    def _acc(self):
        for n, v in api.timings().items():
            self.timings[n] = self.timings.get(n, 0.) + v

    def _index(self, block, comp):
        """Sort_Kmers of a block (or of its complement), once per job: the reference rebuilds an
        index for every command line that names the block (daligner.c:1000, 1025-1047) because
        each line is its own process; here the sorted index (8 B per k-mer) simply stays in HBM
        until finish()."""
        key = (block.name, comp)
        self._pinned = set(self._group) | {self._line_a, key}     # the indexes of the running group stay resident
        self._group.append(key)
        idx = self._idx.pop(key, None)
        if idx is not None:
            self._idx[key] = idx              # most recently used last
        if idx is None:
            block.load_masks(self.masks)
            L = api.lib()
            n = C.c_int(0)
            idx = L.damar_index_build(block.upload_complement() if comp else block.upload(), 0, C.byref(n))
            t = api.timings()
            for nme in ("tuples", "ksort", "table"):
                self.timings[nme] = self.timings.get(nme, 0.) + t[nme]
            self._idx[key] = idx
            self._idx_bytes += L.damar_index_bytes(idx)
            self.index_builds += 1
            self._evict()
        return idx

    def _match_batch(self, jobs):
        """jobs: list of (adb, bdb, aidx, bidx, self, comp, spec) -> ONE damar_match_batch: the seed stages one after
        the other, one report launch over all of them (include/damar_hip.h)."""
        L = api.lib()
        arr = (api.MatchJob * len(jobs))()
        for q, (adb, bdb, aidx, bidx, self_, comp, spec) in enumerate(jobs):
            arr[q].ablock = C.pointer(adb)
            arr[q].bblock = C.pointer(bdb)
            arr[q].aidx, arr[q].bidx = aidx, bidx
            arr[q].self_, arr[q].comp, arr[q].spec = self_, comp, spec
        L.damar_match_batch(arr, len(jobs))
        self.matches += len(jobs)
        if not self.async_tail:
            self.report_launches += api.counters()[5]
        t = api.timings()
        # (asynchronous mode: the last report launch of the call is still in flight -- its time and its seed hits come
        #  from damar_async_counts at finish())
        for nme in ("merge", "ssort", "work", "d2h", "tail") + (() if self.async_tail else ("report",)):
            self.timings[nme] = self.timings.get(nme, 0.) + t[nme]
        for q in range(len(jobs)):
            for i in ((0, 2) if self.async_tail else (0, 1, 2)):
                self.counts[i] += arr[q].counts[i]

    def _spec(self, a, slot=0):
        """Align_Specs per A block and job (the tables depend on the block's base frequencies, daligner.c:951), one
        per subject block of a group: each has its own overlap buffers, which damar_write_overlaps writes and resets."""
        L = api.lib()
        sps = self._spec_of.setdefault(a.name, [])
        while len(sps) <= slot:
            sp = L.New_Align_Spec(self.e, self.s, a.db.freq, self.j, self.symmetric, self.only_identity,
                                  self.no_trace, 1)
            sps.append(sp)
            self._specs.append(sp)
        return sps[slot]

    GROUP = 8        # subject blocks per report launch (x 2 orientations = DAMAR_MAX_JOBS comparisons)

    def run_pairs(self, a, bs, outdir, part=0, nparts=1):
        """Block `a` against the blocks `bs` (at most GROUP), both orientations each (daligner.c:958-1056 for these B
        arguments); the .las files go under outdir exactly as
        daligner.c:1006-1021, 1051-1056 name them.  With nparts > 1 only the read pairs whose B read falls into the
        part's share of B's reads are processed (damar_set_bread_range; one subject block then): the parts' files merge
        into the unsplit pair's files (multi.merge_parts)."""
        L = api.lib()
        assert 1 <= len(bs) <= self.GROUP and (nparts == 1 or len(bs) == 1)
        os.makedirs(outdir, exist_ok=True)
        outabs = os.path.abspath(outdir)

        def odir(part_no):       # absolute: the write may run later on the worker thread
            return os.path.join(outabs, api.get_dir(self.run, part_no)).encode() if part_no > 0 else None
        with _cwd(outdir):
            os.makedirs(api.get_dir(self.run, a.db.part), exist_ok=True)
            self._line_a = (a.name, 0)
            self._group = []
            aidx = self._index(a, 0)
            if nparts > 1:
                nb = bs[0].db.nreads
                L.damar_set_bread_range(nb * part // nparts, nb * (part + 1) // nparts)
            try:
                # One damar_match_batch per subject block (N and C): its k-mer indexes are built just before its seed
                # stages, i.e. beside the report launch of the block before it (the library keeps that launch in flight
                # across calls), and its files are queued behind its own tails.
                for q, b in enumerate(bs):
                    spec = self._spec(a, q)
                    self._group = [(a.name, 0)]
                    if b is a or b.name == a.name:
                        cidx = self._index(a, 1)
                        self._match_batch([(a.db, a.db, aidx, aidx, 1, 0, spec), (a.db, a.cdb, aidx, cidx, 1, 1, spec)])
                        L.damar_write_overlaps(spec, odir(a.db.part), None, a.root.encode(), a.root.encode(), a.last_read())
                    else:
                        if self.symmetric:
                            os.makedirs(api.get_dir(self.run, b.db.part), exist_ok=True)
                        bidx = self._index(b, 0)
                        cidx = self._index(b, 1)
                        self._match_batch([(a.db, b.db, aidx, bidx, 0, 0, spec), (a.db, b.cdb, aidx, cidx, 0, 1, spec)])
                        last = b.last_read() if b.db.part < a.db.part else a.last_read()
                        L.damar_write_overlaps(spec, odir(a.db.part), odir(b.db.part), a.root.encode(), b.root.encode(), last)
            finally:
                if nparts > 1:
                    L.damar_set_bread_range(0, -1)
            self._group = []
            if not self.async_tail or a.db.part <= 0 or any(b.db.part <= 0 for b in bs):
                self.finish()            # unsplit DBs write relative paths: finish inside this cwd

    def run_pair(self, a, b, outdir, part=0, nparts=1):
        """One block pair of a plan line, both orientations."""
        self.run_pairs(a, [b], outdir, part, nparts)

    def run_line(self, a, bs, outdir):
        """One plan line: block `a` against every block in `bs` (daligner <A> <B1> <B2> ...)."""
        for k in range(0, len(bs), self.GROUP):
            self.run_pairs(a, bs[k:k + self.GROUP], outdir)


def run_datander(block, outdir, k=12, w=4, h=35, e=.70, l=500, s=100, j=4, verbose=0, out="tan"):
    """scrub/datander.c:226-258 for one block: Match_Self + tan/<blk>.<blk>.las."""
    L = api.lib()
    api.set_globals(verbose=verbose, minover=2 * l)
    if L.damar_tandem_set_params(k, w, h, j):
        raise ValueError("Illegal combination of filter parameters")
    os.makedirs(os.path.join(outdir, out), exist_ok=True)
    with _cwd(outdir):
        spec = L.New_Align_Spec(e, s, block.db.freq, j, 1, 0, 0, 0)
        cnt = (api.c_int64 * 3)()
        L.damar_match_self(C.byref(block.db), block.upload(), spec, cnt)
        L.Write_Overlap_Buffer(spec, out.encode(), out.encode(), block.root.encode(), block.root.encode(),
                               block.last_read())
        L.Reset_Overlap_Buffer(spec)
        L.Free_Align_Spec(spec)
    return list(cnt)


def hpc_plan(nblocks):
    """The block-pair work list HPCdaligner emits (HPCdaligner.c:628-788): line i compares
    block i against blocks i, i-1, ..., 1."""
    return [(i, list(range(i, 0, -1))) for i in range(1, nblocks + 1)]


def las_stats(path):
    """(records, aligned bp = sum(aepos - abpos)) of a .las file."""
    import numpy as np
    with open(path, "rb") as f:
        raw = f.read()
    novl = int(np.frombuffer(raw, dtype="<i8", count=1)[0])
    tspace = int(np.frombuffer(raw, dtype="<i4", count=1, offset=8)[0])
    tbytes = 1 if tspace <= 125 else 2
    off, bp = 12, 0
    for _ in range(novl):
        rec = np.frombuffer(raw, dtype="<i4", count=10, offset=off)
        bp += int(rec[4]) - int(rec[2])
        off += 40 + tbytes * int(rec[0])
    assert off == len(raw), (off, len(raw))
    return novl, bp

#!/usr/bin/env python3
"""bench.py -- aligned base-pairs/sec of the daligner block-vs-block hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole HPCdaligner block-pair plan (every block against itself and all
lower blocks, both orientations) of ONE database:

  N == 1  BASELINE.json config 2: `simulator 27 -c20 -e.15 -r2` (~50k PacBio-style reads, 540 Mbp),
          DBsplit -s135 -> 4 blocks, 10 block pairs;
  N  > 1  BASELINE.json config 3: `simulator 4.6 -c87 -e.15 -r3` (~400 Mbp), DBsplit -s25 -> 17 blocks,
          153 block pairs, handed out to the N ranks (one process per GPU) from a shared cursor in the
          job's torch.distributed store (damar_amd/multi.py): strong scaling, no collective in the data
          path; RCCL only for the barriers and the final max/sum.  (--config 2|3 overrides the choice;
          with config 2 on many GPUs the pairs are split by B-read range.)

All read blocks (forward and reverse-complemented bases) are resident in HBM on every rank before the
timed region; the timed region covers k-mer index builds, seed merge + sort, band filter +
Local_Alignment waves, the device->host copy of the alignments, the host tail and the sorted .las files
written to tmpfs.  After the timed region every .las of the last step is checked against the md5 of the
file the reference daligner wrote for the same database (tests/golden/config{2,3}_ref_md5.txt).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      dominant kernel, HIP-event timed on the library's stream; PMC-derived fractions from profiles/
  cpu_baseline  (N == 1) the compiled reference daligner (oracle/_ref) on the host cores over the WHOLE plan
  end_to_end    (N == 1) the contract's wall of SURVEY 8(d), "DB load -> last .las closed": the plan through
                damar_amd/bin/daligner -P from cold (process start, DB read from tmpfs, PCIe, index builds)
  one_gpu_same_workload (N > 1) rank 0 alone on the same database after the timed region: the measured
                strong-scaling speedup, independent of the driver's N = 1 run on config 2
  trace_expand  (N == 1, not part of `value`) SURVEY 8(f)4, Compute_Trace_PTS on the block-1 file.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CONFIGS = {
    2: dict(genome=27.0, coverage=20.0, block=135, seed=2, md5="config2_ref_md5.txt",
            text="config 2: simulator 27 -c20 -e.15 -r2, DBsplit -s135"),
    3: dict(genome=4.6, coverage=87.0, block=25, seed=3, md5="config3_ref_md5.txt",
            text="config 3: simulator 4.6 -c87 -e.15 -r3, DBsplit -s25"),
}


def pow2_floor(n):
    p = 1
    while 2 * p <= n:
        p *= 2
    return p


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def kernel_src_sha16():
    """sha256 (16 hex digits) over the sources the device code is built from: ties a counters file under profiles/ to the
    kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "damar_amd", "csrc")
    names = sorted(os.listdir(os.path.join(d, "kernels")))
    for f in [os.path.join(d, "kernels", n) for n in names if n.endswith((".hip", ".h"))] + [os.path.join(d, "shim.hip")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_counters():
    """(counters of the newest profiles/rNN_counters.json, stale?): the PMC-derived fields of `roofline` cannot be measured
    inside this process (rocprofv3 --pmc passes, scripts/gpu_profile_round.sh); they are carried from the file only while its
    `kernel_src_sha16` equals the sha of the kernel sources of THIS tree, else they are reported as null with pmc_stale."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")):
        m = re.match(r"r(\d+)_counters\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return {}, True
    try:
        pmc = json.load(open(best[1]))
    except Exception:
        return {}, True
    pmc["file"] = os.path.relpath(best[1], ROOT)
    return pmc, pmc.get("kernel_src_sha16") != kernel_src_sha16()


def sort_bytes_of(x):
    n, p = 0, 1
    while p < x:
        p <<= 8
        n += 1
    return n


def pipeline_bytes(kmers, bases, maxlen, nreads, pairs, builds, seeds, aligned_bp, trace_vals, k=14):
    """SURVEY 8(d) algorithmic bytes of one pass over a plan (16-byte records, 8-bit digits, one read + one write per pass),
    for the work actually done: `builds` index builds of blocks with kmers[i] k-mers and bases[i] bases, the block pairs
    `pairs` (each: two comparisons, forward and complement), `seeds` seed pairs in all, the records written.
      index build   1 B/base + 16 B x (1 + 2 P_k) per k-mer, P_k = ceil(2k/8)
      merge         16 (K_a + K_b) x 2 (count + emit) + 16 H         seed sort   16 H x 2 P_s
      filter        16 H                                              align       2 L_aln + 2 T"""
    nb = len(kmers)
    pk = (2 * k + 7) // 8
    per_build = sum(bases[i] + 16. * kmers[i] * (1 + 2 * pk) for i in range(nb)) / max(nb, 1)
    idx = per_build * builds
    merge = sum(2 * 32. * (kmers[a] + kmers[b]) for a, b in pairs) + 16. * seeds
    ps = sort_bytes_of(max(maxlen)) + 2 * sort_bytes_of(max(nreads))       # filter.c:2561-2580
    ssort = 16. * seeds * 2 * ps
    filt = 16. * seeds
    align = 2. * aligned_bp + 2. * trace_vals
    return {"index_build": idx, "merge": merge, "seed_sort": ssort, "filter": filt, "align": align,
            "total": idx + merge + ssort + filt + align, "seed_sort_passes": ps}


def check_against_reference(out_dir, md5_name):
    """{files, identical, missing}: every .las of out_dir against the reference's md5 fixture."""
    path = os.path.join(ROOT, "tests", "golden", md5_name)
    if not os.path.exists(path):
        return None
    bad, n = [], 0
    for ln in open(path):
        if ln.startswith("#"):
            continue
        m, rel = ln.split()
        n += 1
        f = os.path.join(out_dir, rel)
        if not os.path.exists(f) or md5_file(f) != m:
            bad.append(rel)
    return {"files": n, "identical": not bad, "differing": bad[:5],
            "against": "md5 of the reference daligner's files (tests/golden/%s)" % md5_name}


def link_db(dbdir, root, dst):
    os.makedirs(dst, exist_ok=True)
    for f in ("%s.db" % root, ".%s.idx" % root, ".%s.bps" % root):
        os.symlink(os.path.join(dbdir, f), os.path.join(dst, f))


def plan_text(root, nblocks, opts="-k14 -j16"):
    return "".join("daligner %s %s.%d %s\n" % (opts, root, a, " ".join("%s.%d" % (root, b) for b in range(a, 0, -1)))
                   for a in range(1, nblocks + 1))


def sum_las(out_dir):
    from damar_amd import driver
    nrec = bp = tv = 0
    for dp, _, fs in os.walk(out_dir):
        if "_parts" in dp:
            continue
        for f in fs:
            if f.endswith(".las"):
                n, b = driver.las_stats(os.path.join(dp, f))
                nrec += n
                bp += b
                tv += os.path.getsize(os.path.join(dp, f)) - 12 - 40 * n
    return nrec, bp, tv


def cpu_baseline(dbdir, root, nblocks, aligned_bp):
    """The reference daligner (oracle/_ref, compiled from /root/reference in the build container) over the
    WHOLE plan on this host's cores: the plan lines one after the other as a cluster job script would run them on one
    node (-j16), and all lines at once with -j16 and with -j32 (filter.c:767 allocas NTHREADS^2 x 2 KB: 2 MB at -j32 still fits
    the default stack, -j64 does not); the best of the three is the baseline."""
    cores = host_cores()
    ref = os.path.join(ROOT, "oracle", "_ref", "daligner")
    kind, exe = "reference", ref
    runs = [("sequential", pow2_floor(min(cores, 16))), ("concurrent", pow2_floor(min(cores, 16)))]
    if cores >= 32 * nblocks:
        runs.append(("concurrent", 32))
    if not os.path.exists(ref):
        kind, exe, runs = "port", os.path.join(ROOT, "oracle", "oracle_daligner"), [("concurrent", 1)]
    lines = [["%s.%d" % (root, a)] + ["%s.%d" % (root, b) for b in range(a, 0, -1)] for a in range(1, nblocks + 1)]
    # one process per BLOCK PAIR, all at once (what `HPCdaligner -B1` would emit): the plan spread over more cores than its
    # few lines can use; same output files
    pairs = [["%s.%d" % (root, a), "%s.%d" % (root, b)] for a in range(1, nblocks + 1) for b in range(a, 0, -1)]
    if kind == "reference" and cores >= 4 * len(pairs):
        runs.append(("per-pair", pow2_floor(min(16, cores // len(pairs)))))
    res = {}
    for mode, nthr in runs:
        work = tempfile.mkdtemp(prefix="damar_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            link_db(dbdir, root, work)
            t0 = time.time()
            if mode == "sequential":
                for ln in lines:
                    subprocess.run([exe, "-k14", "-j%d" % nthr] + ln, cwd=work, check=True, stdout=subprocess.DEVNULL)
            else:
                ps = [subprocess.Popen([exe, "-k14", "-j%d" % nthr] + ln, cwd=work, stdout=subprocess.DEVNULL)
                      for ln in (pairs if mode == "per-pair" else lines)]
                if any(p.wait() != 0 for p in ps):
                    raise RuntimeError("reference daligner failed")
            res["%s -j%d" % (mode, nthr)] = time.time() - t0
        finally:
            shutil.rmtree(work, ignore_errors=True)
    bestk = min(res, key=res.get)
    best = res[bestk]
    nthr = int(bestk.split("-j")[1])
    used = nthr * (len(pairs) if bestk.startswith("per-pair") else len(lines) if bestk.startswith("concurrent") else 1)
    return {"value": aligned_bp / best, "unit": "aligned bp/s", "cores": min(used, cores), "kind": kind,
            "sample": "the whole plan of the step (%d lines, %d block pairs, every .las), daligner -k14: %s; best: %s; host has %d cores"
                      % (len(lines), nblocks * (nblocks + 1) // 2,
                         ", ".join("%s lines %.1f s" % (m, s) for m, s in sorted(res.items())), bestk, cores),
            "wall_s": res}


def end_to_end(dbdir, root, nblocks, md5_name, repeats=3, gpus=1):
    """SURVEY 8(d)'s wall: DB on tmpfs -> last .las closed, through the C driver in plan mode (one process per GPU,
    cold: HIP start, block reads, reverse complements, PCIe, every index build); on several GPUs `daligner -P plan -G<n>`,
    the node scheduler of host/daligner.c.  Best of `repeats`; on one GPU run before this
    process puts its own blocks into HBM (`value` is filled in once the step has said how many bp were aligned)."""
    exe = os.path.join(ROOT, "damar_amd", "bin", "daligner")
    best, chk, node_stats = None, None, None
    for _ in range(repeats):
        work = tempfile.mkdtemp(prefix="damar_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            link_db(dbdir, root, work)
            with open(os.path.join(work, "plan.txt"), "w") as f:
                f.write(plan_text(root, nblocks))
            t0 = time.time()
            env = dict(os.environ)
            if os.environ.get("DAMAR_BENCH_SHARE_GPU"):
                env["DAMAR_SHARE_GPU"] = "1"              # (one-GPU rehearsal)
            if gpus > 1:
                env["DAMAR_PLAN_STATS"] = os.path.join(work, "node_stats.json")      # what every worker did (host/daligner.c node_main)
            subprocess.run([exe, "-P", "plan.txt"] + (["-G%d" % gpus] if gpus > 1 else []), cwd=work, check=True,
                           stdout=subprocess.DEVNULL, env=env)
            dt = time.time() - t0
            if best is None or dt < best:
                best = dt
                if gpus > 1:
                    try:
                        node_stats = json.loads(open(os.path.join(work, "node_stats.json")).read())
                    except Exception:
                        node_stats = None
            if chk is None:
                chk = check_against_reference(work, md5_name)
            time.sleep(0.6)                               # a repeat is a COLD run: not beside the previous worker's teardown
        finally:
            shutil.rmtree(work, ignore_errors=True)
    extra = {}
    if gpus == 1:
        # The command returns when the last .las is closed; its forked worker then still releases HBM and tears the HIP
        # context down (host/daligner.c: plan_main).  Two things show what that early return is worth and that it costs the
        # NEXT command nothing: the same plan with DAMAR_PLAN_TIDY=1 (one process that releases everything before it
        # exits), and two commands back to back (the second starts while the first one's worker is still tearing down).
        def run_once(env_extra, n):
            work = tempfile.mkdtemp(prefix="damar_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            try:
                link_db(dbdir, root, work)
                with open(os.path.join(work, "plan.txt"), "w") as f:
                    f.write(plan_text(root, nblocks))
                env = dict(os.environ, **env_extra)
                t0 = time.time()
                for _ in range(n):
                    subprocess.run([exe, "-P", "plan.txt"], cwd=work, check=True, stdout=subprocess.DEVNULL, env=env)
                dt = time.time() - t0
                ok = check_against_reference(work, md5_name)["identical"]
                return dt, ok
            finally:
                shutil.rmtree(work, ignore_errors=True)
        time.sleep(1.0)                                   # (the last repeat's worker is gone)
        try:
            dt, ok = run_once({"DAMAR_PLAN_TIDY": "1"}, 1)
            extra["tidy_wall_s"] = dt
            time.sleep(0.5)
            dt, ok2 = run_once({}, 2)
            extra["back_to_back_wall_s"] = dt / 2
            extra["back_to_back_identical"] = bool(ok and ok2)
        except Exception as e:
            extra["back_to_back_error"] = str(e)
    out = {"value": None, "unit": "aligned bp/s", "wall_s": best,
           "what": "damar_amd/bin/daligner -P <HPCdaligner plan>%s: process start, DB read from tmpfs, complement, "
                   "upload, index builds, all block pairs, sorted .las on tmpfs (best of %d cold runs; tidy_wall_s = the same "
                   "with the worker's teardown inside the command, back_to_back_wall_s = wall per command of two commands in a row: "
                   "the second waits at the teardown gate, host/damar_gate.h, until the first one's worker has left the GPU)"
                   % (" -G%d (one forked worker per GPU, regions + stealing)" % gpus if gpus > 1 else "", repeats),
           "identical_to_reference": None if chk is None else chk["identical"]}
    out.update(extra)
    if node_stats is not None:
        out["node"] = node_stats
    return out


def trace_expand_leg(dbdir, root, out_dir, with_cpu):
    """Not part of `value`: the next consumer of the records (Compute_Trace_PTS) on the block-1
    self-comparison .las the timed step has just written -- trace points -> edit scripts with
    damar_trace_pts (bin/lastrace -v reports the HIP-event times), next to the reference's own
    Compute_Trace_PTS on one host thread (oracle/_ref/ref_lastrace, or the oracle's port) on the same file,
    outputs compared byte for byte."""
    import re
    name = "%s.1" % root
    las = os.path.join(out_dir, "d001_00001", "%s.%s.las" % (name, name))
    blk = os.path.join(dbdir, name)
    gout = os.path.join(dbdir, "trace_gpu.bin")
    tool = os.path.join(ROOT, "damar_amd", "bin", "lastrace")
    best = None
    txt = subprocess.run([tool, "-v", "-R3", blk, blk, las, gout], check=True, stdout=subprocess.PIPE, text=True).stdout
    for m in re.finditer(r"(\d+) records, (\d+) segments \((\d+) deferred\), (\d+) script values; waves ([\d.]+) ms, "
                         r"device ([\d.]+) ms, call ([\d.]+) ms", txt):     # 3 calls in one process: the first allocates
        cur = dict(records=int(m.group(1)), segments=int(m.group(2)), deferred=int(m.group(3)),
                   script_values=int(m.group(4)), kernel_ms=float(m.group(5)), device_ms=float(m.group(6)),
                   call_ms=float(m.group(7)))
        if best is None or cur["device_ms"] < best["device_ms"]:
            best = cur
    best["workload"] = "block 1 self-comparison of the step, mode GREEDIEST"
    best["segments_per_s"] = best["segments"] / (best["device_ms"] * 1e-3)
    best["note"] = ("kernel_ms = trace_waves_slots, device_ms = all kernels of the call (HIP events), call_ms adds the "
                    "PCIe copies of points and scripts; not included in `value`")
    if with_cpu:
        ref = os.path.join(ROOT, "oracle", "_ref", "ref_lastrace")
        kind, exe = ("reference", ref) if os.path.exists(ref) else ("port", os.path.join(ROOT, "oracle", "oracle_lastrace"))
        cout = os.path.join(dbdir, "trace_cpu.bin")
        t0 = time.time()
        subprocess.run([exe, os.path.join(dbdir, root), las, cout, "0"], check=True)
        dt = time.time() - t0
        same = subprocess.run(["cmp", "-s", gout, cout]).returncode == 0
        best["cpu"] = {"value": best["segments"] / dt, "unit": "segments/s", "cores": 1, "kind": kind,
                       "sample": "the same file, %.1f s wall including opening the DB" % dt}
        best["identical_to_cpu"] = same
        os.unlink(cout)
    os.unlink(gout)
    return best


def plan_leg(api, driver, multi, L, base, title, sim_kw, j, md5_name, md5_tag=None, cpu_line=None):
    """One extra single-GPU measurement over a whole HPCdaligner plan (not part of `value`): database from its seed, blocks
    to HBM (untimed), one warm-up pass and one timed pass of every block pair through the same queue and runner as the
    headline, the md5 fixtures of the reference checked on the files they name, the pipeline's algorithmic bytes against
    the HBM peak, and (cpu_line) the reference daligner timed on ONE plan line of the same database."""
    work = tempfile.mkdtemp(prefix="damar_leg_", dir=base)
    try:
        t0 = time.time()
        nb = api.sim_write_db(work, "SIM", **sim_kw)
        t_gen = time.time() - t0
        dbprefix = os.path.join(work, "SIM")
        blocks = {}
        for i in range(1, nb + 1):
            b = driver.Block("%s.%d" % (dbprefix, i))
            b.upload()
            b.upload_complement()
            blocks[b.name] = b
        units = multi.work_units(nb, 1)

        def one(tag):
            out = os.path.join(work, "out_%s" % tag)
            runner = multi.GpuRunner(dict(j=j), resident=blocks)
            multi.run_queue(dbprefix, units, out, multi.LocalQueue(len(units)), runner)
            runner.finish()
            return out, runner.plan
        out, _ = one("w")
        shutil.rmtree(out, ignore_errors=True)
        L.damar_hip_sync()
        t0 = time.time()
        out, plan = one("s")
        L.damar_hip_sync()
        wall = time.time() - t0
        nrec, bp, tv = sum_las(out)
        want = {}
        for ln in open(os.path.join(ROOT, "tests", "golden", md5_name)):
            if ln.startswith("#"):
                continue
            f = ln.split()
            if md5_tag is None:
                want[f[1]] = f[0]
            elif f[1] == md5_tag:
                want[f[4]] = f[0]
        bad = [rel for rel, m in sorted(want.items()) if not os.path.exists(os.path.join(out, rel)) or md5_file(os.path.join(out, rel)) != m]
        order = sorted(blocks.values(), key=lambda b: b.db.part)
        kmers = [b.db.totlen - 14 * b.db.nreads for b in order]
        pairs = [(a, b) for a in range(nb) for b in range(a + 1)]
        pb = pipeline_bytes(kmers, [b.db.totlen for b in order], [b.db.maxlen for b in order], [b.db.nreads for b in order],
                            pairs, plan.index_builds, plan.counts[0], bp, tv)
        res = {"workload": title, "blocks": nb, "block_pairs": len(pairs), "comparisons": plan.matches,
               "index_builds": plan.index_builds, "wall_s": wall, "ms_per_block_pair": 1e3 * wall / len(pairs),
               "records": nrec, "aligned_bp": bp, "value": bp / wall, "unit": "aligned bp/s",
               "seed_pairs": plan.counts[0], "local_alignments": plan.counts[1],
               "phase_ms": {k: round(v, 1) for k, v in sorted(plan.timings.items())},
               "parity": {"files": len(want), "identical": not bad, "differing": bad[:5],
                          "against": "md5 of the reference daligner's files (tests/golden/%s%s)" % (md5_name, ", sample '%s'" % md5_tag if md5_tag else "")},
               "pipeline": {"algorithmic_bytes": pb["total"], "achieved": pb["total"] / wall / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": pb["total"] / wall / 1e9 / HBM_PEAK_GBS,
                            "stages_bytes": {k: v for k, v in pb.items() if k not in ("total", "seed_sort_passes")}},
               "db_generation_s": t_gen}
        ref = os.path.join(ROOT, "oracle", "_ref", "daligner")
        if cpu_line and os.path.exists(ref):
            cw = tempfile.mkdtemp(prefix="damar_legcpu_", dir=base)
            try:
                link_db(work, "SIM", cw)
                nthr = pow2_floor(min(host_cores(), 16))
                t0 = time.time()
                subprocess.run([ref, "-k14", "-j%d" % nthr] + ["SIM.%d" % x for x in cpu_line], cwd=cw, check=True, stdout=subprocess.DEVNULL)
                dt = time.time() - t0
                npl = len(cpu_line) - 1
                res["cpu"] = {"kind": "reference", "cores": nthr, "wall_s": dt, "ms_per_block_pair": 1e3 * dt / npl,
                              "sample": "daligner -k14 -j%d %s (%d block pair(s) of this plan, one process as the reference runs "
                                        "a plan line)" % (nthr, " ".join("SIM.%d" % x for x in cpu_line), npl)}
            finally:
                shutil.rmtree(cw, ignore_errors=True)
        for b in blocks.values():
            b.close()
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


def datander_leg(api, driver, L, base, tandem_frac):
    """BASELINE config 5: datander (scrub/tandem.c) on ALL FOUR blocks of the config-2 database with tandem arrays implanted
    into a fraction of the reads (plain simulator reads hold no tandem seeds).  Both sides timed process to process -- one
    cold `datander -j16 SIM.1 SIM.2 SIM.3 SIM.4` command each (scrub/datander.c:226-258 takes any number of blocks; the
    reference also as four commands at once, the better of the two counts) -- every tan/*.las md5-checked against the
    reference's; `resident` = one block through the in-process driver with the block already in HBM."""
    work = tempfile.mkdtemp(prefix="damar_tan_", dir=base)
    try:
        nb = api.sim_write_db(work, "SIM", 27., coverage=20., seed=2, block_mbp=135, tandem_frac=tandem_frac)
        want = {}
        for ln in open(os.path.join(ROOT, "tests", "golden", "config5_ref_md5.txt")):
            if not ln.startswith("#"):
                m, tag, blk, novl = ln.split()
                if tag == ("tandem" if tandem_frac else "plain"):
                    want[int(blk)] = (m, int(novl))
        blocks = ["SIM.%d" % i for i in range(1, nb + 1)]
        # process to process, this side
        exe = os.path.join(ROOT, "damar_amd", "bin", "datander")
        best = None
        for _ in range(3):
            shutil.rmtree(os.path.join(work, "tan"), ignore_errors=True)
            t0 = time.time()
            subprocess.run([exe, "-j16"] + blocks, cwd=work, check=True, stdout=subprocess.DEVNULL)
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
            time.sleep(0.5)
        n = bp = 0
        same = True
        for i in range(1, nb + 1):
            las = os.path.join(work, "tan", "SIM.%d.SIM.%d.las" % (i, i))
            ni, bpi = driver.las_stats(las)
            n += ni
            bp += bpi
            same = same and i in want and md5_file(las) == want[i][0]
        res = {"workload": "config 5: datander -k12 -w4 -h35 -e.70 -l500 on the %d blocks (135 Mbp each) of the config-2 database, "
                           "tandem arrays implanted into %.0f %% of the reads; one cold command over all blocks, process start to "
                           "exit (best of 3)" % (nb, 100 * tandem_frac),
               "blocks": nb, "records": n, "aligned_bp": bp, "wall_s": best, "value": bp / best, "unit": "aligned bp/s",
               "identical_to_reference": same}
        # one block resident in HBM, in-process
        b = driver.Block(os.path.join(work, "SIM.1"))
        b.upload()
        rbest = None
        for _ in range(2):                       # (the first call sizes the scratch)
            shutil.rmtree(os.path.join(work, "tan1"), ignore_errors=True)
            os.makedirs(os.path.join(work, "tan1"))
            L.damar_hip_sync()
            t0 = time.time()
            driver.run_datander(b, os.path.join(work, "tan1"), j=8)
            L.damar_hip_sync()
            dt = time.time() - t0
            rbest = dt if rbest is None else min(rbest, dt)
        n1, bp1 = driver.las_stats(os.path.join(work, "tan1", "tan", "SIM.1.SIM.1.las"))
        res["resident"] = {"what": "block 1 already in HBM: index build, self-matching, band filter + waves, host tail, tan/*.las on tmpfs",
                           "records": n1, "aligned_bp": bp1, "wall_s": rbest, "value": bp1 / rbest}
        b.close()
        ref = os.path.join(ROOT, "oracle", "_ref", "datander")
        if os.path.exists(ref):
            nthr = pow2_floor(min(host_cores(), 16))
            walls = {}
            for mode in ("one command", "%d commands at once" % nb):
                cw = tempfile.mkdtemp(prefix="damar_tancpu_", dir=base)
                try:
                    link_db(work, "SIM", cw)
                    t0 = time.time()
                    if mode == "one command":
                        subprocess.run([ref, "-j%d" % nthr] + blocks, cwd=cw, check=True, stdout=subprocess.DEVNULL)
                    else:
                        ps = [subprocess.Popen([ref, "-j%d" % nthr, blk], cwd=cw, stdout=subprocess.DEVNULL) for blk in blocks]
                        if any(p.wait() != 0 for p in ps):
                            raise RuntimeError("reference datander failed")
                    walls[mode] = time.time() - t0
                finally:
                    shutil.rmtree(cw, ignore_errors=True)
            bk = min(walls, key=walls.get)
            res["cpu"] = {"kind": "reference", "cores": nthr * (nb if bk != "one command" else 1), "wall_s": walls[bk], "value": bp / walls[bk],
                          "sample": "datander -j%d over the same %d blocks, process start and DB read included: %s; best: %s"
                                    % (nthr, nb, ", ".join("%s %.2f s" % kv for kv in sorted(walls.items())), bk)}
            res["vs_cpu"] = walls[bk] / best
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


def visible_gpus():
    """GPUs this process could use, counted WITHOUT the HIP / HSA runtime: the kfd topology nodes that have SIMDs
    (/sys/class/kfd/kfd/topology/nodes/*/properties, as host/daligner.c gpu_numa_node_sysfs does), cut down by the
    *_VISIBLE_DEVICES lists the runtime would honour.  (torch.cuda.device_count() is hipGetDeviceCount on a torch without
    amdsmi: the runtime would be up in the process that is about to start the ranks -- ADVICE r5.)"""
    import glob
    n = 0
    for f in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
        try:
            props = dict(ln.split()[:2] for ln in open(f) if len(ln.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def profiler_preloaded(env=None):
    """A profiler / tool library that initialises the GPU runtime before main() (rocprofv3 does): a process started under it
    must not start GPU children of its own (the same rule as damar_profiler_preloaded() in host/daligner.c)."""
    env = os.environ if env is None else env
    if env.get("ROCP_TOOL_LIBRARIES") or env.get("HSA_TOOLS_LIB") or env.get("ROCPROFILER_LIBRARY_PATH"):
        return True
    return any(k in env.get("LD_PRELOAD", "") for k in ("rocprof", "roctracer", "rocprofiler"))


def relaunch_under_torchrun(ngpus, argv=None, run=subprocess.run):
    """`bench.py --gpus N` started plainly: run the same command as N ranks, one per GPU, and hand back its exit code (the
    child's rank 0 prints the JSON line on the stdout it inherits).  Fails loudly when fewer than N GPUs are visible --
    except with DAMAR_BENCH_SHARE_GPU=1, the one-GPU rehearsal in which the ranks share GPU 0 (gloo instead of RCCL)."""
    import socket
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    if profiler_preloaded(env):
        raise SystemExit("bench.py: --gpus %d under a preloaded profiler: this process's GPU runtime is already up and must not "
                         "start the ranks.  Profile `python -m torch.distributed.run ... bench.py --gpus %d` started from outside, "
                         "or one rank (--gpus 1)" % (ngpus, ngpus))
    have = visible_gpus()
    if have < ngpus:
        if not env.get("DAMAR_BENCH_SHARE_GPU"):
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible (DAMAR_BENCH_SHARE_GPU=1 rehearses the ranks on "
                             "one GPU)" % (ngpus, have))
        env.setdefault("DAMAR_BENCH_BACKEND", "gloo")
    with socket.socket() as sk:                      # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="one rank per GPU (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=5)       # (2 until round 4: a step is 0.25 s, and two steps are mostly the pipeline filling and draining)
    ap.add_argument("--warmup", type=int, default=2)      # (the second warm-up step allocates the second pinned landing buffer of the host pipeline)
    ap.add_argument("--config", type=int, default=0, help="BASELINE config of the database: 2 or 3 (default: 2 on one GPU, 3 on several)")
    ap.add_argument("--threads-param", type=int, default=16, help="daligner -j (slice rule only; the reference md5s were made with -j16)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-trace", action="store_true", help="skip the trace-expansion leg (SURVEY 8(f)4)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end leg")
    ap.add_argument("--sync-steps", action="store_true", help="barrier + device sync around EVERY step (rounds 1-2), not around the K steps")
    ap.add_argument("--no-legs", action="store_true", help="skip the config-4 (lead 24 blocks), config-3 and config-5 legs")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()

    if args.gpus is None:                 # `torchrun --nproc-per-node N bench.py`: the launcher says how many
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks as CHILDREN (one process per GPU under
        # torch.distributed.run, the same command the contract names) before this process has touched the GPU -- it never does
        sys.exit(relaunch_under_torchrun(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): one rank per GPU, "
                         "`--nproc-per-node` must equal --gpus" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("DAMAR_BENCH_BACKEND", "nccl")     # "gloo" only to rehearse on one GPU
        if os.environ.get("DAMAR_BENCH_SHARE_GPU"):
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    cfg_no = args.config or (2 if world == 1 else 3)
    cfg = CONFIGS[cfg_no]

    from damar_amd import api, driver, multi
    L = api.lib()
    L.damar_hip_init(local if world > 1 else int(os.environ.get("DAMAR_DEVICE", "0")))

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync_all():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        L.damar_hip_sync()

    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    # one work directory for the job: every rank reads the same DB files and writes its pairs' .las there
    work = os.path.join(base, "damar_bench_%s" % (os.environ.get("MASTER_PORT", str(os.getpid())) if world > 1 else os.getpid()))
    try:
        # ---- untimed: synthetic DB (rank 0), blocks to HBM (every rank) ----
        t_gen = 0.
        if rank == 0:
            shutil.rmtree(work, ignore_errors=True)
            os.makedirs(work)
            t0 = time.time()
            api.sim_write_db(work, "SIM", cfg["genome"], coverage=cfg["coverage"], seed=cfg["seed"], block_mbp=cfg["block"])
            t_gen = time.time() - t0
        barrier()
        nblocks = int(open(os.path.join(work, "SIM.db")).read().split("blocks =")[1].split()[0])
        dbprefix = os.path.join(work, "SIM")
        e2e = None
        if world == 1 and not args.no_e2e:
            try:
                e2e = end_to_end(work, "SIM", nblocks, cfg["md5"])
            except Exception as e:
                e2e = {"error": str(e)}
        blocks = {}
        for i in range(1, nblocks + 1):
            b = driver.Block("%s.%d" % (dbprefix, i))
            b.upload()
            b.upload_complement()
            blocks[b.name] = b
        totbp = sum(b.db.totlen for b in blocks.values())
        nreads = sum(b.db.nreads for b in blocks.values())
        units = multi.work_units(nblocks, world)
        npairs = nblocks * (nblocks + 1) // 2
        store = multi.default_store() if dist is not None else None

        def one_step(tag, my_units=None, queue=None):
            out = os.path.join(work, "out_%s" % tag)
            us = units if my_units is None else my_units
            runner = multi.GpuRunner(dict(j=args.threads_param), resident=blocks)
            if queue is None:
                queue = multi.make_queue(store, tag, us, rank)
            mine = multi.run_queue(dbprefix, us, out, queue, runner)
            runner.finish()          # drains the asynchronous host tail: every .las of this rank is closed
            if any(n > 1 for _, _, _, n in us):
                barrier()
                multi.merge_parts(dbprefix, us, out, rank, world)
            return out, runner.plan, len(mine)

        # Split pairs need every rank's parts closed before they are merged: such a job (config 2 on many GPUs) is timed
        # step by step.  Otherwise the K steps run BACK TO BACK as the consecutive plans of one long job -- one runner, no
        # drain between them, so that the last report launch of a step runs beside the first seed stages of the next, as it
        # does between the lines of any real plan -- bracketed as a whole by barrier + device sync.  Every step still builds
        # its own indexes (end_pass) and writes its own files (two output directories used in turn; the last step's are
        # md5-checked).  --sync-steps gives the step-by-step timing of earlier rounds; `ms_per_step_synced` reports it beside.
        split = any(n > 1 for _, _, _, n in units)
        pipelined = not split and not args.sync_steps

        rank_stats = {"busy_s": 0., "stolen": 0}       # this rank's own time inside the timed region (until ITS last file is closed) and
                                                        # the units it took out of other ranks' regions: the SCALE record explains itself

        def run_steps(tag, nsteps, synced):
            """(elapsed, last_out, plans, units_run): nsteps passes; synced: each on its own between syncs"""
            el, outs, plans, nrun = 0., None, [], 0
            if synced:
                for s in range(nsteps):
                    if outs and rank == 0:
                        shutil.rmtree(outs, ignore_errors=True)
                    sync_all()
                    t0 = time.time()
                    outs, plan, n = one_step("%s%d" % (tag, s))
                    if tag == "s":
                        rank_stats["busy_s"] += time.time() - t0
                    sync_all()
                    el += time.time() - t0
                    plans.append(plan)
                    nrun += n
                return el, outs, plans, nrun
            runner = multi.GpuRunner(dict(j=args.threads_param), resident=blocks)
            sync_all()
            t0 = time.time()
            for s in range(nsteps):
                outs = os.path.join(work, "out_%s%d" % (tag, s & 1))
                queue = multi.make_queue(store, "%s%d" % (tag, s), units, rank)
                nrun += len(multi.run_queue(dbprefix, units, outs, queue, runner))
                if tag == "s":
                    rank_stats["stolen"] += getattr(queue, "stolen", 0)
                runner.end_pass()
            runner.finish()
            if tag == "s":
                rank_stats["busy_s"] = time.time() - t0
            sync_all()
            return time.time() - t0, outs, [runner.plan], nrun

        if args.warmup > 0:
            _, wout, _, _ = run_steps("w", args.warmup, not pipelined)
            barrier()
            if rank == 0:
                for d in {wout, os.path.join(work, "out_w0"), os.path.join(work, "out_w1")}:
                    shutil.rmtree(d, ignore_errors=True)

        tim, cnts, last_out, nmine, builds, nmatch, nlaunch = {}, [0, 0, 0], None, 0, 0, 0, 0
        wave = [0, 0, 0]              # band cells, wave steps per alignment pass, wave-loop iterations (counted by the kernel)
        elapsed, last_out, plans, nmine = run_steps("s", args.steps, not pipelined)
        for plan in plans:
            wave = [x + y for x, y in zip(wave, plan.wave)]
            builds += plan.index_builds
            nmatch += plan.matches
            nlaunch += plan.report_launches
            for k, v in plan.timings.items():
                tim[k] = tim.get(k, 0.) + v
            cnts = [c + d for c, d in zip(cnts, plan.counts)]
        synced_ms = None
        if pipelined and world == 1 and args.steps > 0:
            # the same measurement with every step on its own between syncs (the edge of a step exposed), 3 steps
            es, so, _, _ = run_steps("y", 3, True)
            synced_ms = 1e3 * es / 3
            if rank == 0:
                shutil.rmtree(so, ignore_errors=True)

        tkeys = ["report", "ssort", "ksort", "merge", "tuples", "table", "work", "d2h", "tail", "write"]
        if dist is not None:
            import torch
            rdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            v = torch.tensor([float(x) for x in wave] + [tim.get(k, 0.) for k in tkeys] + [float(c) for c in cnts] +
                             [float(nmatch), float(nlaunch), float(nmine), float(builds)], device=rdev, dtype=torch.float64)
            vmax = v.clone()
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
            vals = [float(x) for x in v.tolist()]
            wave, vals = vals[:3], vals[3:]
            tim = dict(zip(tkeys, vals[:len(tkeys)]))
            cnts = vals[len(tkeys):len(tkeys) + 3]
            nmatch, nlaunch, units_run, builds = vals[-4], vals[-3], vals[-2], vals[-1]
            units_max = float(vmax[-2].item())
        else:
            units_run, units_max = float(nmine), float(nmine)
        per_rank = None
        if dist is not None:
            # one row per rank, gathered as it is (no reduction): units, stolen units, index builds, busy seconds, report ms,
            # seed-side ms (tuples + k-mer sorts + merge + seed sorts + work list), host tail + write ms (thread time)
            import torch
            mine = torch.tensor([float(nmine), float(rank_stats["stolen"]), float(sum(p.index_builds for p in plans)),
                                 float(rank_stats["busy_s"]), float(sum(p.timings.get("report", 0.) for p in plans)),
                                 float(sum(p.timings.get(k, 0.) for p in plans for k in ("tuples", "ksort", "merge", "ssort", "work"))),
                                 float(sum(p.timings.get(k, 0.) for p in plans for k in ("tail", "write")))],
                                device=rdev, dtype=torch.float64)
            rows = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(rows, mine)
            per_rank = [dict(zip(("units", "stolen", "index_builds", "busy_s", "report_ms", "seed_side_ms", "tail_write_ms"),
                                 [round(float(x), 3) for x in r.tolist()])) for r in rows]

        one_gpu = None
        if world > 1:
            # the same database on ONE GPU (rank 0, the others wait): the measured strong-scaling reference
            barrier()
            if rank == 0:
                us1 = multi.work_units(nblocks, 1)
                o1, _, _ = one_step("one_w", us1, multi.LocalQueue(len(us1)))
                shutil.rmtree(o1, ignore_errors=True)
                L.damar_hip_sync()
                t1 = time.time()
                o1, _, _ = one_step("one", us1, multi.LocalQueue(len(us1)))
                L.damar_hip_sync()
                one_gpu = time.time() - t1
                shutil.rmtree(o1, ignore_errors=True)
            barrier()

        if world > 1 and not args.no_e2e:
            # the contract's wall on N GPUs: the C driver's node mode (rank 0 starts it, the other ranks wait; the
            # workers it forks share the GPUs with the idle ranks of this job)
            barrier()
            if rank == 0:
                try:
                    e2e = end_to_end(work, "SIM", nblocks, cfg["md5"], repeats=2, gpus=world)
                except Exception as e:
                    e2e = {"error": str(e)}
            barrier()
        if rank == 0:
            steps = max(1, args.steps)
            nrec, bp, trace_vals = sum_las(last_out)
            parity = check_against_reference(last_out, cfg["md5"])
            nmatch = float(nmatch) / steps                   # comparisons (block pair x orientation) per step, all ranks
            nlaunch = float(nlaunch) / steps                 # launches of the report kernel they took
            kern = {"report2_kernel": tim.get("report", 0.),
                    "onesweep_pass<u64> (seed-pair radix sort)": tim.get("ssort", 0.),
                    "onesweep_pass<u64> (k-mer index radix sort)": tim.get("ksort", 0.),
                    "merge_sweep + merge_emit": tim.get("merge", 0.)}
            what = {"report2_kernel": "band filter + Local_Alignment waves, two read pairs per wavefront (kernels/report_packed.h)"}
            dom = max(kern, key=kern.get)
            H = cnts[0] / steps                      # seed pairs per step
            nsplit = max(n for _, _, _, n in units)
            ngroup = max(len(b) if isinstance(b, tuple) else 1 for _, b, _, _ in units)
            order = sorted(blocks.values(), key=lambda b: b.db.part)
            kmers_b = [b.db.totlen - 14 * b.db.nreads for b in order]
            pb = pipeline_bytes(kmers_b, [b.db.totlen for b in order], [b.db.maxlen for b in order], [b.db.nreads for b in order],
                                [(a, b) for a in range(nblocks) for b in range(a + 1)], builds / steps, H, bp, trace_vals)
            if dom.startswith("report"):
                # filter 16 B/seed + align 2 B per aligned bp + 2 B per trace value
                alg = pb["filter"] + pb["align"]
                nl = nlaunch
            elif "seed-pair" in dom:
                alg = pb["seed_sort"]
                nl = nmatch
            elif "k-mer" in dom:
                alg = pb["index_build"]
                nl = builds / steps
            else:
                alg = pb["merge"]
                nl = nmatch
            dom_ms = kern[dom] / steps               # summed over ranks: the kernel's total device time per step
            ach = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.
            # Counter-derived figures cannot be read from inside this process: they come from the separate rocprofv3 --pmc passes
            # over this very command (scripts/gpu_profile_round.sh), committed under profiles/ per round and carried here
            # ONLY while that file's kernel_src_sha16 matches the kernel sources of this tree
            pmc, stale = load_counters()
            mine = (not stale) and dom.startswith(pmc.get("kernel_symbol", "\0"))

            def carried(key):
                return pmc.get(key) if mine else None
            step_s = elapsed / steps
            roof = {"bound": "hbm", "kernel": dom, "kernel_what": what.get(dom), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": carried("bytes_per_launch"),
                    "launches_per_step": nl, "avg_launch_ms": dom_ms / nl if nl else 0.,
                    "algorithmic_bytes_per_step": alg,
                    "valu_frac": carried("valu_frac"), "salu_frac": carried("salu_frac"),
                    "valu_busy_weighted": carried("valu_busy_weighted"),
                    "active_lane_frac": carried("active_lane_frac"),
                    "lds_bank_conflict_frac": carried("lds_bank_conflict_frac"),
                    # SURVEY 8(d)'s secondary unit of K6 (the reference's WAVE_STATS, align.c:81, 353-368), counted by the
                    # kernel itself in this run: diagonals computed summed over all wave steps, per second of kernel time
                    "band_cells_per_step": wave[0] / steps,
                    "band_cells_per_s": (wave[0] / steps) / (dom_ms * 1e-3) if dom.startswith("report") and dom_ms > 0 else None,
                    "wave_steps_per_step": wave[1] / steps,
                    "halves_per_iteration": wave[1] / wave[2] if wave[2] else None,
                    "pmc_source": pmc.get("source") if mine else None, "pmc_file": pmc.get("file"), "pmc_head": pmc.get("head"),
                    "pmc_stale": not mine, "kernel_src_sha16": kernel_src_sha16(),
                    "pipeline": {"what": "SURVEY 8(d) algorithmic bytes of ALL stages of a step (index builds actually done, merge, seed "
                                         "sort, filter, align) / wall time of the step",
                                 "algorithmic_bytes_per_step": pb["total"], "achieved": pb["total"] / step_s / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": pb["total"] / step_s / 1e9 / HBM_PEAK_GBS,
                                 "stages_bytes": {k: v for k, v in pb.items() if k not in ("total", "seed_sort_passes")}},
                    "note": "integer/branchy wave kernel: its limit is instruction issue and dependent latency, not HBM "
                            "(valu_frac / salu_frac = share of the calibrated issue peaks of the cheapest instructions, valu_busy_weighted "
                            "= vector-pipe time with every instruction kind at its measured cost, profiles/); phase ms per step "
                            "(summed over ranks): " + ", ".join("%s=%.1f" % (k, v / steps) for k, v in sorted(tim.items()))}
            value = bp * args.steps / elapsed
            cpu = trace = None
            if world == 1 and not args.no_cpu:
                try:
                    cpu = cpu_baseline(work, "SIM", nblocks, bp)
                except Exception as e:           # the baseline is reported, never required
                    cpu = {"value": None, "unit": "aligned bp/s", "cores": 0, "kind": "reference", "sample": "failed: %s" % e}
            if e2e is not None and e2e.get("wall_s"):
                e2e["value"] = bp / e2e["wall_s"]
                if cpu and cpu.get("value"):
                    e2e["vs_cpu_whole_plan"] = e2e["value"] / cpu["value"]
            if world == 1 and not args.no_trace:
                try:
                    trace = trace_expand_leg(work, "SIM", last_out, not args.no_cpu)
                except Exception as e:
                    trace = {"error": str(e)}
            legs = None
            if world == 1 and not args.no_legs:
                for b in blocks.values():            # give the headline's blocks back before the other databases come
                    b.close()
                blocks.clear()
                legs = {}
                for key, fn in (
                        ("config4_lead", lambda: plan_leg(
                            api, driver, multi, L, base,
                            "config 4, first 24 of its 255 blocks: simulator 248 -c80 -m15000 -s3000 -e.15 -r4, DBsplit -s78; the whole "
                            "HPCdaligner plan of those blocks (300 block pairs x 2 orientations), daligner -k14 -j8",
                            dict(genome_mbp=248., coverage=80., seed=4, rmean=15000, rsdev=3000, block_mbp=78, max_blocks=24), 8,
                            "config4_ref_md5.txt", "lead", cpu_line=None if args.no_cpu else (24, 2))),
                        ("config3", lambda: plan_leg(
                            api, driver, multi, L, base,
                            "config 3: simulator 4.6 -c87 -e.15 -r3, DBsplit -s25 -> 17 blocks, 153 block pairs x 2 orientations, "
                            "daligner -k14 -j16", dict(genome_mbp=4.6, coverage=87., seed=3, block_mbp=25), 16,
                            "config3_ref_md5.txt", None, cpu_line=None if args.no_cpu else (17, 17, 16, 15, 14))),
                        ("config5_datander", lambda: datander_leg(api, driver, L, base, .3))):
                    try:
                        legs[key] = fn()
                    except Exception as e:       # a leg is reported, never required
                        legs[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            contract = None
            if e2e is not None and e2e.get("wall_s"):
                # like for like with the reference, which is timed to process exit: the command with its worker's teardown
                # inside (DAMAR_PLAN_TIDY=1); the moment the default command RETURNS (every .las closed, the forked worker
                # still leaving the GPU behind the caller) is `returns_after_s`
                cw = e2e.get("tidy_wall_s") or e2e["wall_s"]
                cpu_v = cpu.get("value") if cpu else None
                contract = {"what": "SURVEY 8(d)'s wall for the same plan: DB on tmpfs -> last .las closed AND the GPU released, one "
                                    "cold `daligner -P` command to process exit (process start, block reads, complement, PCIe, every "
                                    "index build, every block pair, teardown); returns_after_s = when the default command hands "
                                    "control back (all .las closed, its worker still tearing down), back_to_back_wall_s = wall per "
                                    "command of two default commands in a row",
                            "value": bp / cw, "unit": "aligned bp/s", "wall_s": cw,
                            "returns_after_s": e2e["wall_s"],
                            "back_to_back_wall_s": e2e.get("back_to_back_wall_s"), "tidy_wall_s": e2e.get("tidy_wall_s"),
                            "vs_cpu": (bp / cw) / cpu_v if cpu_v else None,
                            "vs_cpu_at_return": e2e.get("vs_cpu_whole_plan")}
                # the driver's record keeps `config` and `roofline` verbatim: the contract's scalars ride there too
                for dst in (roof,):
                    dst["contract_wall_s"] = contract["wall_s"]
                    dst["contract_value"] = contract["value"]
                    dst["contract_vs_cpu"] = contract["vs_cpu"]
                    dst["contract_returns_after_s"] = contract["returns_after_s"]
                    dst["contract_back_to_back_wall_s"] = contract["back_to_back_wall_s"]
            line = {"metric": "aligned base-pairs/sec (daligner block-vs-block)",
                    "value": value, "unit": "aligned bp/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps,
                    "ms_per_step_synced": synced_ms,
                    "timing": (("the %d steps back to back between one barrier + device sync on either side (one job, no drain "
                                "between steps; every step builds its own indexes and writes its own files)" % steps) if pipelined else
                               "every step on its own between barrier + device sync; the step times added up") +
                              "; `value` has the blocks already in HBM when the clock starts (block read, reverse complement and "
                              "upload are NOT in it) -- the wall from the DB files on is `contract`",
                    "contract": contract,
                    "higher_is_better": True, "scaling": "strong",
                    "vs_baseline": None,       # the reference publishes no number for this metric (vs the CPU: contract.vs_cpu)
                    "dtype": "int32", "data": "synthetic",
                    "config": {"workload": "%s -> %d blocks, %d block pairs x 2 orientations per step, daligner -k14 -w6 -h35 "
                                           "-e.70 -l1000 -s100" % (cfg["text"], nblocks, npairs),
                               "reads": nreads, "bases": totbp,
                               "records_per_step": nrec, "aligned_bp_per_step": bp,
                               "seed_pairs_per_step": H, "local_alignments_per_step": cnts[1] / steps,
                               "index_builds_per_step": builds / steps,
                               "comparisons_per_step": nmatch, "report_launches_per_step": nlaunch,
                               "parallelism": "%d GPU(s), one process each, ONE database; %d work units per step (%s) "
                                              "pulled from cursors in the job's store (one region of the plan per rank, the others' "
                                              "leftovers after it), no data-path collective; busiest rank ran %d units"
                                              % (world, len(units),
                                                 "block pairs split %d-way by B-read range" % nsplit if nsplit > 1 else
                                                 "one A block against up to %d subject blocks, both orientations; one report "
                                                 "launch per subject block, in flight beside the next block's index builds and "
                                                 "seed stages" % ngroup, int(units_max)),
                               "db_generation_s": t_gen,
                               "per_rank": per_rank,
                               "busy_max_over_mean": (max(r["busy_s"] for r in per_rank) * len(per_rank) /
                                                      max(1e-9, sum(r["busy_s"] for r in per_rank))) if per_rank else None,
                               "contract_wall_s": contract["wall_s"] if contract else None,
                               "contract_value": contract["value"] if contract else None,
                               "contract_vs_cpu": contract["vs_cpu"] if contract else None,
                               "contract_returns_after_s": contract["returns_after_s"] if contract else None,
                               "contract_back_to_back_wall_s": contract["back_to_back_wall_s"] if contract else None},
                    "parity": parity,
                    "roofline": roof, "cpu_baseline": cpu, "end_to_end": e2e, "trace_expand": trace, "legs": legs}
            if one_gpu is not None:
                line["one_gpu_same_workload"] = {"ms_per_step": 1e3 * one_gpu, "value": bp / one_gpu,
                                                 "speedup_of_this_run": (bp * args.steps / elapsed) / (bp / one_gpu)}
                # the N = 1 line of a driver's scaling run is config 2: the like-for-like reference of THIS line is here
                line["scale_ref"] = {"value": bp / one_gpu, "unit": "aligned bp/s", "n_gpus": 1,
                                     "what": "the same database and plan on ONE GPU of this job (rank 0 alone, after the timed region)"}
            print(json.dumps(line))
            sys.stdout.flush()
        barrier()
    finally:
        if dist is not None:
            try:
                dist.barrier()
            except Exception:
                pass
        if not args.keep and rank == 0:
            shutil.rmtree(work, ignore_errors=True)
        if dist is not None:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()

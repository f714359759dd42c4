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


def check_against_reference(out_dir, md5_name):
    """{files, identical, missing}: every .las of out_dir against the reference's md5 fixture."""
    path = os.path.join(ROOT, "tests", "golden", md5_name)
    if not os.path.exists(path):
        return None
    bad, n = [], 0
    for ln in open(path):
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
    WHOLE plan on this host's cores, -j16 (more threads overflow its alloca, filter.c:767): the plan lines
    one after the other as a cluster job script would run them on one node, and all lines at once."""
    cores = host_cores()
    ref = os.path.join(ROOT, "oracle", "_ref", "daligner")
    kind, nthr, exe = "reference", pow2_floor(min(cores, 16)), ref
    if not os.path.exists(ref):
        kind, nthr, exe = "port", 1, os.path.join(ROOT, "oracle", "oracle_daligner")
    lines = [["%s.%d" % (root, a)] + ["%s.%d" % (root, b) for b in range(a, 0, -1)] for a in range(1, nblocks + 1)]
    res = {}
    for mode in (("concurrent",) if kind == "port" else ("sequential", "concurrent")):
        work = tempfile.mkdtemp(prefix="damar_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            link_db(dbdir, root, work)
            t0 = time.time()
            if mode == "sequential":
                for ln in lines:
                    subprocess.run([exe, "-k14", "-j%d" % nthr] + ln, cwd=work, check=True, stdout=subprocess.DEVNULL)
            else:
                ps = [subprocess.Popen([exe, "-k14", "-j%d" % nthr] + ln, cwd=work, stdout=subprocess.DEVNULL) for ln in lines]
                if any(p.wait() != 0 for p in ps):
                    raise RuntimeError("reference daligner failed")
            res[mode] = time.time() - t0
        finally:
            shutil.rmtree(work, ignore_errors=True)
    best = min(res.values())
    used = nthr * (len(lines) if res.get("concurrent") == best else 1)
    return {"value": aligned_bp / best, "unit": "aligned bp/s", "cores": min(used, cores), "kind": kind,
            "sample": "the whole plan of the step (%d lines, %d block pairs, every .las), daligner -k14 -j%d: %s; host has %d cores"
                      % (len(lines), nblocks * (nblocks + 1) // 2, nthr,
                         ", ".join("%s lines %.1f s" % (m, s) for m, s in sorted(res.items())), cores),
            "wall_s": res}


def end_to_end(dbdir, root, nblocks, md5_name, repeats=3):
    """SURVEY 8(d)'s wall: DB on tmpfs -> last .las closed, through the C driver in plan mode (one process,
    cold: HIP start, block reads, reverse complements, PCIe, every index build).  Best of `repeats`; run before this
    process puts its own blocks into HBM (`value` is filled in once the step has said how many bp were aligned)."""
    exe = os.path.join(ROOT, "damar_amd", "bin", "daligner")
    best, chk = None, None
    for _ in range(repeats):
        work = tempfile.mkdtemp(prefix="damar_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            link_db(dbdir, root, work)
            with open(os.path.join(work, "plan.txt"), "w") as f:
                f.write(plan_text(root, nblocks))
            t0 = time.time()
            subprocess.run([exe, "-P", "plan.txt"], cwd=work, check=True, stdout=subprocess.DEVNULL)
            dt = time.time() - t0
            if best is None or dt < best:
                best = dt
            if chk is None:
                chk = check_against_reference(work, md5_name)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return {"value": None, "unit": "aligned bp/s", "wall_s": best,
            "what": "damar_amd/bin/daligner -P <HPCdaligner plan>: process start, DB read from tmpfs, complement, "
                    "upload, index builds, all block pairs, sorted .las on tmpfs (best of %d cold runs)" % repeats,
            "identical_to_reference": None if chk is None else chk["identical"]}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)      # (the second warm-up step allocates the second pinned landing buffer of the host pipeline)
    ap.add_argument("--config", type=int, default=0, help="BASELINE config of the database: 2 or 3 (default: 2 on one GPU, 3 on several)")
    ap.add_argument("--threads-param", type=int, default=16, help="daligner -j (slice rule only; the reference md5s were made with -j16)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-trace", action="store_true", help="skip the trace-expansion leg (SURVEY 8(f)4)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end leg")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

        for w in range(args.warmup):
            out, _, _ = one_step("w%d" % w)
            barrier()
            if rank == 0:
                shutil.rmtree(out, ignore_errors=True)

        # Every step is bracketed by barrier + device sync on both sides and the K step times are added up; between two
        # steps (outside the clock) rank 0 deletes the previous step's output files, which are the harness's, not the path's.
        tim, cnts, last_out, nmine, builds, nmatch, nlaunch = {}, [0, 0, 0], None, 0, 0, 0, 0
        elapsed = 0.
        for s in range(args.steps):
            if last_out and rank == 0:
                shutil.rmtree(last_out, ignore_errors=True)
            sync_all()
            t0 = time.time()
            last_out, plan, n = one_step("s%d" % s)
            sync_all()
            elapsed += time.time() - t0
            nmine += n
            builds += plan.index_builds
            nmatch += plan.matches
            nlaunch += plan.report_launches
            for k, v in plan.timings.items():
                tim[k] = tim.get(k, 0.) + v
            cnts = [c + d for c, d in zip(cnts, plan.counts)]

        tkeys = ["report", "ssort", "ksort", "merge", "tuples", "table", "work", "d2h", "tail", "write"]
        if dist is not None:
            import torch
            rdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            v = torch.tensor([tim.get(k, 0.) for k in tkeys] + [float(c) for c in cnts] +
                             [float(nmatch), float(nlaunch), float(nmine), float(builds)], device=rdev, dtype=torch.float64)
            vmax = v.clone()
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
            vals = [float(x) for x in v.tolist()]
            tim = dict(zip(tkeys, vals[:len(tkeys)]))
            cnts = vals[len(tkeys):len(tkeys) + 3]
            nmatch, nlaunch, units_run, builds = vals[-4], vals[-3], vals[-2], vals[-1]
            units_max = float(vmax[-2].item())
        else:
            units_run, units_max = float(nmine), float(nmine)

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

        if rank == 0:
            steps = max(1, args.steps)
            nrec, bp, trace_vals = sum_las(last_out)
            parity = check_against_reference(last_out, cfg["md5"])
            nmatch = float(nmatch) / steps                   # comparisons (block pair x orientation) per step, all ranks
            nlaunch = float(nlaunch) / steps                 # launches of the report kernel they took
            kern = {"report_kernel (band filter + Local_Alignment waves)": tim.get("report", 0.),
                    "radix sort of seed pairs (hist+scan+scatter, u64 keys)": tim.get("ssort", 0.),
                    "radix sort of the k-mer index (hist+scan+scatter, u32 keys)": tim.get("ksort", 0.),
                    "seed merge (count+scan+emit)": tim.get("merge", 0.)}
            dom = max(kern, key=kern.get)
            H = cnts[0] / steps                      # seed pairs per step
            nsplit = max(n for _, _, _, n in units)
            ngroup = max(len(b) if isinstance(b, tuple) else 1 for _, b, _, _ in units)
            if dom.startswith("report"):
                # filter 16 B/seed + align 2 B per aligned bp + 2 B per trace value
                alg = 16. * H + 2. * bp + 2. * trace_vals
                nl = nlaunch
            elif dom.startswith("radix sort of seed"):
                alg = 16. * H * 2 * 6                # 16-byte records, read+write, P_s = 6 passes
                nl = nmatch
            elif dom.startswith("radix sort of the k-mer"):
                kmers = totbp - 14 * nreads
                alg = (16. * 2 * 4) * kmers / nblocks * builds / steps   # per build: 16 B x (rd+wr) x 4 passes per k-mer
                nl = builds / steps
            else:
                kmers = totbp - 14 * nreads
                alg = 32. * 2 * kmers / nblocks * nmatch + 16. * H
                nl = nmatch
            dom_ms = kern[dom] / steps               # summed over ranks: the kernel's total device time per step
            ach = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.
            # Counter-derived figures cannot be read from inside this process: they come from the separate
            # rocprofv3 --pmc passes over this very command (scripts/gpu_profile_round.sh), committed under
            # profiles/ per round and carried here
            pmc = {}
            for name in ("r02_counters.json", "r01_traffic.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                    break
                except Exception:
                    continue
            traffic = pmc.get("bytes_per_launch") if dom.startswith(pmc.get("kernel", "\0")) else None
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "launches_per_step": nl, "avg_launch_ms": dom_ms / nl if nl else 0.,
                    "algorithmic_bytes_per_step": alg,
                    "valu_frac": pmc.get("valu_frac"), "salu_frac": pmc.get("salu_frac"),
                    "valu_busy_weighted": pmc.get("valu_busy_weighted"),
                    "active_lane_frac": pmc.get("active_lane_frac"), "pmc_source": pmc.get("source"),
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
            line = {"metric": "aligned base-pairs/sec (daligner block-vs-block)",
                    "value": value, "unit": "aligned bp/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps,
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
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
                               "db_generation_s": t_gen},
                    "parity": parity,
                    "roofline": roof, "cpu_baseline": cpu, "end_to_end": e2e, "trace_expand": trace}
            if one_gpu is not None:
                line["one_gpu_same_workload"] = {"ms_per_step": 1e3 * one_gpu, "value": bp / one_gpu,
                                                 "speedup_of_this_run": (bp * args.steps / elapsed) / (bp / one_gpu)}
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

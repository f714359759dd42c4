#!/usr/bin/env python3
"""profiles/rNN_counters.json from one run of scripts/gpu_profile_round.sh (gpurun_out/round/): what bench.py cannot
measure from inside its own process -- the TCC traffic of the dominant kernel per launch (FETCH_SIZE doubled as the
gfx950 note of MI355X_MICROARCH.md prescribes for wide reads, an upper bound for this kernel's narrow ones; WRITE_SIZE
as is) and its instruction issue as a share of the CALIBRATED peaks of tools/roofcal.hip (profiles/r02_roofcal.txt:
0.456 VALU wave-instructions per cycle per SIMD and 0.953 SALU instructions per cycle per CU at 8 waves per SIMD,
nominal 2.4 GHz).

  python scripts/make_counters_json.py gpurun_out/round profiles/r02_counters.json [active_lane_frac]
"""
import csv
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

VALU_PEAK = 0.456 * 2.4e9 * 1024       # wave-instructions per second, whole chip (roofcal, 8 waves/SIMD): the CHEAPEST kind
# `roofcal ops` (profiles/r02_roofcal_ops.txt): only v_add / v_sub / v_mov_b32 / and / or / xor / lshr / ashr issue that fast,
# every other vector instruction costs 4.4 cycles of SIMD time at 5 wavefronts per SIMD; the wave loop's mix (scripts/
# valu_weight.py on the compiler's assembly: 45 % cheap) averages 3.62 cycles per instruction
VALU_WEIGHTED_CYCLES = 3.66        # (round 6 loop: scripts/hotpath.py on the common path, 337 cycles over 92 instructions; 3.62 for the round 2-5 loops)
SALU_PEAK = 0.953 * 2.4e9 * 256        # instructions per second, whole chip


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lanes = float(sys.argv[3]) if len(sys.argv) > 3 else None
    # what the kernel counted itself in the plain bench run of the same round (bench.json: roofline.band_cells_per_step, ...)
    live = {}
    try:
        live = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])["roofline"]
    except Exception:
        pass
    iters = None
    if live.get("wave_steps_per_step") and live.get("halves_per_iteration"):
        iters = live["wave_steps_per_step"] / live["halves_per_iteration"]          # wave-loop iterations per step
        if lanes is None and live.get("band_cells_per_step"):
            lanes = live["band_cells_per_step"] / (iters * 64.)
    stats = list(csv.DictReader(open(src + "/stats/r_kernel_stats.csv")))
    dom = max(stats, key=lambda r: float(r["TotalDurationNs"]))
    name = dom["Name"].split("(")[0]
    dur_s = float(dom["TotalDurationNs"]) * 1e-9
    calls = int(dom["Calls"])
    pmc = open(src + "/pmc_summary.txt").read()
    m = re.search(r"^\s+%s\s+(\{.*\})$" % re.escape(name), pmc, re.M)
    cnt = {}
    for mm in re.finditer(r"^\s+%s\s+(\{.*\})$" % re.escape(name), pmc, re.M):
        for k, v in eval(mm.group(1)).items():
            cnt[k] = float(v.split()[0])
    tr = {}
    for ln in open(src + "/traffic_summary.txt"):
        mm = re.match(r"\('(\w+)', '(\w+)'\) ([\d.e+]+) (\d+) launches", ln)
        if mm and mm.group(1) == name:
            tr[mm.group(2)] = float(mm.group(3)) * 1024.      # KB -> bytes
    out = {"kernel": "report_kernel" if name.startswith("report") else name, "kernel_symbol": name,
           "launches": calls, "avg_launch_ms": 1e3 * dur_s / calls,
           "bytes_per_launch": (2 * tr["FETCH_SIZE"] + tr["WRITE_SIZE"]) / calls if "FETCH_SIZE" in tr and "WRITE_SIZE" in tr else None,
           "fetch_bytes_per_launch_x2": 2 * tr.get("FETCH_SIZE", 0) / calls, "write_bytes_per_launch": tr.get("WRITE_SIZE", 0) / calls,
           "valu_frac": cnt["SQ_INSTS_VALU"] / dur_s / VALU_PEAK if "SQ_INSTS_VALU" in cnt else None,
           "salu_frac": cnt["SQ_INSTS_SALU"] / dur_s / SALU_PEAK if "SQ_INSTS_SALU" in cnt else None,
           "valu_busy_weighted": cnt["SQ_INSTS_VALU"] * VALU_WEIGHTED_CYCLES / (dur_s * 2.4e9 * 1024) if "SQ_INSTS_VALU" in cnt else None,
           "active_lane_frac": lanes,
           "halves_per_iteration": live.get("halves_per_iteration"), "band_cells_per_step": live.get("band_cells_per_step"),
           "wave_loop_iterations_per_step": iters,
           "lds_bank_conflict_frac": cnt["SQ_LDS_BANK_CONFLICT"] / cnt["SQ_LDS_IDX_ACTIVE"] if cnt.get("SQ_LDS_IDX_ACTIVE") else None,
           "wave_cycles_share": {k: cnt[k] / cnt["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")
                                 if k in cnt and cnt.get("SQ_WAVE_CYCLES")},
           "counters_of_the_run": cnt,            # sums over the `launches` launches of the profiled command (4 passes over the plan)
           "counters_per_launch": {k: v / calls for k, v in cnt.items()},
           "source": "rocprofv3 --kernel-trace --stats and separate --pmc passes of `python3 bench.py --steps 1 --warmup 0 --no-cpu "
                     "--no-trace --no-e2e --no-legs` (scripts/gpu_profile_round.sh); issue peaks from tools/roofcal.hip (profiles/r02_roofcal.txt)"}
    if iters and live.get("launches_per_step"):
        out["per_iteration"] = {k.replace("SQ_INSTS_", "").lower(): v / calls * live["launches_per_step"] / iters
                                for k, v in cnt.items() if k.startswith("SQ_INSTS_")}
    # what ties this file to a tree: bench.py carries its figures only while the kernel sources still hash to this
    import bench
    out["kernel_src_sha16"] = bench.kernel_src_sha16()
    out["bench_py_sha16"] = __import__("hashlib").sha256(open(bench.__file__, "rb").read()).hexdigest()[:16]
    try:
        out["head"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=os.path.dirname(os.path.abspath(bench.__file__)),
                                     stdout=subprocess.PIPE, text=True).stdout.strip() or None
    except Exception:
        out["head"] = None
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("kernel_symbol", "avg_launch_ms", "bytes_per_launch", "valu_frac", "salu_frac", "valu_busy_weighted",
                                          "active_lane_frac", "halves_per_iteration", "per_iteration", "lds_bank_conflict_frac",
                                          "wave_cycles_share") if k in out}, indent=1))


if __name__ == "__main__":
    main()

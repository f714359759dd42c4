#!/bin/bash
# the whole GPU suite, then one bench run (phase times in the roofline note)
mkdir -p gpurun_out
( while true; do sleep 60; echo "[suite] still running $(date +%T)"; done ) &
KEEP=$!
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/suite.log 2>&1
rc=$?
kill $KEEP
tail -4 gpurun_out/suite.log
[ $rc = 0 ] || { tail -40 gpurun_out/suite.log; exit 1; }
DAMAR_HOSTPROF=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-trace $BENCH_ARGS > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || { tail -5 gpurun_out/quick_bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "launch ms", d["roofline"]["avg_launch_ms"], "launches", d["roofline"]["launches_per_step"], "parity", d["parity"]["identical"])
print(d["roofline"]["note"][-230:])
print("e2e", d["end_to_end"])
PY
grep "damar host wall\|scratch grow" gpurun_out/quick_bench.err | tail -6
if [ -n "$JUNK" ]; then timeout -k 10 300 python3 scripts/seed_junk.py 2>&1 | tail -6; fi

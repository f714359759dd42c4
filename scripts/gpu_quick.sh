#!/bin/bash
# quick GPU check: the parity tests of config 1/2 and one bench run (phase times in the roofline note)
mkdir -p gpurun_out
if [ -z "$NOTESTS" ]; then timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }; fi
tail -2 gpurun_out/quick_tests.log
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-trace --no-e2e $BENCH_ARGS > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || { tail -5 gpurun_out/quick_bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "launch ms", d["roofline"]["avg_launch_ms"], "parity", d["parity"])
print(d["roofline"]["note"][-230:])
PY

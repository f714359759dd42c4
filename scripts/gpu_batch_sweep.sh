#!/bin/bash
# report-kernel time against the number of comparisons per launch (DAMAR_BATCH)
mkdir -p gpurun_out
for f in ${SWEEP:-1 2 4 16}; do
  export DAMAR_BATCH=$f
  timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-trace --no-e2e > gpurun_out/batch_$f.json 2> gpurun_out/batch_$f.err || { echo "batch $f failed"; tail -5 gpurun_out/batch_$f.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/batch_$f.json").read().strip().splitlines()[-1])
print("batch $f", "%.1f ms/step" % d["ms_per_step"], "launches", d["roofline"]["launches_per_step"], "avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], d["parity"]["identical"], d["roofline"]["note"][-150:])
PY
done

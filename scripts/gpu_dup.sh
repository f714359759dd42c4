#!/bin/bash
# experiment: report-kernel time against the work per launch (DAMAR_DUP_WORK=f runs every item f times)
mkdir -p gpurun_out
for f in 0 2 3; do
  if [ $f = 0 ]; then unset DAMAR_DUP_WORK; else export DAMAR_DUP_WORK=$f; fi
  timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-trace --no-e2e > gpurun_out/dup_$f.json 2> gpurun_out/dup_$f.err || { echo "dup $f failed"; tail -5 gpurun_out/dup_$f.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/dup_$f.json").read().strip().splitlines()[-1])
print("dup $f", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["note"][-260:])
PY
done

#!/bin/bash
# overlapped report launches: step time against the report kernel's residency (DAMAR_SLOTS) and the comparisons per launch
mkdir -p gpurun_out
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-trace --no-e2e > gpurun_out/ov_$tag.json 2> gpurun_out/ov_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/ov_$tag.err; return; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/ov_$tag.json").read().strip().splitlines()[-1])
print("$tag", "%.1f ms/step" % d["ms_per_step"], d["parity"]["identical"], d["roofline"]["note"][-150:])
PY
}
if [ -n "$SWEEP2" ]; then
run on_b1_w4   DAMAR_OVERLAP=1 DAMAR_SLOTS=8192 DAMAR_BATCH=1
run on_b1_w5   DAMAR_OVERLAP=1 DAMAR_BATCH=1
run on_b2_w4   DAMAR_OVERLAP=1 DAMAR_SLOTS=8192 DAMAR_BATCH=2
run on_b3_w4   DAMAR_OVERLAP=1 DAMAR_SLOTS=8192 DAMAR_BATCH=3
run on_b2_w45  DAMAR_OVERLAP=1 DAMAR_SLOTS=9216 DAMAR_BATCH=2
run on_b2_w35  DAMAR_OVERLAP=1 DAMAR_SLOTS=7168 DAMAR_BATCH=2
exit 0
fi
run off        DAMAR_OVERLAP=0
run on_b4      DAMAR_OVERLAP=1
run on_b2      DAMAR_OVERLAP=1 DAMAR_BATCH=2
run on_b8      DAMAR_OVERLAP=1 DAMAR_BATCH=8
run on_b4_w4   DAMAR_OVERLAP=1 DAMAR_SLOTS=8192
run on_b4_w3   DAMAR_OVERLAP=1 DAMAR_SLOTS=6144
run on_b2_w4   DAMAR_OVERLAP=1 DAMAR_SLOTS=8192 DAMAR_BATCH=2

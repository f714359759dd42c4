#!/bin/bash
# radix tile of 2048 items (8 per thread, ~64 VGPRs) against 4096 (16 per thread, 128 VGPRs): alone and beside the report kernel
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" > gpurun_out/rs_$tag.json 2> gpurun_out/rs_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/rs_$tag.err; return; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/rs_$tag.json").read().strip().splitlines()[-1])
print("$tag", "%.1f ms/step" % d["ms_per_step"], d["parity"]["identical"], d["roofline"]["note"][-150:])
PY
}
A="--steps 3 --warmup 1 --no-cpu --no-trace --no-e2e"
run t16_alone DAMAR_OVERLAP=0 python3 bench.py $A
run t8_alone  DAMAR_OVERLAP=0 python3 scripts/bench_lib.py build/rs8 $A
run t16_over  python3 bench.py $A
run t8_over   python3 scripts/bench_lib.py build/rs8 $A

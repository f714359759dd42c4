#!/bin/bash
# radix tiles of 1024 / 2048 items (4 / 8 per thread) against 4096 (16 per thread, 128 VGPRs): alone and beside the report kernel
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" > gpurun_out/rs_$tag.json 2> gpurun_out/rs_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/rs_$tag.err; return; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/rs_$tag.json").read().strip().splitlines()[-1])
print("$tag", "%.1f ms/step" % d["ms_per_step"], d["parity"]["identical"], d["roofline"]["note"][-150:])
PY
}
A="--steps 3 --warmup 1 --no-cpu --no-trace --no-e2e"
for r in ${ROUNDS:-4 8}; do
run t${r}_alone  DAMAR_OVERLAP=0 python3 scripts/bench_lib.py build/rs$r $A
run t${r}_over   python3 scripts/bench_lib.py build/rs$r $A
done

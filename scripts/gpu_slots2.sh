#!/bin/bash
# overlapped default: resident report wavefronts per SIMD (DAMAR_SLOTS = 2048 x w) once more, after the per-block batching
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu --no-trace --no-e2e > gpurun_out/s2_$tag.json 2> gpurun_out/s2_$tag.err || { echo "$tag failed"; return; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/s2_$tag.json").read().strip().splitlines()[-1])
print("$tag", "%.1f ms/step" % d["ms_per_step"], d["parity"]["identical"], d["roofline"]["note"][-150:])
PY
}
run w3    DAMAR_SLOTS=6144
run w35   DAMAR_SLOTS=7168
run w4    DAMAR_SLOTS=8192
run w45   DAMAR_SLOTS=9216
run w5    DAMAR_SLOTS=10240

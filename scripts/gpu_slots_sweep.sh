#!/bin/bash
# report-kernel time against the resident wavefronts per SIMD (DAMAR_SLOTS = 2048 x waves per SIMD for the packed kernel)
mkdir -p gpurun_out
for w in ${SWEEP:-1 2 3 4}; do
  export DAMAR_SLOTS=$((2048 * w))
  timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-trace --no-e2e > gpurun_out/slots_$w.json 2> gpurun_out/slots_$w.err || { echo "slots $w failed"; tail -5 gpurun_out/slots_$w.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/slots_$w.json").read().strip().splitlines()[-1])
print("waves/SIMD $w", "%.1f ms/step" % d["ms_per_step"], d["parity"]["identical"], d["roofline"]["note"][-150:])
PY
done

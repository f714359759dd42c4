#!/bin/bash
# prefix-table resolution against merge time (warm) and the cold plan run (end_to_end)
mkdir -p gpurun_out
for t in ${SWEEP:-28 26 24}; do
  export DAMAR_TBITS=$t
  timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-trace > gpurun_out/tbits_$t.json 2> gpurun_out/tbits_$t.err || { echo "tbits $t failed"; tail -5 gpurun_out/tbits_$t.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/tbits_$t.json").read().strip().splitlines()[-1])
print("tbits $t", "%.1f ms/step" % d["ms_per_step"], d["parity"]["identical"], "e2e %.3f s" % d["end_to_end"]["wall_s"], d["roofline"]["note"][-150:])
PY
done

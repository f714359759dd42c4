#!/bin/bash
# the tree's library, with and without the overlap, twice (noise)
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/ab_$tag.err; return; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
print("$tag", "%.1f ms/step" % d["ms_per_step"], d["parity"]["identical"], d["roofline"]["note"][-150:])
PY
}
A="--steps 3 --warmup 1 --no-cpu --no-trace --no-e2e"
run alone1 DAMAR_OVERLAP=0 python3 bench.py $A
run over1  python3 bench.py $A
run alone2 DAMAR_OVERLAP=0 python3 bench.py $A
run over2  python3 bench.py $A

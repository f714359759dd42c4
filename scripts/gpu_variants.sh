#!/bin/bash
# bench.py on the tree's library and on the variants under build/ named in VARIANTS
mkdir -p gpurun_out
run() { # name, command...
  name=$1; shift
  timeout -k 10 300 "$@" --steps 2 --warmup 1 --no-cpu --no-trace --no-e2e > gpurun_out/var_$name.json 2> gpurun_out/var_$name.err || { echo "$name failed"; tail -5 gpurun_out/var_$name.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/var_$name.json").read().strip().splitlines()[-1])
print("$name", "%.1f ms/step" % d["ms_per_step"], d["parity"]["identical"], d["roofline"]["note"][-150:])
PY
}
run tree python3 bench.py
for v in $VARIANTS; do run $v python3 scripts/bench_lib.py build/$v; done

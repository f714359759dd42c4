#!/bin/bash
# wall time of the bench legs (config 4 lead plan, config 3, datander) under environment variants:
# gpurun -- bash scripts/gpu_sweep_legs.sh "A=1" "B=2 C=3" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "X=1" "$@"; do
  echo "== $v"
  env $v timeout -k 10 400 python bench.py --steps 1 --warmup 1 --no-cpu --no-trace --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['legs'].items():
    print('  %-18s %.3f s  %s' % (k, v.get('wall_s', 0), ' '.join('%s=%.0f' % (a, b) for a, b in sorted((v.get('phase_ms') or {}).items()))))"
done 2>&1 | tee gpurun_out/sweep_legs.txt

#!/bin/bash
# bench step time under environment variants: gpurun -- bash scripts/gpu_sweep_env.sh "A=1" "B=2 C=3" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "X=1" "$@"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench.py --steps 3 --warmup 2 --no-cpu --no-trace --no-e2e --no-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms/step  synced %.1f  identical %s  %s' % (d['ms_per_step'], d['ms_per_step_synced'], d['parity']['identical'], d['roofline']['note'].split('ranks): ')[-1]))"
done 2>&1 | tee -a gpurun_out/sweep_env.txt

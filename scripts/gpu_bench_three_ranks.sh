#!/bin/bash
# rehearsal of `bench.py --gpus 3` on a one-GPU box: two ranks share the card (gloo for the barriers), config 3, one step
mkdir -p gpurun_out
( while true; do sleep 60; echo "[3 ranks] still running $(date +%T)"; done ) &
KEEP=$!
DAMAR_BENCH_SHARE_GPU=1 DAMAR_BENCH_BACKEND=gloo timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 \
  --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 3 --steps 1 --warmup 0 > gpurun_out/three_ranks.json 2> gpurun_out/three_ranks.err
rc=$?
kill $KEEP
tail -3 gpurun_out/three_ranks.err
python3 - <<PY
import json
d = json.loads(open("gpurun_out/three_ranks.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "ms/step %.0f" % d["ms_per_step"], "parity", d["parity"], "\n", d["config"]["parallelism"], "\n one gpu:", d.get("one_gpu_same_workload"))
PY
exit $rc

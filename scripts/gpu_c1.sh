#!/bin/bash
# Config-1 end-to-end parity on the GPU box: reference binary vs MI355X daligner.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$ROOT/gpurun_out/c1
rm -rf "$W" && mkdir -p "$W" && cd "$W"
$ROOT/damar_amd/bin/simdb . SIM ${GENOME:-0.5} -c20 -r${SEED:-1} -e.15 -S${BLOCK:-200}
echo "== reference (oracle/_ref/daligner -j4)"
( time $ROOT/oracle/_ref/daligner -k14 -j4 SIM.1 SIM.1 ) 2>&1 | tail -4
mv d001_00001 ref_d001_00001
echo "== MI355X"; export DAMAR_DEBUG=${DAMAR_DEBUG:-}
( time timeout -k 10 ${TMO:-300} $ROOT/damar_amd/bin/daligner -v -k14 -j4 SIM.1 SIM.1 ) > gpu.log 2>&1 || true; tail -50 gpu.log; if grep -q "Memory access fault" gpu.log; then echo GPU_FAULT; exit 9; fi
ls -la ref_d001_00001 d001_00001
md5sum ref_d001_00001/*.las d001_00001/*.las
cmp ref_d001_00001/SIM.1.SIM.1.las d001_00001/SIM.1.SIM.1.las && echo "LAS IDENTICAL"

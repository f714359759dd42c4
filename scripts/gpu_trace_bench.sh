#!/bin/bash
# Trace-point expansion at config-2 size: one block pair of config 2 through the MI355X
# daligner, then every record of its .las through damar_trace_pts (bin/lastrace -v prints the kernel and
# call times), next to the reference's Compute_Trace_PTS (oracle/_ref/ref_lastrace, one host thread) on the
# same file when it is small enough (SAMPLE=1 -> only the self pair of block 1).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${WORKDIR:-/dev/shm/damar_tr}
rm -rf "$W" && mkdir -p "$W" && cd "$W"
$ROOT/damar_amd/bin/simdb . SIM ${GENOME:-27} -c20 -r${SEED:-2} -e.15 -S${BLOCK:-135} > nblocks.txt
echo "blocks: $(cat nblocks.txt)"
A=${A:-1}; B=${B:-1}
timeout -k 10 300 $ROOT/damar_amd/bin/daligner -k14 -j16 SIM.$A SIM.$B > dal.log 2>&1 || { tail -5 dal.log; exit 9; }
LAS=$(ls d001_*/SIM.$A.SIM.$B.las | head -1); ls -la $LAS
for m in ${MODES:-0 1 -1}; do
  for rep in 1 2; do
    timeout -k 10 300 $ROOT/damar_amd/bin/lastrace -v -m$m SIM.$A SIM.$B $LAS gpu_$m.bin
  done
done
for rep in 1 2; do
  echo "Compute_Trace_MID:"; timeout -k 10 300 $ROOT/damar_amd/bin/lastrace -v -M SIM.$A SIM.$B $LAS gpu_mid.bin
done
if [ -n "$REF" ]; then
  t0=$(date +%s%N); $ROOT/oracle/_ref/ref_lastrace SIM $LAS ref_mid.bin 0 mid; t1=$(date +%s%N)
  echo "reference Compute_Trace_MID mode 0, 1 thread: $(( (t1 - t0) / 1000000 )) ms (DB open included)"
  cmp gpu_mid.bin ref_mid.bin && echo "MID: IDENTICAL to the reference"
fi
if [ -n "$REF" ]; then
  for m in ${MODES:-0}; do
    t0=$(date +%s%N); $ROOT/oracle/_ref/ref_lastrace SIM $LAS ref_$m.bin $m; t1=$(date +%s%N)
    echo "reference Compute_Trace_PTS mode $m, 1 thread: $(( (t1 - t0) / 1000000 )) ms (DB open included)"
    cmp gpu_$m.bin ref_$m.bin && echo "mode $m: IDENTICAL to the reference"
  done
fi
ls -la gpu_0.bin
rm -rf "$W"

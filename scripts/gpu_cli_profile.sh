#!/bin/bash
# Where the wall time of one plan line goes through the command-line driver (process start, DB read,
# upload, first-use allocations, kernels, host tail): config-3-like DB, line SIM.9 against SIM.9 .. SIM.1.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/dev/shm/damar_cli
rm -rf $W && mkdir -p $W && cd $W
$ROOT/damar_amd/bin/simdb . SIM ${GENOME:-4.6} -c${COV:-87} -r3 -e.15 -S${BLOCK:-25} > nblocks.txt
A=${A:-9}; bs=""; for b in $(seq $A -1 1); do bs="$bs SIM.$b"; done
for rep in 1 2; do
  t0=$(date +%s%N)
  DAMAR_HOSTPROF=1 DAMAR_CLIPROF=1 $ROOT/damar_amd/bin/daligner -k14 -j16 SIM.$A $bs > out.log 2> err.log
  t1=$(date +%s%N); echo "line SIM.$A x $A blocks: $(( (t1 - t0) / 1000000 )) ms"
  grep -i "damar host\|cli:\|scratch grow" err.log out.log | tail -30
done
rm -rf $W

#!/bin/bash
# Which allocations stall in the driver: every plan line of a config-3-like DB with DAMAR_HOSTPROF=1.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/dev/shm/damar_st
rm -rf $W && mkdir -p $W && cd $W
$ROOT/damar_amd/bin/simdb . SIM 4.6 -c87 -r3 -e.15 -S25 > nblocks.txt
NB=$(cat nblocks.txt)
for a in $(seq 1 $NB); do bs=""; for b in $(seq $a -1 1); do bs="$bs SIM.$b"; done
  t0=$(date +%s%N)
  DAMAR_HOSTPROF=1 $ROOT/damar_amd/bin/daligner -k14 -j16 SIM.$a $bs > out.log 2> err.log
  t1=$(date +%s%N); echo "line $a: $(( (t1 - t0) / 1000000 )) ms $(grep -c hipMalloc err.log) slow allocations"
  grep "hipMalloc" err.log
done
rm -rf $W

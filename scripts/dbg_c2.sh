#!/bin/bash
ROOT=$(pwd)
W=/dev/shm/dbg_c2; rm -rf $W; mkdir -p $W; cd $W
$ROOT/damar_amd/bin/simdb . SIM 27 -c20 -r2 -e.15 -S135 > /dev/null
ulimit -c 0
for line in "SIM.2 SIM.2 SIM.1" "SIM.3 SIM.3 SIM.2 SIM.1"; do
  timeout -k 10 200 $ROOT/damar_amd/bin/daligner -v -k14 -j16 $line > log.txt 2>&1
  echo "line [$line] rc=$?"
  tail -6 log.txt
done
rm -rf $W

#!/bin/bash
# datander on one config-2 block: GPU binary vs the reference binary (wall time, identical output)
set -e
ROOT=$(pwd)
W=$(mktemp -d /dev/shm/dtan.XXXX)
$ROOT/damar_amd/bin/simdb $W SIM 27 -c20 -r2 -e.15 -S135 > /dev/null
cd $W
mkdir g r
for d in g r; do for f in SIM.db .SIM.idx .SIM.bps; do ln -s ../$f $d/$f; done; done
( cd g; /usr/bin/env time -f "gpu datander %e s" $ROOT/damar_amd/bin/datander -j16 SIM.1 2>&1 | tail -2 ) || true
( cd g; s=$(date +%s.%N); $ROOT/damar_amd/bin/datander -j16 SIM.1 > /dev/null; e=$(date +%s.%N); echo "gpu datander wall $(python3 -c "print($e-$s)")" )
if [ -x $ROOT/oracle/_ref/datander ]; then ( cd r; s=$(date +%s.%N); $ROOT/oracle/_ref/datander -j16 SIM.1 > /dev/null; e=$(date +%s.%N); echo "ref datander -j16 wall $(python3 -c "print($e-$s)")" ); md5sum g/tan/*.las r/tan/*.las; fi
rm -rf $W

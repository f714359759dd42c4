#!/bin/bash
ROOT=$(pwd); W=$(mktemp -d /dev/shm/caps.XXXX)
for f in G.db .G.idx .G.bps; do cp tests/golden/tiny2/$f $W/; done
cd $W; DAMAR_TEST_SMALL_CAPS=1 $ROOT/damar_amd/bin/daligner -v -k14 -j4 G.1 G.1 2>&1 | tail -15
rm -rf $W

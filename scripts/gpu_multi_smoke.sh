#!/bin/bash
# damar_amd.multi on the GPU box with one rank (RCCL init, plan, LAmerge), against the golden files
set -e
ROOT=$(pwd)
W=$(mktemp -d /dev/shm/multi.XXXX)
for f in G.db .G.idx .G.bps; do cp $ROOT/tests/golden/tiny2/$f $W/; done
cd $ROOT
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 -m damar_amd.multi $W/G 2 $W 2>&1 | tail -3
bad=0
for f in $(cd tests/golden/tiny2/las && ls */*.las); do cmp -s tests/golden/tiny2/las/$f $W/$f || { echo "DIFF $f"; bad=1; }; done
md5sum $W/G.1.las $W/G.2.las; grep "tiny2.*-$" tests/golden/lamerge_ref_md5.txt
echo "bad=$bad"
rm -rf $W

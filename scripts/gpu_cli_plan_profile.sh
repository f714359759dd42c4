#!/bin/bash
# where the wall time of `daligner -P` on config 2 goes (cold process, DB on tmpfs)
R=$GRAFT_REPO_ROOT
W=$(mktemp -d /dev/shm/planprof.XXXX); cd $W
$R/damar_amd/bin/simdb . SIM 27 -c20 -r2 -e.15 -S135 > /dev/null
for a in 1 2 3 4; do bs=""; for b in $(seq $a -1 1); do bs="$bs SIM.$b"; done; echo "daligner -k14 -j16 SIM.$a $bs"; done > plan.txt
for i in 1 2; do
  rm -rf d001_*; s=$(date +%s.%N)
  DAMAR_CLIPROF=1 DAMAR_HOSTPROF=1 $R/damar_amd/bin/daligner -P plan.txt > out$i.txt 2>&1
  e=$(date +%s.%N); echo "run $i wall $(python3 -c "print('%.3f' % ($e-$s))") s"; grep -E "cli:|damar host|hipMalloc|scratch grow" out$i.txt | cut -c1-600
done
rm -rf $W

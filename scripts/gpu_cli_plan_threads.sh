#!/bin/bash
# cold `daligner -P` on config 2 with different host-pipeline thread counts
R=$GRAFT_REPO_ROOT
W=$(mktemp -d /dev/shm/planprof.XXXX); cd $W
$R/damar_amd/bin/simdb . SIM 27 -c20 -r2 -e.15 -S135 > /dev/null
for a in 1 2 3 4; do bs=""; for b in $(seq $a -1 1); do bs="$bs SIM.$b"; done; echo "daligner -k14 -j16 SIM.$a $bs"; done > plan.txt
for v in "X=1" "DAMAR_PLAN_TIDY=1"; do
  for i in 1 2 3; do
    rm -rf d001_*; sleep 1; s=$(date +%s.%N)
    env $v DAMAR_CLIPROF=1 $R/damar_amd/bin/daligner -P plan.txt > out.txt 2>&1
    e=$(date +%s.%N); echo "$v run $i wall $(python3 -c "print('%.3f' % ($e-$s))") s $(grep -E "cli: [+]" out.txt | tr "\n" " ")"
  done
done
rm -rf $W

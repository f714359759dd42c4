#!/bin/bash
# Config 2 (BASELINE.json): simulator 27 -c20 -e.15 -r2, DBsplit -s135 -> 4 blocks, 10 block
# pairs.  Runs the HPCdaligner plan with the reference binary and with the MI355X daligner
# (same -j), then compares every .las byte for byte.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${WORKDIR:-/dev/shm/damar_c2}
J=${J:-16}
rm -rf "$W" && mkdir -p "$W/ref" "$W/gpu" && cd "$W"
$ROOT/damar_amd/bin/simdb . SIM ${GENOME:-27} -c${COV:-20} -r${SEED:-2} -e.15 -S${BLOCK:-135} > nblocks.txt
NB=$(cat nblocks.txt); echo "blocks: $NB"
for d in ref gpu; do for f in SIM.db .SIM.idx .SIM.bps; do ln -s $W/$f $W/$d/$f; done; done
if [ -z "$REFMD5" ]; then
echo "== reference plan (-j$J)"
cd $W/ref; t0=$(date +%s%N)
for a in $(seq 1 $NB); do bs=""; for b in $(seq $a -1 1); do bs="$bs SIM.$b"; done
  $ROOT/oracle/_ref/daligner -k14 -j$J SIM.$a $bs; done
t1=$(date +%s%N); echo "reference wall: $(( (t1 - t0) / 1000000 )) ms"
fi
echo "== MI355X plan (-j$J)"
cd $W/gpu; t0=$(date +%s%N)
for a in $(seq 1 $NB); do bs=""; for b in $(seq $a -1 1); do bs="$bs SIM.$b"; done
  timeout -k 10 ${TMO:-600} $ROOT/damar_amd/bin/daligner -k14 -j$J SIM.$a $bs > gpu_$a.log 2>&1 || { tail -5 gpu_$a.log; echo GPU_FAIL; exit 9; }
  if grep -q "Memory access fault" gpu_$a.log; then echo GPU_FAULT; exit 9; fi
done
t1=$(date +%s%N); echo "MI355X wall (incl. process start, DB load, PCIe): $(( (t1 - t0) / 1000000 )) ms"
if [ -n "$REFMD5" ]; then      # REFMD5=<file of an earlier run of the same configuration>: skip the reference run
  cd $W/gpu; if md5sum --quiet -c $REFMD5; then echo "ALL LAS IDENTICAL (md5 of the reference run in $REFMD5)"; rm -rf "$W"; exit 0; else echo "MD5 MISMATCH"; exit 1; fi
fi
cd $W; bad=0; n=0
for f in $(cd ref && ls d001_*/*.las); do n=$((n+1)); if ! cmp -s ref/$f gpu/$f; then echo "DIFF $f"; bad=$((bad+1)); fi; done
ls -la ref/d001_00001 | head -5
echo "files compared: $n, differing: $bad"
( cd ref && md5sum d001_*/*.las ) > $ROOT/gpurun_out/c2_ref_md5.txt
[ $bad -eq 0 ] && echo "ALL LAS IDENTICAL"
rm -rf "$W"
exit $bad

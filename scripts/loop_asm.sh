#!/bin/bash
# the wave loop of report2_kernel (duo_run's inner loop) out of an assembly file: prints the file of duo_run and where the loop is
# usage: scripts/loop_asm.sh build/asm/report.s build/asm/duo_run.s
S=$(grep -n "^_Z7duo_runiPKjj:" $1 | cut -d: -f1)
E=$(awk -v s=$S 'NR>s && /^\.Lfunc_end/ {print NR; exit}' $1)
awk -v s=$S -v e=$E 'NR>=s && NR<=e' $1 > $2
grep -n "Loop Header: Depth=2" $2
grep -c scratch_ $2

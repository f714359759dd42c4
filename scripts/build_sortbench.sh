#!/bin/bash
# builds damar_amd/bin/sortbench (the library's configuration) and tile-shape variants of it
set -e
cd "$(dirname "$0")/../damar_amd/csrc"
mkdir -p ../bin
H="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../include -I. -Ikernels -Wno-unused-value"
$H -o ../bin/sortbench tools/sortbench.hip &
for v in "$@"; do
  IFS=, read t i w <<< "$v"
  $H -DOS_THREADS=$t -DOS_ITEMS=$i -DOS_MINW=$w -o ../bin/sortbench_${t}_${i}_${w} tools/sortbench.hip &
done
wait

#!/bin/bash
# experiment variant of the library: report.hip with the given -D switches, under build/exp_<name>/
# usage: scripts/build_exp.sh <name> -DDAMAR_EXP_...   (results of such a build are wrong on purpose, see report_packed.h)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
make -s -C damar_amd/csrc
mkdir -p build/exp_$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Idamar_amd/csrc "$@" -Wno-unused-value \
  -c damar_amd/csrc/kernels/report.hip -o build/exp_$name/report.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp_$name/libdamar_hip.so build/obj/sort_scan.o build/obj/radix_sort.o \
  build/obj/kmer_index.o build/obj/seed_merge.o build/obj/trace_pts.o build/exp_$name/report.o build/obj/shim.o build/obj/db.o build/obj/las.o \
  build/obj/redundancy.o build/obj/bridge.o -lm -lpthread -lz

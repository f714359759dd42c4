#!/bin/bash
# a variant of the library with extra flags for kernels/report.hip, under build/<name>/ (scripts/bench_lib.py runs bench.py on it)
# usage: scripts/build_variant.sh <name> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/.."
name=$1; shift
make -s -C damar_amd/csrc
mkdir -p build/$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Idamar_amd/csrc -Wno-unused-value $@ \
  -c damar_amd/csrc/kernels/report.hip -o build/$name/report.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/$name/libdamar_hip.so build/obj/sort_scan.o \
  build/obj/kmer_index.o build/obj/seed_merge.o build/obj/trace_pts.o build/$name/report.o build/obj/shim.o build/obj/db.o build/obj/las.o \
  build/obj/redundancy.o build/obj/bridge.o -lm -lpthread -lz

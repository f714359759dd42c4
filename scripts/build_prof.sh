#!/bin/bash
# profiling variant of the library (report.hip with -DDAMAR_PROF) under build/prof/
set -e
cd "$(dirname "$0")/.."
make -s -C damar_amd/csrc
mkdir -p build/prof
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Idamar_amd/csrc -DDAMAR_PROF -Wno-unused-value \
  -c damar_amd/csrc/kernels/report.hip -o build/prof/report.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/prof/libdamar_hip.so build/obj/sort_scan.o build/obj/radix_sort.o \
  build/obj/kmer_index.o build/obj/seed_merge.o build/obj/trace_pts.o build/prof/report.o build/obj/shim.o build/obj/db.o build/obj/las.o \
  build/obj/redundancy.o build/obj/bridge.o -lm -lpthread -lz

#!/bin/bash
# profiling variant of the library with extra -D switches: build/prof_<name>/   (scripts/build_prof_var.sh m0 -DDUO_MARGIN=0)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
make -s -C damar_amd/csrc
mkdir -p build/prof_$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Idamar_amd/csrc -DDAMAR_PROF "$@" -Wno-unused-value \
  -c damar_amd/csrc/kernels/report.hip -o build/prof_$name/report.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/prof_$name/libdamar_hip.so build/obj/sort_scan.o build/obj/radix_sort.o \
  build/obj/kmer_index.o build/obj/seed_merge.o build/obj/trace_pts.o build/prof_$name/report.o build/obj/shim.o build/obj/db.o build/obj/las.o \
  build/obj/redundancy.o build/obj/bridge.o -lm -lpthread -lz

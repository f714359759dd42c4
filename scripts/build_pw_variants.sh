#!/bin/bash
# cut-down variants of pair_work_mark (kernels/seed_merge.hip) that price its phases: build/exp_pw_<name>/libdamar_hip.so with the
# phase(s) named switched off at run time in a way the compiler cannot see (results are WRONG on purpose).  Timed by
# scripts/gpu_pw_variants.sh build/exp_pw_full build/exp_pw_nobig build/exp_pw_noscreen build/exp_pw_noclassify
set -e
cd "$(dirname "$0")/.."
make -s -C damar_amd/csrc
mkdir -p build/pw
python3 - <<'PY'
src = open('damar_amd/csrc/kernels/seed_merge.hip').read()
a = src.index("void pair_work_mark(")
b = src.index("void damar_launch_pair_work(")
body = src[a:b]
def emit(name, f):
    nb = f(body)
    assert nb != body or name == 'full', name
    open('build/pw/seed_merge_%s.hip' % name, 'w').write(src[:a] + nb + src[b:])
emit('full', lambda s: s)
emit('nobig', lambda s: s.replace("const u32 nbig = nb;", "const u32 nbig = nb & 0u;"))
emit('noscreen', lambda s: s.replace("const u32 n_lo = nlo, n_hi = nhi;", "const u32 n_lo = nlo & 0u, n_hi = nhi & 0u;").replace("const u32 nbig = nb;", "const u32 nbig = nb & 0u;"))
emit('noclassify', lambda s: s.replace("const u32 nheads = nh;", "const u32 nheads = nh & 0u;"))
PY
for v in full nobig noscreen noclassify; do
  mkdir -p build/exp_pw_$v
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Idamar_amd/csrc -Idamar_amd/csrc/kernels -Wno-unused-value -Wno-pass-failed \
      -c build/pw/seed_merge_$v.hip -o build/exp_pw_$v/seed_merge.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp_pw_$v/libdamar_hip.so build/obj/sort_scan.o build/obj/radix_sort.o build/obj/kmer_index.o \
      build/exp_pw_$v/seed_merge.o build/obj/trace_pts.o build/obj/report.o build/obj/shim.o build/obj/db.o build/obj/las.o build/obj/redundancy.o build/obj/bridge.o \
      -lm -lpthread -lz ) &
done
wait
ls -la build/exp_pw_*/libdamar_hip.so

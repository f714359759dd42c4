/* roofcal.hip -- issue-rate calibration for the counters the report kernel is read with (gfx950).
 *
 * "VALU busy" derived from SQ_ACTIVE_INST_VALU needs a calibrated scale.  This
 * program runs pure instruction streams with W waves per SIMD on every SIMD of the chip:
 *   valu : 64 independent v_add_u32 per loop trip (8 accumulators)
 *   salu : 64 s_add_u32 per loop trip (8 scalar accumulators)
 *   mix  : 32 v_add_u32 interleaved with 32 s_add_u32
 *   perm : 32 ds_bpermute_b32 + 32 v_add_u32 (the lane-exchange the wave kernel uses)
 * and prints wave-instructions per cycle per SIMD (VALU) / per CU (SALU) from HIP-event time at the
 * measured clock.  Under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
 * SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- roofcal` the same launches give the counter
 * values at a known, saturated issue rate: the scale for report_kernel's counters.
 *
 *   roofcal [trips]      (default 20000 loop trips per wave)
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define V8(a) "v_add_u32 %0, %0, " a "\n v_add_u32 %1, %1, " a "\n v_add_u32 %2, %2, " a "\n v_add_u32 %3, %3, " a "\n" \
              "v_add_u32 %4, %4, " a "\n v_add_u32 %5, %5, " a "\n v_add_u32 %6, %6, " a "\n v_add_u32 %7, %7, " a "\n"
#define S8    "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n" \
              "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"

__global__ __launch_bounds__(64) void cal_valu(unsigned *out, int trips)
{ unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  for (int t = 0; t < trips; t++)
    asm volatile(V8("1") V8("2") V8("3") V8("1") V8("2") V8("3") V8("1") V8("2")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u)
    out[0] = a0;
}

__global__ __launch_bounds__(64) void cal_salu(unsigned *out, int trips)
{ unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
  for (int t = 0; t < trips; t++)
    asm volatile(S8 S8 S8 S8 S8 S8 S8 S8
                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
  if ((s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) == 0x12345678u)
    out[0] = s0;
}

#define M4 "v_add_u32 %0, %0, 1\n s_add_u32 %4, %4, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %5, %5, 1\n" \
           "v_add_u32 %2, %2, 1\n s_add_u32 %6, %6, 1\n v_add_u32 %3, %3, 1\n s_add_u32 %7, %7, 1\n"
__global__ __launch_bounds__(64) void cal_mix(unsigned *out, int trips)
{ unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, s0 = 0, s1 = 1, s2 = 2, s3 = 3;
  for (int t = 0; t < trips; t++)
    asm volatile(M4 M4 M4 M4 M4 M4 M4 M4
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
  if ((a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3) == 0x12345678u)
    out[0] = a0;
}

__global__ __launch_bounds__(64) void cal_perm(unsigned *out, int trips)
{ int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
  const int src = ((threadIdx.x + 1) & 63) << 2;
  for (int t = 0; t < trips; t++)
    { for (int i = 0; i < 8; i++)
        { a0 = __builtin_amdgcn_ds_bpermute(src, a0) + 1;
          a1 = __builtin_amdgcn_ds_bpermute(src, a1) + 1;
          a2 = __builtin_amdgcn_ds_bpermute(src, a2) + 1;
          a3 = __builtin_amdgcn_ds_bpermute(src, a3) + 1;
        }
    }
  if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345678)
    out[0] = a0;
}

typedef void (*kern_t)(unsigned *, int);

static double run(kern_t k, int blocks, int trips, unsigned *out)
{ hipEvent_t e0, e1;
  float ms;
  CHECK(hipEventCreate(&e0));  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, trips / 8);       /* warm */
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, trips);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));  CHECK(hipEventDestroy(e1));
  return ms * 1e-3;
}

int main(int argc, char **argv)
{ int trips = argc > 1 ? atoi(argv[1]) : 20000;
  hipDeviceProp_t pr;
  unsigned *out;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  const double ghz = pr.clockRate * 1e-6;
  CHECK(hipMalloc(&out, 64));
  setvbuf(stdout, NULL, _IOLBF, 0);
  printf("device %s (%s), %d CUs, nominal %.2f GHz; %d trips x 64 instructions per wave\n", pr.name, pr.gcnArchName, cus, ghz, trips);
  struct { const char *name; kern_t k; double per_trip_valu, per_trip_salu, per_trip_lds; } ks[] =
    { { "valu", cal_valu, 64, 0, 0 }, { "salu", cal_salu, 0, 64, 0 }, { "mix", cal_mix, 32, 32, 0 }, { "perm", cal_perm, 32, 0, 32 } };
  for (auto &kk : ks)
    for (int w = 1; w <= 8; w *= 2)
      { const int blocks = cus * 4 * w;
        const double s = run(kk.k, blocks, trips, out);
        const double cyc = s * ghz * 1e9;
        const double waves_per_simd = w;
        printf("%-5s %d waves/SIMD: %8.3f ms", kk.name, w, s * 1e3);
        if (kk.per_trip_valu > 0)
          printf("  VALU %.3f wave-instr/cycle/SIMD (%.2f cycles each)", kk.per_trip_valu * trips * waves_per_simd / cyc,
                 cyc / (kk.per_trip_valu * trips * waves_per_simd));
        if (kk.per_trip_salu > 0)
          printf("  SALU %.3f instr/cycle/CU (%.2f cycles each per CU)", kk.per_trip_salu * trips * waves_per_simd * 4 / cyc,
                 cyc / (kk.per_trip_salu * trips * waves_per_simd * 4));
        if (kk.per_trip_lds > 0)
          printf("  bpermute %.3f /cycle/CU", kk.per_trip_lds * trips * waves_per_simd * 4 / cyc);
        printf("   (at the nominal clock)\n");
      }
  CHECK(hipFree(out));
  return 0;
}

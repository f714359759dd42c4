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
#include <cstring>

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

/* `roofcal ops`: what ONE vector instruction of the kinds the wave loop is made of costs the SIMD, 64 per loop trip on 8
   independent registers (so that the rate, not the latency, is measured), 5 and 8 wavefronts per SIMD. */
#define OP8(T) T("%0", "%1") T("%1", "%2") T("%2", "%3") T("%3", "%4") T("%4", "%5") T("%5", "%6") T("%6", "%7") T("%7", "%0")
#define OPK(name, T, SETUP, ...)                                                                                       \
__global__ __launch_bounds__(64) void name(unsigned *out, int trips)                                                    \
{ unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;                                    \
  asm volatile(SETUP ::: "vcc", "s10", "s11");                                                                          \
  for (int t = 0; t < trips; t++)                                                                                       \
    asm volatile(OP8(T) OP8(T) OP8(T) OP8(T) OP8(T) OP8(T) OP8(T) OP8(T)                                                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : __VA_ARGS__);    \
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u)                                                           \
    out[0] = a0;                                                                                                        \
}
#define T_ADD(d, s)   "v_add_u32 " d ", " d ", " s "\n"
#define T_MOV(d, s)   "v_mov_b32 " d ", " s "\n"
#define T_CND(d, s)   "v_cndmask_b32 " d ", " d ", " s ", vcc\n"
#define T_CND64(d, s) "v_cndmask_b32_e64 " d ", " d ", " s ", s[10:11]\n"
#define T_CNDK(d, s)  "v_cndmask_b32_e64 " d ", 0, 1, s[10:11]\n"
#define T_CNDI(d, s)  "v_cndmask_b32 " d ", " s ", " d ", vcc\n"
#define T_BFI(d, s)   "v_bfi_b32 " d ", " s ", " d ", " s "\n"
#define T_MAX(d, s)   "v_max_i32 " d ", " d ", " s "\n"
#define T_AND(d, s)   "v_and_b32 " d ", " d ", " s "\n"
#define T_LSHL(d, s)  "v_lshlrev_b32 " d ", 1, " s "\n"
#define T_SUB(d, s)   "v_sub_u32 " d ", " d ", " s "\n"
#define T_CMPV(d, s)  "v_cmp_gt_i32 vcc, " d ", " s "\n"
#define T_PAIRV(d, s) "v_cmp_gt_i32 vcc, " d ", " s "\n v_cndmask_b32 " d ", " d ", " s ", vcc\n"
#define T_PAIRS(d, s) "v_cmp_gt_i32_e64 s[10:11], " d ", " s "\n v_cndmask_b32_e64 " d ", " d ", " s ", s[10:11]\n"
#define T_OR(d, s)    "v_or_b32 " d ", " d ", " s "\n"
#define T_XOR(d, s)   "v_xor_b32 " d ", " d ", " s "\n"
#define T_LSHR(d, s)  "v_lshrrev_b32 " d ", 1, " s "\n"
#define T_ASHR(d, s)  "v_ashrrev_i32 " d ", 1, " s "\n"
#define T_MIN(d, s)   "v_min_i32 " d ", " d ", " s "\n"
#define T_ADDL(d, s)  "v_add_lshl_u32 " d ", " d ", " s ", 2\n"
#define T_LOR(d, s)   "v_lshl_or_b32 " d ", " s ", 3, " d "\n"
#define T_ANDOR(d, s) "v_and_or_b32 " d ", " s ", 31, " d "\n"
#define T_MBCNT(d, s) "v_mbcnt_lo_u32_b32 " d ", -1, " s "\n"
#define T_FFBL(d, s)  "v_ffbl_b32 " d ", " s "\n"
#define T_MUL(d, s)   "v_mul_lo_u32 " d ", " d ", " s "\n"
#define T_DPP(d, s)   "v_mov_b32_dpp " d ", " s " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_MAXD(d, s)  "v_max_i32_dpp " d ", " s ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_CMP(d, s)   "v_cmp_gt_i32_e64 s[10:11], " d ", " s "\n"
#define T_RDL(d, s)   "v_readlane_b32 s10, " d ", 3\n"
#define T_BCNT(d, s)  "v_bcnt_u32_b32 " d ", " s ", " d "\n"
#define T_ADD3(d, s)  "v_add3_u32 " d ", " d ", " s ", " s "\n"
#define T_BFE(d, s)   "v_bfe_u32 " d ", " s ", 3, 5\n"
#define T_ALIGN(d, s) "v_alignbit_b32 " d ", " d ", " s ", 7\n"
OPK(op_add,  T_ADD,  "", "memory")
OPK(op_mov,  T_MOV,  "", "memory")
OPK(op_cnd,  T_CND,  "s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555", "memory")
OPK(op_cnd64, T_CND64, "s_mov_b32 s10, 0x55555555\n s_mov_b32 s11, 0x55555555", "memory")
OPK(op_cndk, T_CNDK, "s_mov_b32 s10, 0x55555555\n s_mov_b32 s11, 0x55555555", "memory")
OPK(op_cndi, T_CNDI, "s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555", "memory")
OPK(op_bfi,  T_BFI,  "", "memory")
OPK(op_max,  T_MAX,  "", "memory")
OPK(op_and,  T_AND,  "", "memory")
OPK(op_lshl, T_LSHL, "", "memory")
OPK(op_sub,  T_SUB,  "", "memory")
OPK(op_cmpv, T_CMPV, "", "vcc")
OPK(op_pairv, T_PAIRV, "", "vcc")
OPK(op_pairs, T_PAIRS, "", "s10", "s11")
OPK(op_or,   T_OR,   "", "memory")
OPK(op_xor,  T_XOR,  "", "memory")
OPK(op_lshr, T_LSHR, "", "memory")
OPK(op_ashr, T_ASHR, "", "memory")
OPK(op_min,  T_MIN,  "", "memory")
OPK(op_addl, T_ADDL, "", "memory")
OPK(op_lor,  T_LOR,  "", "memory")
OPK(op_andor, T_ANDOR, "", "memory")
OPK(op_mbcnt, T_MBCNT, "", "memory")
OPK(op_ffbl, T_FFBL, "", "memory")
OPK(op_mul,  T_MUL,  "", "memory")
OPK(op_dpp,  T_DPP,  "", "memory")
OPK(op_maxd, T_MAXD, "", "memory")
OPK(op_cmp,  T_CMP,  "", "s10", "s11")
OPK(op_rdl,  T_RDL,  "", "s10")
OPK(op_bcnt, T_BCNT, "", "memory")
OPK(op_add3, T_ADD3, "", "memory")
OPK(op_bfe,  T_BFE,  "", "memory")
OPK(op_algn, T_ALIGN, "", "memory")

__global__ __launch_bounds__(64) void op_shl64(unsigned *out, int trips)
{ unsigned long long a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
#define T_SHL(d, s) "v_lshlrev_b64 " d ", 1, " s "\n"
  for (int t = 0; t < trips; t++)
    asm volatile(OP8(T_SHL) OP8(T_SHL) OP8(T_SHL) OP8(T_SHL) OP8(T_SHL) OP8(T_SHL) OP8(T_SHL) OP8(T_SHL)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u)
    out[0] = (unsigned) a0;
}

__global__ __launch_bounds__(64) void op_mov64(unsigned *out, int trips)
{ unsigned long long a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
#define T_MOV64(d, s) "v_mov_b64 " d ", " s "\n"
  for (int t = 0; t < trips; t++)
    asm volatile(OP8(T_MOV64) OP8(T_MOV64) OP8(T_MOV64) OP8(T_MOV64) OP8(T_MOV64) OP8(T_MOV64) OP8(T_MOV64) OP8(T_MOV64)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u)
    out[0] = (unsigned) a0;
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

static double run(kern_t k, int blocks, int trips, unsigned *out);

static int ops_main(int trips)
{ hipDeviceProp_t pr;
  unsigned *out;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  const double ghz = pr.clockRate * 1e-6;
  CHECK(hipMalloc(&out, 64));
  setvbuf(stdout, NULL, _IOLBF, 0);
  printf("device %s (%s), %d CUs, nominal %.2f GHz; cycles of SIMD time per wave64 instruction (rate, 8 independent registers)\n",
         pr.name, pr.gcnArchName, cus, ghz);
  struct { const char *name; kern_t k; } ks[] =
    { { "v_add_u32", op_add }, { "v_mov_b32", op_mov }, { "v_mov_b64", op_mov64 }, { "v_cndmask_b32 (vcc)", op_cnd }, { "v_cndmask_b32 (vcc, other operand order)", op_cndi },
      { "v_cndmask_b32_e64 (SGPR pair)", op_cnd64 }, { "v_cndmask_b32_e64 0, 1, SGPR", op_cndk }, { "v_bfi_b32", op_bfi }, { "v_max_i32", op_max },
      { "v_and_b32", op_and }, { "v_or_b32", op_or }, { "v_xor_b32", op_xor }, { "v_lshlrev_b32", op_lshl }, { "v_lshrrev_b32", op_lshr },
      { "v_ashrrev_i32", op_ashr }, { "v_min_i32", op_min }, { "v_add_lshl_u32", op_addl }, { "v_lshl_or_b32", op_lor }, { "v_and_or_b32", op_andor },
      { "v_mbcnt_lo_u32_b32", op_mbcnt }, { "v_ffbl_b32", op_ffbl }, { "v_sub_u32", op_sub }, { "v_cmp_gt_i32 -> vcc", op_cmpv },
      { "v_cmp -> vcc + v_cndmask vcc (per PAIR)", op_pairv }, { "v_cmp -> SGPR + v_cndmask_e64 SGPR (per PAIR)", op_pairs }, { "v_mul_lo_u32", op_mul },
      { "v_mov_b32_dpp row_shr", op_dpp }, { "v_max_i32_dpp row_shr", op_maxd }, { "v_cmp_gt_i32 -> SGPR pair", op_cmp },
      { "v_readlane_b32", op_rdl }, { "v_bcnt_u32_b32", op_bcnt }, { "v_add3_u32", op_add3 }, { "v_bfe_u32", op_bfe },
      { "v_alignbit_b32", op_algn }, { "v_lshlrev_b64", op_shl64 } };
  for (auto &kk : ks)
    { printf("%-44s", kk.name);
      for (int w = 1; w <= 8; w = (w == 1 ? 5 : w == 5 ? 8 : 9))
        { const double s = run(kk.k, cus * 4 * w, trips, out);
          printf("  %d waves/SIMD: %5.2f", w, s * ghz * 1e9 / (64.0 * trips * w));
        }
      printf("\n");
    }
  CHECK(hipFree(out));
  return 0;
}

int main(int argc, char **argv)
{ if (argc > 1 && strcmp(argv[1], "ops") == 0)
    return ops_main(argc > 2 ? atoi(argv[2]) : 20000);
  int trips = argc > 1 ? atoi(argv[1]) : 20000;
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

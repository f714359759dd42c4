/* dev_common.h -- shared declarations of the gfx950 kernels of the overlap path.
 * Wave size is 64 everywhere (CDNA4); nothing here is portable to 32-wide hardware. */
#ifndef DAMAR_DEV_COMMON_H
#define DAMAR_DEV_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#define HIP_CHECK(expr)                                                                   \
  do { hipError_t e_ = (expr);                                                            \
       if (e_ != hipSuccess)                                                              \
         { fprintf(stderr, "damar: HIP error %s at %s:%d: %s\n", hipGetErrorName(e_),     \
                   __FILE__, __LINE__, hipGetErrorString(e_));                            \
           fflush(NULL);                                                                  \
           _exit(1);                                                                      \
         }                                                                                \
     } while (0)

typedef unsigned long long u64;
typedef unsigned int       u32;
typedef unsigned short     u16;
typedef unsigned char      u8;

#define WAVE 64

#ifdef __HIPCC__
__device__ __forceinline__ int  lane_id()            { return (int) (threadIdx.x & 63); }
/* Ballot of a condition that is already a lane mask: HIP's __ballot(int) first turns the bool into
   0/1 per lane and compares it again (two VALU instructions); the builtin takes the mask as it is. */
__device__ __forceinline__ u64  wballot(bool p)      { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wany(bool p)         { return __builtin_amdgcn_ballot_w64(p) != 0; }
__device__ __forceinline__ u64  lanes_below(int l)   { return (l == 0) ? 0ull : (~0ull >> (64 - l)); }
__device__ __forceinline__ int  bcast_i(int v, int l){ return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int  first_i(int v)       { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64  bcast_u64(u64 v, int l)
{ u32 lo = (u32) __builtin_amdgcn_readlane((int) (u32) v, l);
  u32 hi = (u32) __builtin_amdgcn_readlane((int) (u32) (v >> 32), l);
  return ((u64) hi << 32) | lo;
}
__device__ __forceinline__ int wave_max_i(int v)
{ for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = (t > v) ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_min_i(int v)
{ for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = (t < v) ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{ for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
/* inclusive prefix sum across the 64 lanes */
__device__ __forceinline__ int wave_incl_scan_i(int v)
{ int l = lane_id();
  for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o); if (l >= o) v += t; }
  return v;
}
/* all of this wave's earlier global stores are performed before later loads issue */
__device__ __forceinline__ void wave_mem_sync()
{ __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_s_waitcnt(0);
}
#endif

#endif

/* sort_scan.hip -- device-wide exclusive scans of u32 counts for gfx950 (the offsets of the
 * compactions and of the seed-pair emission; the reference does these as per-thread prefix
 * sums on the host, dalign/filter.c:2704-2739).  The radix sort lives in radix_sort.hip.
 */
#include "dev_common.h"
#include "kernels.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS   16
#define SCAN_TILE    (SCAN_THREADS * SCAN_ITEMS)

/***** exclusive scan of u32 -> u32, total returned as u64 *********************************/

__device__ __forceinline__ u32 block_excl_scan_256(u32 v, u32 *lds4, u32 *total)
{ int l = lane_id(), w = threadIdx.x >> 6;
  u32 inc = (u32) wave_incl_scan_i((int) v);
  if (l == 63) lds4[w] = inc;
  __syncthreads();
  u32 base = 0, tot = 0;
  for (int i = 0; i < 4; i++)
    { u32 s = lds4[i];
      if (i < w) base += s;
      tot += s;
    }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS)
void scan_tile_sums(const u32 *__restrict__ in, u64 n, u32 *__restrict__ tsum)
{ __shared__ u32 lds4[4];
  u64 base = (u64) blockIdx.x * SCAN_TILE + (u64) threadIdx.x * SCAN_ITEMS;
  u32 s = 0;
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + i < n) s += in[base + i];
  u32 tot;
  block_excl_scan_256(s, lds4, &tot);
  if (threadIdx.x == 0) tsum[blockIdx.x] = tot;
}

/* one block: exclusive scan of the tile sums in place (64-bit carry), total to *total */
__global__ __launch_bounds__(SCAN_THREADS)
void scan_tile_offsets(u32 *__restrict__ tsum, u32 ntiles, u64 *__restrict__ total)
{ __shared__ u32 lds4[4];
  u64 carry = 0;
  for (u32 b = 0; b < ntiles; b += SCAN_THREADS)
    { u32 i = b + threadIdx.x;
      u32 v = (i < ntiles) ? tsum[i] : 0, tot;
      u32 ex = block_excl_scan_256(v, lds4, &tot);
      if (i < ntiles) tsum[i] = (u32) (carry + ex);
      carry += tot;
    }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(SCAN_THREADS)
void scan_tile_apply(const u32 *__restrict__ in, u32 *__restrict__ out, u64 n, const u32 *__restrict__ toff)
{ __shared__ u32 lds4[4];
  u64 base = (u64) blockIdx.x * SCAN_TILE + (u64) threadIdx.x * SCAN_ITEMS;
  u32 v[SCAN_ITEMS], s = 0;
  for (int i = 0; i < SCAN_ITEMS; i++)
    { v[i] = (base + i < n) ? in[base + i] : 0;
      s += v[i];
    }
  u32 tot;
  u32 ex = block_excl_scan_256(s, lds4, &tot) + toff[blockIdx.x];
  for (int i = 0; i < SCAN_ITEMS; i++)
    { if (base + i < n) out[base + i] = ex;
      ex += v[i];
    }
}

size_t damar_scan_workspace_bytes(u64 n)
{ u64 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  return ((((size_t) (ntiles + 1) * sizeof(u32)) + 63) & ~(size_t) 63) + 64;   /* last 64 B: total */
}

/* out may alias in.  *total_dev (device u64) receives the grand total. */
void damar_exclusive_scan_u32(const u32 *in, u32 *out, u64 n, void *work, u64 *total_dev, hipStream_t st)
{ u32 ntiles = (u32) ((n + SCAN_TILE - 1) / SCAN_TILE);
  u32 *tsum = (u32 *) work;
  if (n == 0)
    { HIP_CHECK(hipMemsetAsync(total_dev, 0, sizeof(u64), st));
      return;
    }
  hipLaunchKernelGGL(scan_tile_sums, dim3(ntiles), dim3(SCAN_THREADS), 0, st, in, n, tsum);
  hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, st, tsum, ntiles, total_dev);
  hipLaunchKernelGGL(scan_tile_apply, dim3(ntiles), dim3(SCAN_THREADS), 0, st, in, out, n, tsum);
}

/* First two phases only: work[t] = exclusive offset of tile t (SCAN_TILE items per tile),
 * *total_dev = grand total.  For consumers that rebuild the offsets inside a tile themselves. */
void damar_tile_offsets_u32(const u32 *in, u64 n, void *work, u64 *total_dev, hipStream_t st)
{ u32 ntiles = (u32) ((n + SCAN_TILE - 1) / SCAN_TILE);
  u32 *tsum = (u32 *) work;
  if (n == 0)
    { HIP_CHECK(hipMemsetAsync(total_dev, 0, sizeof(u64), st));
      return;
    }
  hipLaunchKernelGGL(scan_tile_sums, dim3(ntiles), dim3(SCAN_THREADS), 0, st, in, n, tsum);
  hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, st, tsum, ntiles, total_dev);
}

/* in-place exclusive scan of `ntiles` counts (one workgroup), *total_dev = their sum */
void damar_scan_tile_counts(u32 *tcount, u32 ntiles, u64 *total_dev, hipStream_t st)
{ hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, st, tcount, ntiles, total_dev);
}

/* loads this file's code object now (a lazy load otherwise happens at the first launch, on the launching thread): called by
   the library's start-up thread, beside the caller's first uploads (shim.hip damar_hip_init) */
void damar_preload_scan(void)
{ hipFuncAttributes fa;
  (void) hipFuncGetAttributes(&fa, (const void *) scan_tile_sums);
}

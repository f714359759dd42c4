/* sort_scan.hip -- device-wide exclusive scan and stable LSD radix sort of (key, u32)
 * pairs for gfx950.
 *
 * Replaces the reference's threaded 16-byte-record radix sort (dalign/filter.c:230-435
 * lex_thread / lex_sort), which it uses for the k-mer index (filter.c:854) and for the
 * seed pairs (filter.c:2776).  Both orders the reference produces are total orders on
 * (key, original position), so any stable sort on the same key bits yields the same
 * sequence.
 *
 * Layout: keys and payloads are separate arrays (SoA) so every pass streams
 * 4/8-byte keys and 4-byte payloads with fully coalesced 256-B wave accesses instead
 * of the reference's 16-byte AoS records.  HBM-bound: per pass the histogram kernel
 * reads the keys once, the scatter kernel reads keys+payloads once and writes them
 * once; the scatter stages each 4096-item tile through LDS so that every digit's
 * items leave the CU as one contiguous run.
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

/***** radix sort ******************************************************************************/

#define RS_THREADS 256
#ifndef RS_ROUNDS
#ifndef RS_ROUNDS
#ifndef RS_ROUNDS
#define RS_ROUNDS  16                              /* items per thread (scripts/gpu_rs8.sh builds 4 and 8) */
#endif
#endif
#endif
#define RS_TILE    (RS_THREADS * RS_ROUNDS)     /* 4096 items per workgroup            */
#define RS_WSPAN   (RS_TILE / 4)                /* contiguous items owned by one wave  */

template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS)
void radix_hist(const KeyT *__restrict__ keys, u64 n, int shift, u32 mask,
                u32 *__restrict__ ghist, u32 ntiles)
{ __shared__ u32 hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  u64 base = (u64) blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_ROUNDS; r++)
    { u64 i = base + (u64) r * RS_THREADS + threadIdx.x;
      if (i < n)
        atomicAdd(&hist[(u32) (keys[i] >> shift) & mask], 1u);
    }
  __syncthreads();
  ghist[(u64) threadIdx.x * ntiles + blockIdx.x] = hist[threadIdx.x];
}

/* One workgroup per digit: exclusive prefix of the digit's tile counts in place (its row of ghist) and the digit's total.
   With the 256 totals a scatter workgroup finds its global offsets itself -- one launch between histogram and scatter
   instead of the three of a device-wide scan over all 256 x ntiles counts. */
__global__ __launch_bounds__(RS_THREADS)
void radix_row_scan(u32 *__restrict__ ghist, u32 ntiles, u32 *__restrict__ dtot)
{ __shared__ u32 lds4[4];
  u32 *row = ghist + (u64) blockIdx.x * ntiles;
  u32 carry = 0;
  for (u32 b = 0; b < ntiles; b += RS_THREADS * 8)
    { const u32 base = b + threadIdx.x * 8;
      u32 v[8], sum = 0, tot;
#pragma unroll
      for (int i = 0; i < 8; i++)
        { v[i] = (base + i < ntiles) ? row[base + i] : 0;
          sum += v[i];
        }
      u32 ex = block_excl_scan_256(sum, lds4, &tot) + carry;
#pragma unroll
      for (int i = 0; i < 8; i++)
        { if (base + i < ntiles) row[base + i] = ex;
          ex += v[i];
        }
      carry += tot;
    }
  if (threadIdx.x == 0)
    dtot[blockIdx.x] = carry;
}

/* Stable scatter of one tile.  Wave w owns items [w*1024,(w+1)*1024) of the tile in
 * rounds of 64 consecutive items, so (wave, round, lane) order == input order. */
template <typename KeyT, bool HV>          /* HV: a u32 payload travels with the key */
__global__ __launch_bounds__(RS_THREADS, 4)
void radix_scatter(const KeyT *__restrict__ kin, const u32 *__restrict__ vin,
                   KeyT *__restrict__ kout, u32 *__restrict__ vout, u64 n,
                   int shift, u32 mask, const u32 *__restrict__ gscan, const u32 *__restrict__ dtot, u32 ntiles)
{ /* keys and payload are staged through the SAME buffer one after the other: 32 + 6 KB of LDS per workgroup for u64 keys
     instead of 54 KB, i.e. 4 resident workgroups per CU instead of 2 (3 x 54 KB does not fit the 160 KB) */
  __shared__ KeyT skey[RS_TILE];
  u32 *const sval = (u32 *) skey;
  __shared__ u32  cnt[4][256];
  __shared__ u32  dstart[256];
  __shared__ u32  gadj[256];
  __shared__ u32  lds4[4];

  const int l = lane_id(), w = threadIdx.x >> 6;
  const u64 tbase = (u64) blockIdx.x * RS_TILE;
  const u64 wbase = tbase + (u64) w * RS_WSPAN;

  for (int i = 0; i < 4; i++)
    cnt[i][threadIdx.x] = 0;
  __syncthreads();

  KeyT key[RS_ROUNDS];
  u32  rnk[RS_ROUNDS];
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++)
    { u64  i = wbase + (u64) r * 64 + l;
      bool ok = i < n;
      key[r] = ok ? kin[i] : (KeyT) 0;
    }
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++)
    { u64  i = wbase + (u64) r * 64 + l;
      bool ok = i < n;
      u32  d = (u32) (key[r] >> shift) & mask;
      u64  peers = __ballot(ok);
#pragma unroll
      for (int b = 0; b < 8; b++)
        { bool bit = (d >> b) & 1;
          u64  m = __ballot(bit);
          peers &= bit ? m : ~m;
        }
      u32 before = cnt[w][d];
      u32 mine   = (u32) __popcll(peers & lanes_below(l));
      rnk[r] = before + mine;
      if (ok && mine == 0)
        cnt[w][d] = before + (u32) __popcll(peers);
    }
  __syncthreads();

  { u32 c0 = cnt[0][threadIdx.x], c1 = cnt[1][threadIdx.x], c2 = cnt[2][threadIdx.x], c3 = cnt[3][threadIdx.x];
    u32 tot = c0 + c1 + c2 + c3, all;
    const u32 dbase = block_excl_scan_256(dtot[threadIdx.x], lds4, &all);      /* where this digit's items start */
    u32 ex = block_excl_scan_256(tot, lds4, &all);
    cnt[0][threadIdx.x] = 0;
    cnt[1][threadIdx.x] = c0;
    cnt[2][threadIdx.x] = c0 + c1;
    cnt[3][threadIdx.x] = c0 + c1 + c2;
    dstart[threadIdx.x] = ex;
    gadj[threadIdx.x]   = dbase + gscan[(u64) threadIdx.x * ntiles + blockIdx.x] - ex;
  }
  __syncthreads();

#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++)
    { u64 i = wbase + (u64) r * 64 + l;
      if (i < n)
        { u32 d  = (u32) (key[r] >> shift) & mask;
          u32 lp = dstart[d] + cnt[w][d] + rnk[r];
          rnk[r] = lp;                             /* position inside the tile's output */
          skey[lp] = key[r];
        }
    }
  __syncthreads();

  const u32 have = (n - tbase < (u64) RS_TILE) ? (u32) (n - tbase) : (u32) RS_TILE;
  u32 gdst[RS_ROUNDS];                             /* where this thread's output positions go (n < 2^32) */
#pragma unroll
  for (int q = 0; q < RS_ROUNDS; q++)
    { const u32 i = threadIdx.x + (u32) q * RS_THREADS;
      if (i < have)
        { KeyT k = skey[i];
          u32  d = (u32) (k >> shift) & mask;
          gdst[q] = gadj[d] + i;
          kout[gdst[q]] = k;
        }
    }
  if (HV)
    { __syncthreads();
#pragma unroll
      for (int r = 0; r < RS_ROUNDS; r++)
        { u64 i = wbase + (u64) r * 64 + l;
          if (i < n)                                 /* (the payload is loaded only now: 16 registers less while ranking) */
            sval[rnk[r]] = vin[i];
        }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < RS_ROUNDS; q++)
        { const u32 i = threadIdx.x + (u32) q * RS_THREADS;
          if (i < have)
            vout[gdst[q]] = sval[i];
        }
    }
}

size_t damar_sort_workspace_bytes(u64 n)
{ u64 ntiles = (n + RS_TILE - 1) / RS_TILE;
  return (size_t) (256 * ntiles) * sizeof(u32) + 256 * sizeof(u32) + 256;
}

/* Sorts on key bits [0, nbits).  Ping-pongs between (k0,v0) and (k1,v1); returns 0 if
 * the result is in (k0,v0), 1 if in (k1,v1). */
template <typename KeyT, bool HV>
static int radix_sort_impl(KeyT *k0, u32 *v0, KeyT *k1, u32 *v1, u64 n, int nbits,
                           void *work, hipStream_t st)
{ u32  ntiles = (u32) ((n + RS_TILE - 1) / RS_TILE);
  u32 *ghist = (u32 *) work;
  u32 *dtot  = (u32 *) ((char *) work + (((size_t) 256 * ntiles * sizeof(u32) + 63) & ~(size_t) 63));
  int  side = 0;
  if (n == 0)
    return 0;
  for (int shift = 0; shift < nbits; shift += 8)
    { int  bits = (nbits - shift < 8) ? nbits - shift : 8;
      u32  mask = (1u << bits) - 1;
      KeyT *ki = side ? k1 : k0, *ko = side ? k0 : k1;
      u32  *vi = side ? v1 : v0, *vo = side ? v0 : v1;
      hipLaunchKernelGGL(radix_hist<KeyT>, dim3(ntiles), dim3(RS_THREADS), 0, st, ki, n, shift, mask, ghist, ntiles);
      hipLaunchKernelGGL(radix_row_scan, dim3(256), dim3(RS_THREADS), 0, st, ghist, ntiles, dtot);
      hipLaunchKernelGGL((radix_scatter<KeyT, HV>), dim3(ntiles), dim3(RS_THREADS), 0, st,
                         ki, vi, ko, vo, n, shift, mask, ghist, dtot, ntiles);
      side ^= 1;
    }
  return side;
}

int damar_radix_sort_u32(u32 *k0, u32 *v0, u32 *k1, u32 *v1, u64 n, int nbits, void *work, hipStream_t st)
{ return radix_sort_impl<u32, true>(k0, v0, k1, v1, n, nbits, work, st); }

/* keys only */
int damar_radix_sort_keys_u32(u32 *k0, u32 *k1, u64 n, int nbits, void *work, hipStream_t st)
{ return radix_sort_impl<u32, false>(k0, NULL, k1, NULL, n, nbits, work, st); }

int damar_radix_sort_u64(u64 *k0, u32 *v0, u64 *k1, u32 *v1, u64 n, int nbits, void *work, hipStream_t st)
{ return radix_sort_impl<u64, true>(k0, v0, k1, v1, n, nbits, work, st); }

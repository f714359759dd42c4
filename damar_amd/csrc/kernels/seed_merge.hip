/* seed_merge.hip -- merge of two sorted k-mer indexes into seed pairs, and the
 * read-pair work list for the report kernel (gfx950).
 *
 * Restates reference dalign/filter.c:1039-1165 (count_thread), :1170-1358
 * (merge_thread), the slice bookkeeping of :2606-2620 / :2804-2816 and the pre-check
 * of :2210-2220 as data-parallel passes:
 *
 *   merge_count : one thread per A-index entry -> how many B entries pair with it
 *   (scan)      : hit offsets
 *   merge_limit : self mode only, drops runs whose mutual count reaches `limit`
 *   merge_emit  : one thread per seed pair -> 64-bit sort key + diagonal
 *   pair_flags  : heads of (bread, aread) runs that the report loop would examine
 *
 * Seed layout in HBM: key = bread << (abits+pbits) | aread << pbits | apos (u64) and
 * val = diag (int32), 12 B per seed instead of the reference's 16-byte SeedPair.
 * Sorting stably on the key's low bits gives the reference's total order (bread,
 * aread, apos, then bpos ascending) because emission order for one A entry is
 * B-index order.
 */
#include "dev_common.h"
#include "kernels.h"

__device__ __forceinline__ u32 read_of_pos(const DevBlock &b, u32 p)
{ u32 r = b.coarse[p >> COARSE_SHIFT];
  while (b.boff[r + 1] <= p)
    r += 1;
  return r;
}

template <typename CodeT>
__device__ __forceinline__ void code_run(const CodeT *__restrict__ codes, const u32 *__restrict__ table,
                                         int shift, CodeT c, u32 *lb, u32 *ub)
{ u32 q = (u32) (c >> shift);
  u32 lo = table[q], hi = table[q + 1];
  if (shift == 0)
    { *lb = lo; *ub = hi; return; }
  u32 a = lo, b = hi;
  while (a < b)
    { u32 m = (a + b) >> 1;
      if (codes[m] < c) a = m + 1; else b = m;
    }
  *lb = a;
  b = hi;
  while (a < b)
    { u32 m = (a + b) >> 1;
      if (codes[m] <= c) a = m + 1; else b = m;
    }
  *ub = a;
}

/* number of entries of bpos[jb,ib) strictly below `bound` (bpos ascending in a run) */
__device__ __forceinline__ u32 count_below(const u32 *__restrict__ bpos, u32 jb, u32 ib, u32 bound)
{ u32 a = jb, b = ib;
  while (a < b)
    { u32 m = (a + b) >> 1;
      if (bpos[m] < bound) a = m + 1; else b = m;
    }
  return a - jb;
}

template <typename CodeT>
__global__ __launch_bounds__(256)
void merge_count(MergeArgs m, u32 *__restrict__ cnt, u32 *__restrict__ jbout)
{ u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m.alen)
    return;
  const CodeT *acode = (const CodeT *) m.acode, *bcode = (const CodeT *) m.bcode;
  const CodeT c = acode[i];
  u32 jb, ib, n = 0;
  code_run<CodeT>(bcode, m.btab, m.kbits - m.btbits, c, &jb, &ib);
  if (ib > jb)
    { if (!m.self)
        { u32 ja, ia;                                          /* filter.c:1334-1335 */
          code_run<CodeT>(acode, m.atab, m.kbits - m.atbits, c, &ja, &ia);
          if ((u64) (ia - ja) * (u64) (ib - jb) < (u64) m.limit)
            n = ib - jb;
        }
      else
        { u32 p = m.apos[i], bound;                            /* filter.c:1219-1246 */
          if (m.identity)
            bound = m.comp ? m.ablk.boff[read_of_pos(m.ablk, p) + 1] : p;
          else
            bound = m.ablk.boff[read_of_pos(m.ablk, p)];
          n = count_below(m.bpos, jb, ib, bound);
        }
    }
  cnt[i]   = n;
  jbout[i] = jb;
}

void damar_launch_merge_count(const MergeArgs *m, u32 *cnt, u32 *jb, hipStream_t st)
{ if (m->alen == 0)
    return;
  if (m->wide)
    hipLaunchKernelGGL(merge_count<u64>, dim3((m->alen + 255) / 256), dim3(256), 0, st, *m, cnt, jb);
  else
    hipLaunchKernelGGL(merge_count<u32>, dim3((m->alen + 255) / 256), dim3(256), 0, st, *m, cnt, jb);
}

/* self mode: a run's mutual count is off[run end] - off[run start]; runs at or over
 * the limit contribute nothing (filter.c:1248 `if (ct < limit)`). */
template <typename CodeT>
__global__ __launch_bounds__(256)
void merge_limit(MergeArgs m, const u32 *__restrict__ off, u64 total, u32 *__restrict__ cnt)
{ u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m.alen || cnt[i] == 0)
    return;
  u32 ja, ia;
  const CodeT *acode = (const CodeT *) m.acode;
  code_run<CodeT>(acode, m.atab, m.kbits - m.atbits, acode[i], &ja, &ia);
  u64 hi = (ia >= m.alen) ? total : (u64) off[ia];
  if (hi - (u64) off[ja] >= (u64) m.limit)
    cnt[i] = 0;
}

void damar_launch_merge_limit(const MergeArgs *m, const u32 *off, u64 total, u32 *cnt, hipStream_t st)
{ if (m->alen == 0)
    return;
  if (m->wide)
    hipLaunchKernelGGL(merge_limit<u64>, dim3((m->alen + 255) / 256), dim3(256), 0, st, *m, off, total, cnt);
  else
    hipLaunchKernelGGL(merge_limit<u32>, dim3((m->alen + 255) / 256), dim3(256), 0, st, *m, off, total, cnt);
}

/* hitgram[ct] = number of equal-code runs whose mutual count is ct (< ngram), filter.c:1039-1165
 * count_thread.  Only launched when the host must lower the cap under memory pressure
 * (filter.c:2634-2699), so plain global atomics will do.  Cross: ct = na * nb; self: the run's
 * count is the sum of its entries' counts = off[run end] - off[run start] of the current scan. */
template <typename CodeT>
__global__ __launch_bounds__(256)
void merge_hitgram(MergeArgs m, const u32 *__restrict__ off, u64 total, u32 ngram, unsigned long long *__restrict__ gram)
{ u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m.alen)
    return;
  const CodeT *acode = (const CodeT *) m.acode, *bcode = (const CodeT *) m.bcode;
  const CodeT c = acode[i];
  if (i > 0 && acode[i - 1] == c)
    return;                                     /* one thread per run of A */
  u32 jb, ib, ja, ia;
  code_run<CodeT>(bcode, m.btab, m.kbits - m.btbits, c, &jb, &ib);
  if (ib <= jb)
    return;
  code_run<CodeT>(acode, m.atab, m.kbits - m.atbits, c, &ja, &ia);
  u64 ct;
  if (!m.self)
    ct = (u64) (ia - ja) * (u64) (ib - jb);
  else
    ct = ((ia >= m.alen) ? total : (u64) off[ia]) - (u64) off[ja];
  if (ct < (u64) ngram)
    atomicAdd(&gram[ct], 1ull);
}

void damar_launch_merge_hitgram(const MergeArgs *m, const u32 *off, u64 total, u32 ngram, unsigned long long *gram,
                                hipStream_t st)
{ if (m->alen == 0)
    return;
  if (m->wide)
    hipLaunchKernelGGL(merge_hitgram<u64>, dim3((m->alen + 255) / 256), dim3(256), 0, st, *m, off, total, ngram, gram);
  else
    hipLaunchKernelGGL(merge_hitgram<u32>, dim3((m->alen + 255) / 256), dim3(256), 0, st, *m, off, total, ngram, gram);
}

/* One workgroup per tile of DAMAR_SCAN_TILE A entries: the tile's hit counts are scanned in
 * LDS, then the tile's hits are dealt out to the threads in order, each finding its A entry by
 * a search of the LDS prefix (no walk over the global offsets), so the seed pairs leave in
 * fully coalesced runs. */
#define ME_ITEMS (DAMAR_SCAN_TILE / 256)
__global__ __launch_bounds__(256)
void merge_emit(MergeArgs m, const u32 *__restrict__ cnt, const u32 *__restrict__ toff,
                const u32 *__restrict__ jb, u64 nhits, u64 *__restrict__ keys, u32 *__restrict__ vals,
                u32 *__restrict__ pid)
{ __shared__ u32 loc[DAMAR_SCAN_TILE + 1];
  __shared__ u32 wsum[4];
  const u32 a0 = blockIdx.x * (u32) DAMAR_SCAN_TILE;
  const int l = lane_id(), w = threadIdx.x >> 6;
  /* thread t owns entries [t*ME_ITEMS, (t+1)*ME_ITEMS) of the tile */
  u32 v[ME_ITEMS], s = 0;
  for (int i = 0; i < ME_ITEMS; i++)
    { const u32 a = a0 + threadIdx.x * ME_ITEMS + i;
      v[i] = (a < m.alen) ? cnt[a] : 0;
      s += v[i];
    }
  u32 inc = (u32) wave_incl_scan_i((int) s);
  if (l == 63) wsum[w] = inc;
  __syncthreads();
  u32 base = 0, T = 0;
  for (int i = 0; i < 4; i++)
    { const u32 x = wsum[i];
      if (i < w) base += x;
      T += x;
    }
  u32 ex = base + inc - s;
  for (int i = 0; i < ME_ITEMS; i++)
    { loc[threadIdx.x * ME_ITEMS + i] = ex;
      ex += v[i];
    }
  if (threadIdx.x == 255)
    loc[DAMAR_SCAN_TILE] = T;
  __syncthreads();
  const u64 h0 = toff[blockIdx.x];
  for (u32 t = threadIdx.x; t < T; t += 256)
    { /* last entry whose first hit is <= t */
      u32 a = 0, b = DAMAR_SCAN_TILE;
      while (b - a > 1)
        { const u32 mid = (a + b) >> 1;
          if (loc[mid] <= t) a = mid; else b = mid;
        }
      const u32 ai = a0 + a;
      const u32 bi = jb[ai] + (t - loc[a]);
      const u32 pa = m.apos[ai], pb = m.bpos[bi];
      const u32 ra = read_of_pos(m.ablk, pa), rb = read_of_pos(m.bblk, pb);
      const u32 xa = pa - m.ablk.boff[ra], xb = pb - m.bblk.boff[rb];
      const u64 h = h0 + t;
      if (h < nhits)
        { const u64 key = ((u64) rb << (m.abits + m.pbits)) | ((u64) ra << m.pbits) | (u64) xa;
          if (m.dbits)
            keys[h] = (key << m.dbits) | (u64) xb;
          else
            { keys[h] = key;
              vals[h] = (u32) ((int) xa - (int) xb);
            }
          if (pid != NULL)                     /* the read pair alone, for the early cut (damar_launch_pair_cut) */
            pid[h] = (rb << m.abits) | ra;
        }
    }
}

void damar_launch_merge_emit(const MergeArgs *m, const u32 *cnt, const u32 *toff, const u32 *jb, u64 nhits,
                             u64 *keys, u32 *vals, u32 *pid, hipStream_t st)
{ if (nhits == 0)
    return;
  const u32 ntiles = (m->alen + DAMAR_SCAN_TILE - 1) / DAMAR_SCAN_TILE;
  hipLaunchKernelGGL(merge_emit, dim3(ntiles), dim3(256), 0, st, *m, cnt, toff, jb, nhits, keys, vals, pid);
}

/* flags[i] = 1 iff hit i starts a (bread,aread) run that report_thread would enter:
 * the run has >= minhit hits (filter.c:2215: hit i+minhit-1 is the same pair) and does
 * not start within the last minhit hits of its thread slice (filter.c:2212-2214:
 * nidx < end - minhit).  Slices end where the reference's NTHREADS partition ends:
 * first index >= (nhits*t)>>nshift whose bread differs from its predecessor's. */
#define SCREEN_MAX   48
#define SCREEN_PANEL 50000                     /* PANEL_SIZE, filter.c:73 */
/* ends of the reference's NTHREADS slices: once per launch, not once per block */
template <typename K>
__global__ __launch_bounds__(64)
void slice_ends(const K *__restrict__ keys, u64 nhits, int bshift, int nshift, u64 *__restrict__ send)
{ const int nthr = 1 << nshift, t = threadIdx.x;
  if (t >= nthr)
    return;
  u64 e;
  if (t == nthr - 1)
    e = nhits;
  else
    { e = (nhits * (u64) (t + 1)) >> nshift;
      if (e > 0)
        { u64 d = keys[e - 1] >> bshift, a = e, b = nhits;     /* first index with bread != d */
          while (a < b)
            { u64 mid = (a + b) >> 1;
              if ((keys[mid] >> bshift) == d) a = mid + 1; else b = mid;
            }
          e = a;
        }
    }
  send[t] = e;
}

/* One workgroup per tile of DAMAR_SCAN_TILE seeds: the head predicate of every seed as one bit
 * (64 seeds per ballot word), and the tile's head count.  The list of heads is then expanded
 * from the bit words (pair_heads_expand) after a scan over the tile counts only -- no 4-byte
 * flag and offset per seed, no device-wide scan over all seeds. */
#define PH_ROUNDS (DAMAR_SCAN_TILE / 256)
template <typename K>
__global__ __launch_bounds__(256)
void pair_heads_mark(const K *__restrict__ keys, u64 nhits, int pbits, int minhit, int nshift,
                     const u64 *__restrict__ send, u64 *__restrict__ bits, u32 *__restrict__ tcount)
{ __shared__ u32 wsum[4];
  const int nthr = nshift < 0 ? 0 : 1 << nshift;       /* nshift < 0: no slices (the seeds went through the early cut) */
  const int l = lane_id(), w = threadIdx.x >> 6;
  const u64 base = (u64) blockIdx.x * DAMAR_SCAN_TILE;
  u32 mine = 0;
  for (int r = 0; r < PH_ROUNDS; r++)
    { const u64 i = base + (u64) r * 256u + threadIdx.x;
      bool f = false;
      if (i < nhits)
        { const K pr = keys[i] >> pbits;
          if ((i == 0 || (keys[i - 1] >> pbits) != pr) && i + (u64) (minhit - 1) < nhits &&
              (keys[i + (u64) (minhit - 1)] >> pbits) == pr)
            { f = true;
              for (int t = 0; t < nthr; t++)
                { u64 e = send[t];
                  if (i < e)
                    { if (i + (u64) minhit >= e) f = false;
                      break;
                    }
                }
            }
        }
      const u64 m = __ballot(f);
      if (l == 0)
        { bits[(base >> 6) + (u64) r * 4 + w] = m;
          mine += (u32) __popcll(m);
        }
    }
  if (l == 0) wsum[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0)
    tcount[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

/* toff = exclusive scan of tcount; one wave per tile: lane j owns bit word j of the tile */
__global__ __launch_bounds__(64)
void pair_heads_expand(const u64 *__restrict__ bits, const u32 *__restrict__ toff, u32 *__restrict__ heads)
{ const u64 wi = (u64) blockIdx.x * 64u + threadIdx.x;
  u64 m = bits[wi];
  const int c = __popcll(m);
  u32 o = toff[blockIdx.x] + (u32) (wave_incl_scan_i(c) - c);
  const u32 first = (u32) (wi << 6);
  while (m)
    { const int b = __ffsll((long long) m) - 1;
      m &= m - 1;
      heads[o++] = first + (u32) b;
    }
}

/* heads = ascending indices of the run heads; *total_dev = their number.  bits: 64 u64 words per
 * tile of DAMAR_SCAN_TILE seeds; scan_work: damar_scan_workspace_bytes(nhits) */
template <typename K>
static void pair_heads_impl(const K *keys, u64 nhits, int pbits, int abits, int minhit, int nshift,
                            u64 *send /* 64 entries of scratch */, u64 *bits, void *scan_work, u64 *total_dev,
                            u32 *heads, hipStream_t st)
{ if (nhits == 0)
    { HIP_CHECK(hipMemsetAsync(total_dev, 0, sizeof(u64), st));
      return;
    }
  if (nshift > 6)
    nshift = 6;
  const u32 ntiles = (u32) ((nhits + DAMAR_SCAN_TILE - 1) / DAMAR_SCAN_TILE);
  u32 *tcount = (u32 *) scan_work;
  if (nshift >= 0)
    hipLaunchKernelGGL(slice_ends<K>, dim3(1), dim3(64), 0, st, keys, nhits, abits + pbits, nshift, send);
  hipLaunchKernelGGL(pair_heads_mark<K>, dim3(ntiles), dim3(256), 0, st, keys, nhits, pbits, minhit, nshift, send, bits, tcount);
  damar_scan_tile_counts(tcount, ntiles, total_dev, st);
  hipLaunchKernelGGL(pair_heads_expand, dim3(ntiles), dim3(64), 0, st, bits, tcount, heads);
}

void damar_launch_pair_heads(const u64 *keys, u64 nhits, int pbits, int abits, int minhit, int nshift,
                             u64 *send, u64 *bits, void *scan_work, u64 *total_dev, u32 *heads, hipStream_t st)
{ pair_heads_impl<u64>(keys, nhits, pbits, abits, minhit, nshift, send, bits, scan_work, total_dev, heads, st);
}

/***** the early cut ***************************************************************************************
 * Of the seed pairs two blocks share, a few per cent belong to read pairs with at least minhit seeds; the rest are
 * chance k-mer matches between unrelated reads that report_thread skips at its first test (filter.c:2212-2215).
 * Rather than carrying them through the six passes of the seed sort, the read pair of every seed (a u32,
 * bread << abits | aread, written by merge_emit next to the seed) is sorted on its own -- the same order, so the
 * same indices, as the seeds will have -- the reference's head test (run of >= minhit seeds, not inside the last
 * minhit seeds of a thread slice) is evaluated there, the surviving pairs are marked in a bitmap over the pair ids,
 * and only their seeds are kept for the sort.  Everything after the sort sees exactly the seeds of the read pairs the
 * reference enters, in the reference's order. */

/* heads = indices (in the sorted pair ids) of the runs report_thread enters; *total_dev = their number */
void damar_launch_pair_heads_ids(const u32 *pids, u64 nhits, int abits, int minhit, int nshift,
                                 u64 *send, u64 *bits, void *scan_work, u64 *total_dev, u32 *heads, hipStream_t st)
{ pair_heads_impl<u32>(pids, nhits, 0, abits, minhit, nshift, send, bits, scan_work, total_dev, heads, st);
}

__global__ __launch_bounds__(256)
void pair_bitmap_set(const u32 *__restrict__ pids, const u32 *__restrict__ heads, u32 nheads, int abits, u32 b_lo, u32 b_hi,
                     u32 *__restrict__ bitmap)
{ const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nheads)
    return;
  const u32 p = pids[heads[t]], rb = p >> abits;
  if (rb < b_lo || rb >= b_hi)                    /* a scheduler may hand this call a B-read range only */
    return;
  atomicOr(&bitmap[p >> 5], 1u << (p & 31));
}

/* bitmap must be zero: (2^idbits + 31) / 32 words */
void damar_launch_pair_bitmap(const u32 *pids, const u32 *heads, u32 nheads, int abits, u32 b_lo, u32 b_hi, u32 *bitmap,
                              hipStream_t st)
{ if (nheads == 0)
    return;
  hipLaunchKernelGGL(pair_bitmap_set, dim3((nheads + 255) / 256), dim3(256), 0, st, pids, heads, nheads, abits, b_lo, b_hi, bitmap);
}

/* survivors of a tile of DAMAR_SCAN_TILE seeds */
__global__ __launch_bounds__(256)
void seed_cut_count(const u64 *__restrict__ keys, u64 nhits, int pbits, const u32 *__restrict__ bitmap, u32 *__restrict__ tcount)
{ __shared__ u32 wsum[4];
  const int l = lane_id(), w = threadIdx.x >> 6;
  const u64 base = (u64) blockIdx.x * DAMAR_SCAN_TILE;
  u32 mine = 0;
  for (int r = 0; r < PH_ROUNDS; r++)
    { const u64 i = base + (u64) r * 256u + threadIdx.x;
      bool f = false;
      if (i < nhits)
        { const u32 p = (u32) (keys[i] >> pbits);
          f = (bitmap[p >> 5] >> (p & 31)) & 1;
        }
      mine += (u32) __popcll(__ballot(f));
    }
  if (l == 0) wsum[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0)
    tcount[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

/* toff = exclusive scan of the tile counts; survivors keep their order */
__global__ __launch_bounds__(256)
void seed_cut_scatter(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u64 nhits, int pbits,
                      const u32 *__restrict__ bitmap, const u32 *__restrict__ toff, u64 *__restrict__ okeys, u32 *__restrict__ ovals)
{ __shared__ u32 wsum[PH_ROUNDS][4];
  const int l = lane_id(), w = threadIdx.x >> 6;
  const u64 base = (u64) blockIdx.x * DAMAR_SCAN_TILE;
  u64 key[PH_ROUNDS];
  u64 mask[PH_ROUNDS];
  for (int r = 0; r < PH_ROUNDS; r++)
    { const u64 i = base + (u64) r * 256u + threadIdx.x;
      bool f = false;
      key[r] = 0;
      if (i < nhits)
        { key[r] = keys[i];
          const u32 p = (u32) (key[r] >> pbits);
          f = (bitmap[p >> 5] >> (p & 31)) & 1;
        }
      mask[r] = __ballot(f);
      if (l == 0) wsum[r][w] = (u32) __popcll(mask[r]);
    }
  __syncthreads();
  u32 o = toff[blockIdx.x];
  for (int r = 0; r < PH_ROUNDS; r++)
    { u32 before = 0;
      for (int x = 0; x < 4; x++)
        { const u32 c = wsum[r][x];
          if (x < w) before += c;
        }
      const u64 i = base + (u64) r * 256u + threadIdx.x;
      if ((mask[r] >> l) & 1)
        { const u32 g = o + before + (u32) __popcll(mask[r] & lanes_below(l));
          okeys[g] = key[r];
          if (vals != NULL)
            ovals[g] = vals[i];
        }
      o += wsum[r][0] + wsum[r][1] + wsum[r][2] + wsum[r][3];
    }
}

/* tcount: one u32 per tile of DAMAR_SCAN_TILE seeds (scan workspace); *total_dev = survivors after the first call */
void damar_launch_seed_cut_count(const u64 *keys, u64 nhits, int pbits, const u32 *bitmap, u32 *tcount, u64 *total_dev,
                                 hipStream_t st)
{ const u32 ntiles = (u32) ((nhits + DAMAR_SCAN_TILE - 1) / DAMAR_SCAN_TILE);
  hipLaunchKernelGGL(seed_cut_count, dim3(ntiles), dim3(256), 0, st, keys, nhits, pbits, bitmap, tcount);
  damar_scan_tile_counts(tcount, ntiles, total_dev, st);
}

void damar_launch_seed_cut_scatter(const u64 *keys, const u32 *vals, u64 nhits, int pbits, const u32 *bitmap,
                                   const u32 *toff, u64 *okeys, u32 *ovals, hipStream_t st)
{ const u32 ntiles = (u32) ((nhits + DAMAR_SCAN_TILE - 1) / DAMAR_SCAN_TILE);
  hipLaunchKernelGGL(seed_cut_scatter, dim3(ntiles), dim3(256), 0, st, keys, vals, nhits, pbits, bitmap, toff, okeys, ovals);
}

/* Screen of the run heads (the vast majority of runs are a few chance k-mer matches between
 * unrelated reads).  A run that fits one A-panel (filter.c:2251: all apos <= PANEL_SIZE) gets its
 * bucket scores computed here exactly as pass 1 of the report loop does (filter.c:2268-2277:
 * a seed adds min(kmer, apos - previous apos in its bucket)); a seed fires only if its bucket
 * plus a neighbour reach hitmin (filter.c:2297).  Runs that cannot fire are dropped from the
 * work list: the report kernel would not have emitted anything for them.  One thread per head,
 * heads compacted first so that the wavefronts are full. */
__global__ __launch_bounds__(256)
void pair_screen(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u64 nhits, int pbits, int dbits,
                 const u32 *__restrict__ heads, u32 nheads, int minhit, int binshift, int kmer, int hitmin,
                 int abits, u32 b_lo, u32 b_hi, u32 *__restrict__ keep)
{ u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nheads)
    return;
  const u64 i = heads[t];
  const u64 pmask = (1ull << pbits) - 1;
  pbits += dbits;                                 /* from here on: the shift that leaves the read pair */
  const u64 pr = keys[i] >> pbits;
  u32 f = 1;
  int n = minhit;
  { const u32 rb = (u32) (pr >> abits);          /* a scheduler may hand this call a B-read range only */
    if (rb < b_lo || rb >= b_hi)
      { keep[t] = 0;
        return;
      }
  }
  while (n <= SCREEN_MAX && i + (u64) n < nhits && (keys[i + (u64) n] >> pbits) == pr)
    n += 1;
  if (n <= SCREEN_MAX && (int) ((keys[i + (u64) (n - 1)] >> dbits) & pmask) <= SCREEN_PANEL)
    { bool ok = false;
      for (int x = 0; x < n && !ok; x++)
        { const int dx = seed_diag(keys[i + (u64) x], vals, i + (u64) x, pmask, dbits) >> binshift;
          int s0 = 0, s1 = 0, p0 = 0, p1 = 0;
          for (int y = 0; y < n; y++)
            { const u64 ky = keys[i + (u64) y];
              const int dy = seed_diag(ky, vals, i + (u64) y, pmask, dbits) >> binshift;
              const int ap = (int) ((ky >> dbits) & pmask);
              if (dy == dx)
                { s0 += (ap - p0 >= kmer) ? kmer : ap - p0;  p0 = ap; }
              else if (dy == dx + 1)
                { s1 += (ap - p1 >= kmer) ? kmer : ap - p1;  p1 = ap; }
            }
          ok = s0 + s1 >= hitmin;
        }
      if (!ok)
        f = 0;
    }
  keep[t] = f;
}

void damar_launch_pair_screen(const u64 *keys, const u32 *vals, u64 nhits, int pbits, int dbits, const u32 *heads, u32 nheads,
                              int minhit, int binshift, int kmer, int hitmin, int abits, u32 b_lo, u32 b_hi,
                              u32 *keep, hipStream_t st)
{ if (nheads == 0)
    return;
  hipLaunchKernelGGL(pair_screen, dim3((nheads + 255) / 256), dim3(256), 0, st, keys, vals, nhits, pbits, dbits, heads, nheads,
                     minhit, binshift, kmer, hitmin, abits, b_lo, b_hi, keep);
}

/* out[off[i]] = src[i] for the kept entries */
__global__ __launch_bounds__(256)
void compact_u32(const u32 *__restrict__ src, const u32 *__restrict__ keep, const u32 *__restrict__ off, u32 n,
                 u32 *__restrict__ out)
{ u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i < n && keep[i])
    out[off[i]] = src[i];
}

void damar_launch_compact_u32(const u32 *src, const u32 *keep, const u32 *off, u32 n, u32 *out, hipStream_t st)
{ if (n == 0)
    return;
  hipLaunchKernelGGL(compact_u32, dim3((n + 255) / 256), dim3(256), 0, st, src, keep, off, n, out);
}

__global__ __launch_bounds__(256)
void compact_index(const u32 *__restrict__ flags, const u32 *__restrict__ off, u64 n, u32 *__restrict__ out)
{ u64 i = (u64) blockIdx.x * 256u + threadIdx.x;
  if (i < n && flags[i])
    out[off[i]] = (u32) i;
}

void damar_launch_compact_index(const u32 *flags, const u32 *off, u64 n, u32 *out, hipStream_t st)
{ if (n == 0)
    return;
  hipLaunchKernelGGL(compact_index, dim3((u32) ((n + 255) / 256)), dim3(256), 0, st, flags, off, n, out);
}

/* Largest-first processing order for the report kernel: one alignment is a long serial chain of
 * wave steps, so a long read pair started late would leave the rest of the chip idle at the end
 * of the launch.  The number of seeds of a pair is a good stand-in for the length of its
 * alignment; key[j] sorts ascending into "most seeds first". */
__global__ __launch_bounds__(256)
void work_cost(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u64 nhits, int pbits0, int abits, int dbits,
               const u32 *__restrict__ aboff, const u32 *__restrict__ bboff,
               const u32 *__restrict__ work, u32 nwork, u32 coarse, u32 *__restrict__ key, u32 *__restrict__ val)
{ u32 j = blockIdx.x * 256u + threadIdx.x;
  if (j >= nwork)
    return;
  const int pbits = pbits0 + dbits;               /* the shift that leaves the read pair */
  const u64 i = work[j], pr = keys[i] >> pbits;
  u64 a = i + 1, b = (j + 1 < nwork) ? (u64) work[j + 1] : nhits;       /* run ends at or before the next head */
  while (a < b)
    { u64 mid = (a + b) >> 1;
      if ((keys[mid] >> pbits) == pr) a = mid + 1; else b = mid;
    }
  u64 n = a - i;
  if (coarse >= 0xfffffffdu)    /* cost = the expected length of the alignment instead of a seed count: the wave kernel's
                                   launch ends with its longest serial chains, so those must start first */
    { const u64 pm = (1ull << pbits0) - 1;
      const u64 ext = ((keys[a - 1] >> dbits) & pm) - ((keys[i] >> dbits) & pm);      /* extent of the seeds on A */
      const u32 ra = (u32) (pr & ((1ull << abits) - 1)), rb = (u32) (pr >> abits);
      const int alen = (int) (aboff[ra + 1] - aboff[ra]) - 1, blen = (int) (bboff[rb + 1] - bboff[rb]) - 1;
      const int d = seed_diag(keys[i], vals, i, pm, dbits);             /* diagonal a - b of the first seed */
      const int geo = min(alen, blen + d) - max(0, d);                  /* overlap of the two reads on it */
      u64 len = (coarse == 0xffffffffu) ? ext : (coarse == 0xfffffffeu ? (u64) max(geo, 0) : max(ext, (u64) max(geo, 0)));
      n = len >> (pbits0 > 16 ? pbits0 - 16 : 0);
      coarse = 0;
    }
  if (n > WORK_COST_MAX)
    n = WORK_COST_MAX;
  if (coarse == 1)              /* size classes (powers of two): largest class first, reference order inside */
    key[j] = (u32) __clz((int) n) - (32 - WORK_COST_BITS);
  else
    key[j] = coarse ? (n >= coarse ? 0u : 1u) : (u32) (WORK_COST_MAX - n);
  val[j] = j;
}

void damar_launch_work_cost(const u64 *keys, const u32 *vals, u64 nhits, int pbits, int abits, int dbits, const u32 *aboff,
                            const u32 *bboff, const u32 *work, u32 nwork, u32 coarse, u32 *key, u32 *val, hipStream_t st)
{ if (nwork == 0)
    return;
  hipLaunchKernelGGL(work_cost, dim3((nwork + 255) / 256), dim3(256), 0, st, keys, vals, nhits, pbits, abits, dbits, aboff, bboff,
                     work, nwork, coarse, key, val);
}

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

SEED_PRIO_VAR(g_merge_prio, 0)
SEED_PRIO_SETTER(damar_merge_set_prio, g_merge_prio)


/* number of entries of bpos[jb,ib) strictly below `bound` (bpos ascending in a run) */
__device__ __forceinline__ u32 count_below(const u32 *__restrict__ bpos, u32 jb, u32 ib, u32 bound)
{ u32 a = jb, b = ib;
  while (a < b)
    { u32 m = (a + b) >> 1;
      if (bpos[m] < bound) a = m + 1; else b = m;
    }
  return a - jb;
}

/***** the merge of two sorted k-mer indexes ***************************************************************
 * filter.c:1039-1165 (count_thread) and :1170-1358 (merge_thread) walk the two sorted lists with two pointers.
 * Here the A index is cut into tiles of MT_A entries; a tile needs exactly the B entries whose codes lie between
 * its first and its last code, a contiguous piece of the B index found once per tile (merge_tiles).  A workgroup
 * streams its A codes into registers (8 consecutive entries per thread) and its B piece into LDS; a thread finds
 * the B run of its first entry by one binary search in LDS and walks on from there -- both lists are sorted, so
 * the walk advances about one B entry per A entry.  No prefix table, no per-entry search in HBM: the two code
 * arrays are read once per sweep, front to back.
 *
 * Two sweeps per comparison: COUNT (hits per tile; the host needs the total before it sizes the seed arrays)
 * and, after a scan over the tile totals, EMIT (the same counts again, then the seed pairs).  The caps of
 * filter.c:1248 / 1335 need whole-run figures -- the length of an A run (cross comparisons) or the hits of a
 * whole A run (self comparisons): runs are delimited by a max-/min-scan over the tile's head flags, and a run
 * that crosses a tile border is completed from what merge_tiles found for the border runs. */

#ifndef MT_PER
#define MT_PER  4                        /* consecutive entries per thread */
#endif
#define MT_A    (MT_PER * 256)           /* A entries per tile            */
#define MT_BCAP (3 * MT_A)               /* B codes staged in LDS per tile (a piece is about as long as the tile) */
#define MT_HITS 4                        /* seed pairs a thread has in flight in the EMIT loop */
#define MT_NBK  2048                     /* buckets over a tile's code range: where a code's entries start in the staged B piece */

template <typename CodeT>
__device__ __forceinline__ u32 lower_bound_c(const CodeT *c, u32 lo, u32 hi, CodeT x)      /* first i in [lo,hi): c[i] >= x */
{ while (lo < hi)
    { const u32 mid = (lo + hi) >> 1;
      if (c[mid] < x) lo = mid + 1; else hi = mid;
    }
  return lo;
}
template <typename CodeT>
__device__ __forceinline__ u32 upper_bound_c(const CodeT *c, u32 lo, u32 hi, CodeT x)      /* first i in [lo,hi): c[i] > x */
{ while (lo < hi)
    { const u32 mid = (lo + hi) >> 1;
      if (c[mid] <= x) lo = mid + 1; else hi = mid;
    }
  return lo;
}

/* One thread per tile: the piece [b0,b1) of B that holds the codes of the tile, the start ja of the A run its first
   entry belongs to and the end ia of the A run of its last entry.
   The two searches in B were 2 x 27 dependent loads from HBM per thread and nothing else -- 95 us per comparison of two
   78 Mbp blocks on a stream whose kernels run one after the other.  A workgroup first brings MT_SMP evenly spaced B codes
   into LDS (one round of independent loads; the lines are the same for every workgroup), searches there, and is left with
   the 13 or 14 steps inside one stretch of blen / MT_SMP entries. */
#define MT_SMP 8192
template <typename CodeT>
__global__ __launch_bounds__(256)
void merge_tiles(MergeArgs m, u32 ntiles, MergeTile *__restrict__ tiles)
{ SEED_PRIO(g_merge_prio);
  constexpr u32 NS = sizeof(CodeT) == 4 ? MT_SMP : MT_SMP / 2;               /* (32 KB of LDS either way) */
  __shared__ CodeT smp[NS];                                  /* smp[j] = the last code of stretch j of B */
  const CodeT *acode = (const CodeT *) m.acode, *bcode = (const CodeT *) m.bcode;
  const u32 stride = (m.blen + NS - 1) / NS;
  if (m.blen > 0)
    for (u32 j = threadIdx.x; j < NS; j += 256)
      { const u64 e = (u64) (j + 1) * stride - 1;
        smp[j] = bcode[e < m.blen ? (u32) e : m.blen - 1];
      }
  __syncthreads();
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= ntiles)
    return;
  const u32 a0 = t * (u32) MT_A, a1 = min(m.alen, a0 + (u32) MT_A);
  const CodeT c0 = acode[a0], c1 = acode[a1 - 1];
  MergeTile tl;
  tl.b0 = tl.b1 = m.blen;
  if (m.blen > 0)
    { u32 lo = 0, hi = NS;                                   /* first stretch whose last code is >= c0: it holds B's first code >= c0 */
      while (lo < hi)
        { const u32 mid = (lo + hi) >> 1;
          if (smp[mid] < c0) lo = mid + 1; else hi = mid;
        }
      u32 l0 = m.blen, h0 = m.blen, l1 = m.blen, h1 = m.blen;
      if (lo < NS)                                           /* (a stretch that starts beyond the end repeats the last code: never the first hit) */
        { const u64 s0 = (u64) lo * stride;
          l0 = (u32) s0;  h0 = (u32) min((u64) m.blen, s0 + stride);
        }
      lo = 0;  hi = NS;                                      /* first stretch whose last code is > c1 */
      while (lo < hi)
        { const u32 mid = (lo + hi) >> 1;
          if (smp[mid] <= c1) lo = mid + 1; else hi = mid;
        }
      if (lo < NS)
        { const u64 s0 = (u64) lo * stride;
          l1 = (u32) s0;  h1 = (u32) min((u64) m.blen, s0 + stride);
        }
      while (l0 < h0 || l1 < h1)                             /* the two searches step together: their loads are in flight together */
        { const u32 m0 = (l0 + h0) >> 1, m1 = (l1 + h1) >> 1;
          const CodeT v0 = bcode[m0 < m.blen ? m0 : m.blen - 1], v1 = bcode[m1 < m.blen ? m1 : m.blen - 1];
          if (l0 < h0)
            { if (v0 < c0) l0 = m0 + 1; else h0 = m0; }
          if (l1 < h1)
            { if (v1 <= c1) l1 = m1 + 1; else h1 = m1; }
        }
      tl.b0 = l0;  tl.b1 = l1;
    }
  /* (a run that crosses a tile border is a few entries long as a rule: looked for within 64 entries first) */
  tl.ja = a0;
  if (a0 > 0 && acode[a0 - 1] == c0)
    { u32 lo = a0 > 64 ? a0 - 64 : 0;
      if (lo > 0 && acode[lo] >= c0)
        lo = 0;
      tl.ja = lower_bound_c<CodeT>(acode, lo, a0, c0);
    }
  tl.ia = a1;
  if (a1 < m.alen && acode[a1] == c1)
    { u32 hi = m.alen - a1 > 64 ? a1 + 64 : m.alen;
      if (hi < m.alen && acode[hi - 1] <= c1)
        hi = m.alen;
      tl.ia = upper_bound_c<CodeT>(acode, a1, hi, c1);
    }
  { const u64 span = (u64) (c1 - c0);                        /* buckets of the tile's code range (merge_sweep_fast) */
    u32 sh = 0;
    while ((span >> sh) >= (u64) MT_NBK)
      sh += 1;
    tl.sh = sh;
  }
  /* pad[0]: set by merge_fast when the tile needs the general kernel; pad[1]: the longest B run whose mutual count with
     the tile's longest possible A run (tl.ia - tl.ja entries) stays below the cap of filter.c:1248 / 1335 */
  { const u32 arun = tl.ia - tl.ja;
    tl.pad[0] = 0;
    tl.pad[1] = (arun == 0) ? 0xffffffffu : (u32) (((u64) m.limit - 1) / (u64) arun);
    tl.pad[2] = 0;
  }
  tiles[t] = tl;
}

/* hits of A entry i (position word p of its k-mer) against the B run bpos[jb, jb+nb) in a self comparison:
   the B entries before `bound` (filter.c:1219-1246).  Block and complement block have the same reads, so A's and B's
   position words are of one kind; packed words (read << rpbits | offset) order like block offsets, and the bounds
   "start of A's read", "end of A's read", "A's own position" need no look-up at all. */
__device__ __forceinline__ u32 self_hits(const MergeArgs &m, u32 p, u32 jb, u32 nb)
{ u32 bound;
  const int rp = m.ablk.rpbits;
  if (rp)
    { const u32 ra = p >> rp;
      bound = m.identity ? (m.comp ? (ra + 1) << rp : p) : ra << rp;
      if (m.identity && m.comp && ((ra + 1) >> (32 - rp)) != 0)        /* (ra + 1) << rp would wrap: every entry is below */
        return nb;
    }
  else if (m.identity)
    bound = m.comp ? m.ablk.boff[read_of_pos(m.ablk, p) + 1] : p;
  else
    bound = m.ablk.boff[read_of_pos(m.ablk, p)];
  return count_below(m.bpos, jb, jb + nb, bound);
}

__device__ __forceinline__ u64 block_sum_u64(u64 v, u64 *red)      /* red: 4 words of LDS */
{ for (int o = 32; o > 0; o >>= 1)
    { const u32 lo = (u32) __shfl_xor((int) (u32) v, o), hi = (u32) __shfl_xor((int) (u32) (v >> 32), o);
      v += ((u64) hi << 32) | lo;
    }
  __syncthreads();
  if (lane_id() == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

/* tcount[tile] = hits of the tile, cnt[i] / jbg[i] = hits of A entry i and where its B run starts (and, with gram, the
   histogram of the runs' mutual counts, filter.c:1039-1165).  INLDS: the tile's B piece fits the LDS stage. */
template <typename CodeT, bool INLDS>
__device__ __forceinline__ void merge_sweep_tile(const MergeArgs &m, const MergeTile tl, const u32 tile,
                                                 u32 *__restrict__ tcount, u32 *__restrict__ cnt, u32 *__restrict__ jbg,
                                                 unsigned long long *__restrict__ gram, u32 ngram,
                                                 CodeT *sb, u32 *loc, u32 *sjb, u32 *sw4, u64 *red)
{ const CodeT *acode = (const CodeT *) m.acode, *bcode = (const CodeT *) m.bcode;
  const int l = lane_id(), w = threadIdx.x >> 6;
  const u32 a0 = tile * (u32) MT_A, a1 = min(m.alen, a0 + (u32) MT_A), nat = a1 - a0;
  const u32 nbt = tl.b1 - tl.b0;
  const u32 e0 = threadIdx.x * MT_PER;                        /* tile-local index of this thread's first entry */
  const int nv = (e0 >= nat) ? 0 : (int) min((u32) MT_PER, nat - e0);

  constexpr u32 VPK = 16 / sizeof(CodeT);                     /* codes per 16-byte vector */
  typedef CodeT VecT __attribute__((ext_vector_type(16 / sizeof(CodeT))));
  CodeT a[MT_PER];
  if (nv == MT_PER)                                            /* (a0 + e0 is a multiple of MT_PER: aligned) */
    {
#pragma unroll
      for (int q = 0; q < MT_PER / (int) VPK; q++)
        { const VecT v = ((const VecT *) (acode + a0 + e0))[q];
#pragma unroll
          for (int e = 0; e < (int) VPK; e++)
            a[q * VPK + e] = v[e];
        }
    }
  else
    {
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        a[k] = (k < nv) ? acode[a0 + e0 + k] : (CodeT) 0;
    }
  /* the tile's piece of B into LDS: 16-byte loads from the aligned address at or below b0, all issued before the first
     is waited for; sb[sh + j] = B[j] */
  const u32 b0a = tl.b0 & ~(VPK - 1), sh = tl.b0 - b0a;
  const CodeT *B;
  if (INLDS)
    { const u32 nvec = (tl.b1 - b0a + VPK - 1) / VPK;
      VecT v[MT_BCAP / (int) VPK / 256 + 1];
#pragma unroll
      for (int r = 0; r < MT_BCAP / (int) VPK / 256 + 1; r++)
        { const u32 x = threadIdx.x + (u32) r * 256;
          if (x < nvec)
            { if (b0a + (x + 1) * VPK <= m.blen)
                v[r] = ((const VecT *) (bcode + b0a))[x];
              else                                               /* the last vector of the index: no read past its end */
                {
#pragma unroll
                  for (int e = 0; e < (int) VPK; e++)
                    v[r][e] = (b0a + x * VPK + e < m.blen) ? bcode[b0a + x * VPK + e] : (CodeT) 0;
                }
            }
        }
#pragma unroll
      for (int r = 0; r < MT_BCAP / (int) VPK / 256 + 1; r++)
        { const u32 x = threadIdx.x + (u32) r * 256;
          if (x < nvec)
            ((VecT *) sb)[x] = v[r];
        }
      B = sb + sh;
    }
  else
    B = bcode + tl.b0;
  /* the code before this thread's first entry decides whether that entry starts a run */
  bool head0 = true;
  { CodeT prev = a[MT_PER - 1];                               /* of the lane below (it holds MT_PER entries whenever this one holds any) */
    if (sizeof(CodeT) == 8)
      { const u32 lo = (u32) __shfl_up((int) (u32) prev, 1), hi = (u32) __shfl_up((int) (u32) ((u64) prev >> 32), 1);
        prev = (CodeT) (((u64) hi << 32) | lo);
      }
    else
      prev = (CodeT) (u32) __shfl_up((int) (u32) prev, 1);
    if (nv > 0)
      { if (e0 == 0)
          head0 = (tl.ja == a0);
        else
          { if (l == 0) prev = acode[a0 + e0 - 1];
            head0 = (prev != a[0]);
          }
      }
  }
  __syncthreads();

  /* ---- B runs of the thread's entries: one search, then a walk ---- */
  u32 jb[MT_PER], nb[MT_PER];
  { u32 p = (nv > 0) ? lower_bound_c<CodeT>(B, 0, nbt, a[0]) : 0;
#pragma unroll
    for (int k = 0; k < MT_PER; k++)
      { jb[k] = 0;  nb[k] = 0;
        if (k < nv)
          { const CodeT c = a[k];
            if (k > 0 && c == a[k - 1])
              { jb[k] = jb[k - 1];  nb[k] = nb[k - 1]; }
            else
              { int steps = 0;
                while (p < nbt && B[p] < c)
                  { p += 1;
                    if (++steps == 4) { p = lower_bound_c<CodeT>(B, p, nbt, c);  break; }
                  }
                u32 q = p;
                steps = 0;
                while (q < nbt && B[q] == c)
                  { q += 1;
                    if (++steps == 4) { q = upper_bound_c<CodeT>(B, q, nbt, c);  break; }
                  }
                jb[k] = p;  nb[k] = q - p;
                p = q;
              }
          }
      }
  }

  /* ---- A runs: rs = start of the entry's run, re = its end, as tile-local indices (rs 0 with no head at or
          before the entry: the run began in an earlier tile; re nat with no head after it: it may go on) ---- */
  u32 hbits = 0;                                              /* head flags of the thread's entries */
#pragma unroll
  for (int k = 0; k < MT_PER; k++)
    { const bool h = (k >= nv) ? true : (k == 0 ? head0 : a[k] != a[k - 1]);
      hbits |= (u32) h << k;
    }
  u32 rs[MT_PER], re[MT_PER];
  bool rs_open[MT_PER];                                       /* no head at or before the entry inside the tile */
  { /* forward: latest head at or before the entry (1-based, 0 = none) */
    u32 tmax = 0;
#pragma unroll
    for (int k = 0; k < MT_PER; k++)
      if ((hbits >> k) & 1) tmax = e0 + k + 1;
    u32 x = tmax;
    for (int o = 1; o < 64; o <<= 1)
      { const u32 t = (u32) __shfl_up((int) x, o);
        if (l >= o) x = max(x, t);
      }
    if (l == 63) sw4[w] = x;
    u32 carry = (u32) __shfl_up((int) x, 1);
    if (l == 0) carry = 0;
    __syncthreads();
    for (int i = 0; i < 4; i++)
      if (i < w) carry = max(carry, sw4[i]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MT_PER; k++)
      { if ((hbits >> k) & 1) carry = e0 + k + 1;
        rs_open[k] = (carry == 0);
        rs[k] = carry ? carry - 1 : 0;
      }
    /* backward: first head after the entry (none: nat) */
    const u32 NONE = 0xffffffffu;
    u32 tmin = NONE;
#pragma unroll
    for (int k = MT_PER - 1; k >= 0; k--)
      if ((hbits >> k) & 1) tmin = e0 + k;
    x = tmin;
    for (int o = 1; o < 64; o <<= 1)
      { const u32 t = (u32) __shfl_down((int) x, o);
        if (l + o < 64) x = min(x, t);
      }
    if (l == 0) sw4[w] = x;
    u32 cb = (u32) __shfl_down((int) x, 1);
    if (l == 63) cb = NONE;
    __syncthreads();
    for (int i = 0; i < 4; i++)
      if (i > w) cb = min(cb, sw4[i]);
    __syncthreads();
#pragma unroll
    for (int k = MT_PER - 1; k >= 0; k--)
      { re[k] = (cb == NONE || cb > nat) ? nat : cb;
        if ((hbits >> k) & 1) cb = e0 + k;
      }
  }

  /* ---- hits per entry ---- */
  u32 n[MT_PER];
#pragma unroll
  for (int k = 0; k < MT_PER; k++)
    { n[k] = 0;
      if (k < nv && nb[k] > 0)
        { if (!m.self)
            { const u32 ja = rs_open[k] ? tl.ja : a0 + rs[k];
              const u32 ia = (re[k] >= nat) ? tl.ia : a0 + re[k];                 /* filter.c:1334-1335 */
              if ((u64) (ia - ja) * (u64) nb[k] < (u64) m.limit)
                n[k] = nb[k];
            }
          else
            n[k] = self_hits(m, m.apos[a0 + e0 + k], tl.b0 + jb[k], nb[k]);
        }
    }

  /* prefix of the hits inside the tile: loc[i] = hits of the entries before i */
  u64 T64;
  for (int round = 0; ; round++)
    { u32 s = 0;
      u64 s64 = 0;
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        { s += n[k];  s64 += n[k]; }
      u32 inc = (u32) wave_incl_scan_i((int) s);
      if (l == 63) sw4[w] = inc;
      __syncthreads();
      u32 ex = inc - s;
      for (int i = 0; i < 4; i++)
        if (i < w) ex += sw4[i];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        { loc[e0 + k] = ex;
          ex += n[k];
        }
      if (threadIdx.x == 255)
        loc[MT_A] = ex;
      T64 = block_sum_u64(s64, red);                          /* (its barriers also publish loc) */
      if (!m.self || round == 1)
        break;
      /* self comparison: a run whose hits reach the cap contributes nothing (filter.c:1248).  The run's hits inside
         the tile come from the prefix; the part of a border run that lies in another tile is counted here again. */
      u64 xhead = 0, xtail = 0;
      if (tl.ja < a0)
        { if (threadIdx.x == 0) { sjb[0] = jb[0];  sjb[1] = nb[0]; }
          __syncthreads();
          const u32 j0 = tl.b0 + sjb[0], nn = sjb[1];
          u64 acc = 0;
          if (nn > 0)
            for (u32 i = tl.ja + threadIdx.x; i < a0; i += 256)
              acc += self_hits(m, m.apos[i], j0, nn);
          xhead = block_sum_u64(acc, red);
        }
      if (tl.ia > a1)
        { const u32 le = nat - 1;
          if (threadIdx.x == le / MT_PER)
            {
#pragma unroll
              for (int k = 0; k < MT_PER; k++)
                if (k == (int) (le % MT_PER)) { sjb[0] = jb[k];  sjb[1] = nb[k]; }
            }
          __syncthreads();
          const u32 j0 = tl.b0 + sjb[0], nn = sjb[1];
          u64 acc = 0;
          if (nn > 0)
            for (u32 i = a1 + threadIdx.x; i < tl.ia; i += 256)
              acc += self_hits(m, m.apos[i], j0, nn);
          xtail = block_sum_u64(acc, red);
        }
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        if (k < nv && nb[k] > 0)
          { u64 tot = (u64) (loc[re[k]] - loc[rs[k]]);
            if (rs_open[k]) tot += xhead;
            if (re[k] >= nat) tot += xtail;
            if (gram != NULL && ((hbits >> k) & 1) && tot < (u64) m.limit && tot < (u64) ngram)
              atomicAdd(&gram[tot], 1ull);                    /* one count per run, by its first entry */
            if (tot >= (u64) m.limit)
              n[k] = 0;
          }
      __syncthreads();
    }

  if (!m.self && gram != NULL)
    {
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        if (k < nv && nb[k] > 0 && ((hbits >> k) & 1))
          { const u32 ia = (re[k] >= nat) ? tl.ia : a0 + re[k];
            const u64 ct = (u64) (ia - (a0 + e0 + k)) * (u64) nb[k];
            if (ct < (u64) ngram)
              atomicAdd(&gram[ct], 1ull);
          }
    }
  if (threadIdx.x == 0)
    tcount[tile] = (T64 > 0xffffffffull) ? 0xffffffffu : (u32) T64;
  /* what the EMIT pass needs of this sweep: 8 bytes per A entry instead of a second walk */
  typedef u32 V4 __attribute__((ext_vector_type(4)));
  if (nv == MT_PER && MT_PER == 4)
    { V4 vc, vj;
#pragma unroll
      for (int k = 0; k < 4; k++)
        { vc[k] = n[k];  vj[k] = tl.b0 + jb[k]; }
      *(V4 *) (cnt + a0 + e0) = vc;
      *(V4 *) (jbg + a0 + e0) = vj;
    }
  else
    {
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        if (k < nv)
          { cnt[a0 + e0 + k] = n[k];
            jbg[a0 + e0 + k] = tl.b0 + jb[k];
          }
    }
}

/* The common tile in ONE pass over LDS, without a search, as a kernel of its own (merge_fast): the tile's B piece is staged
 * as in the general kernel, and a table over the tile's code range -- MT_NBK buckets of 2^sh codes (merge_tiles chose sh
 * so that the range fits), first[bucket] = where the bucket's entries start in the piece, filled by the B entries that
 * open a bucket -- answers "where would code c sit" with one look-up and a walk over the bucket's few entries (both
 * indexes sample the same code space: about one B entry per A entry, one or two per bucket).  An A code whose bucket is
 * empty has no partner.  That replaces a binary search of ~10 dependent LDS reads per thread.
 * The caps of filter.c:1248 / 1335 compare a run's mutual count with `limit` (10000 unless memory is short): a count is at
 * most (length of the A run) x (length of the B run), and an A run of this tile is no longer than tl.ia - tl.ja, so when
 * no entry of the tile has a B run longer than tl.pad[1] (merge_tiles) nothing can be capped and neither the run
 * boundaries nor the per-run sums are needed.  Otherwise -- or when the piece does not fit the stage -- the tile is
 * flagged (tiles[].pad[0]) and left to the general kernel, which runs behind this one over the flagged tiles only.
 * Round 5, measured (profiles/r05_seq_kernel_stats.csv): the general kernel alone 1.00 ms per 135 M x 135 M comparison,
 * with the table inside it 0.85 (it stayed at 590 vector + 350 scalar instructions per wavefront: the kernel's many
 * arguments -- two block descriptors -- live in scalar registers that spill), this kernel: see DESIGN.md section 3. */
static_assert(MT_BCAP + 16 < (1 << 16), "a packed entry holds the piece-relative start and the hits in 16 bits each");
struct MergeFastArgs
{ const void *acode, *bcode;
  const u32  *apos, *bpos;
  u32 alen, blen;
  int self, identity, comp, rpbits;          /* rpbits: position words are read << rpbits | offset (self comparisons need it here) */
};

template <typename CodeT>
__global__ __launch_bounds__(256, 8)
void merge_fast(MergeFastArgs m, MergeTile *__restrict__ tiles, u32 *__restrict__ tcount, u32 *__restrict__ cnt)
{ SEED_PRIO(g_merge_prio);
  __shared__ __attribute__((aligned(16))) CodeT sb[MT_BCAP + 16 / sizeof(CodeT)];
  __shared__ __attribute__((aligned(16))) u16 first[MT_NBK];
  __shared__ u32 wsum[4];
  __shared__ u32 deep;
  const CodeT *acode = (const CodeT *) m.acode, *bcode = (const CodeT *) m.bcode;
  const u32 tile = blockIdx.x;
  const MergeTile tl = tiles[tile];
  const u32 a0 = tile * (u32) MT_A, a1 = min(m.alen, a0 + (u32) MT_A), nat = a1 - a0;
  const u32 nbt = tl.b1 - tl.b0;
  if (nbt > MT_BCAP || (m.self && m.rpbits == 0))
    { if (threadIdx.x == 0)
        tiles[tile].pad[0] = 1;
      return;
    }
  const u32 e0 = threadIdx.x * MT_PER;
  const int nv = (e0 >= nat) ? 0 : (int) min((u32) MT_PER, nat - e0);
  const u32 sh = tl.sh, nmax = tl.pad[1];
  constexpr u32 VPK = 16 / sizeof(CodeT);
  typedef CodeT VecT __attribute__((ext_vector_type(16 / sizeof(CodeT))));
  typedef u32 V4 __attribute__((ext_vector_type(4)));

  const CodeT c0 = acode[a0];
  CodeT a[MT_PER];
  if (nv == MT_PER)
    {
#pragma unroll
      for (int q = 0; q < MT_PER / (int) VPK; q++)
        { const VecT v = ((const VecT *) (acode + a0 + e0))[q];
#pragma unroll
          for (int e = 0; e < (int) VPK; e++)
            a[q * VPK + e] = v[e];
        }
    }
  else
    {
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        a[k] = (k < nv) ? acode[a0 + e0 + k] : (CodeT) 0;
    }
  /* the B piece into LDS (16-byte loads from the aligned address at or below b0, all in flight together), the table to "empty" */
  const u32 b0a = tl.b0 & ~(VPK - 1), shf = tl.b0 - b0a;
  { const u32 nvec = (tl.b1 - b0a + VPK - 1) / VPK;
    const u32 whole = (m.blen - b0a) / VPK;                    /* vectors that lie inside the index */
    VecT v[MT_BCAP / (int) VPK / 256 + 1];
#pragma unroll
    for (int r = 0; r < MT_BCAP / (int) VPK / 256 + 1; r++)
      { const u32 x = threadIdx.x + (u32) r * 256;
        if (x < nvec)
          { if (x < whole)
              v[r] = ((const VecT *) (bcode + b0a))[x];
            else
              {
#pragma unroll
                for (int e = 0; e < (int) VPK; e++)
                  v[r][e] = (b0a + x * VPK + e < m.blen) ? bcode[b0a + x * VPK + e] : (CodeT) 0;
              }
          }
      }
    { const V4 ones = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
      ((V4 *) first)[threadIdx.x] = ones;                     /* MT_NBK u16 = 256 x 16 bytes */
    }
    if (threadIdx.x == 0)
      deep = 0;
#pragma unroll
    for (int r = 0; r < MT_BCAP / (int) VPK / 256 + 1; r++)
      { const u32 x = threadIdx.x + (u32) r * 256;
        if (x < nvec)
          ((VecT *) sb)[x] = v[r];
      }
  }
  const CodeT *B = sb + shf;
  __syncthreads();
  /* every B entry that opens a bucket says where (the piece's codes lie in [c0, last code of the tile]) */
  for (u32 j = threadIdx.x; j < nbt; j += 256)
    { const u32 bk = (u32) ((B[j] - c0) >> sh);
      if (j == 0 || (u32) ((B[j - 1] - c0) >> sh) != bk)
        first[bk] = (u16) j;
    }
  __syncthreads();

  u32 jb[MT_PER], n[MT_PER];
  u32 longest = 0;
#pragma unroll
  for (int k = 0; k < MT_PER; k++)
    { jb[k] = 0;  n[k] = 0;
      if (k < nv)
        { const CodeT c = a[k];
          if (k > 0 && c == a[k - 1])
            { jb[k] = jb[k - 1];  n[k] = n[k - 1]; }
          else
            { u32 p = first[(u32) ((c - c0) >> sh)];
              if (p != 0xffffu)
                { int steps = 0;
                  while (p < nbt && B[p] < c)
                    { p += 1;
                      if (++steps == 4) { p = lower_bound_c<CodeT>(B, p, nbt, c);  break; }
                    }
                  u32 q = p;
                  steps = 0;
                  while (q < nbt && B[q] == c)
                    { q += 1;
                      if (++steps == 4) { q = upper_bound_c<CodeT>(B, q, nbt, c);  break; }
                    }
                  jb[k] = p;  n[k] = q - p;
                }
            }
          longest = max(longest, n[k]);
        }
    }
  if (longest > nmax)
    deep = 1;                                                 /* a cap might apply: the general kernel decides */
  /* n[] holds the length of the B run so far: in a self comparison only the entries before A's bound count
     (filter.c:1219-1246; packed position words order like block offsets, the bounds need no look-up) */
  u32 sum = 0;
  if (m.self)
    { const int rp = m.rpbits;
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        if (n[k] > 0)
          { const u32 p = m.apos[a0 + e0 + k], ra = p >> rp;
            u32 bound = m.identity ? (m.comp ? (ra + 1) << rp : p) : ra << rp;
            if (!(m.identity && m.comp && ((ra + 1) >> (32 - rp)) != 0))      /* (else (ra + 1) << rp would wrap: every entry is below) */
              n[k] = count_below(m.bpos, tl.b0 + jb[k], tl.b0 + jb[k] + n[k], bound);
          }
    }
#pragma unroll
  for (int k = 0; k < MT_PER; k++)
    sum += n[k];
  sum = (u32) wave_sum_i((int) sum);
  if (lane_id() == 0)
    wsum[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (deep)
    { if (threadIdx.x == 0)
        tiles[tile].pad[0] = 1;
      return;
    }
  if (threadIdx.x == 0)
    { tcount[tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];   /* (at most MT_A x MT_BCAP: no overflow) */
      tiles[tile].pad[2] = 1;                                 /* done here: its entries are PACKED */
    }
  /* what the EMIT pass needs of an entry, in ONE word: hits << 16 | start of its B run inside the tile's piece (both below
     MT_BCAP + 4 < 2^16) -- 4 bytes written and read per A entry instead of 8 (the general kernel keeps two arrays: its
     pieces and counts know no such bound) */
  if (nv == MT_PER && MT_PER == 4)
    { V4 vc;
#pragma unroll
      for (int k = 0; k < 4; k++)
        vc[k] = (n[k] << 16) | jb[k];
      *(V4 *) (cnt + a0 + e0) = vc;
    }
  else
    {
#pragma unroll
      for (int k = 0; k < MT_PER; k++)
        if (k < nv)
          cnt[a0 + e0 + k] = (n[k] << 16) | jb[k];
    }
}

template <typename CodeT>
__global__ __launch_bounds__(256, 8)          /* <= 64 VGPRs: two wavefronts per SIMD still find room beside a resident report launch */
void merge_sweep(MergeArgs m, const MergeTile *__restrict__ tiles, u32 *__restrict__ tcount, u32 *__restrict__ cnt,
                 u32 *__restrict__ jbg, unsigned long long *__restrict__ gram, u32 ngram, int only_flagged)
{ SEED_PRIO(g_merge_prio);
  __shared__ __attribute__((aligned(16))) CodeT sb[MT_BCAP + 16 / sizeof(CodeT)];
  __shared__ u32 loc[MT_A + 1];
  __shared__ u32 sjb[2];
  __shared__ u32 sw4[4];
  __shared__ u64 red[4];
  const MergeTile tl = tiles[blockIdx.x];
  if (only_flagged && tl.pad[0] == 0)          /* merge_fast has done this tile */
    return;
  if (tl.b1 - tl.b0 <= MT_BCAP)                /* (the stage has room for the alignment slack on top) */
    merge_sweep_tile<CodeT, true>(m, tl, blockIdx.x, tcount, cnt, jbg, gram, ngram, sb, loc, sjb, sw4, red);
  else
    merge_sweep_tile<CodeT, false>(m, tl, blockIdx.x, tcount, cnt, jbg, gram, ngram, sb, loc, sjb, sw4, red);
}

/* EMIT: one workgroup per tile of A entries; the tile's hit counts are scanned in LDS, then the tile's hits are dealt
 * out to the threads in order, each finding its A entry by a search of the LDS prefix (no walk over global offsets),
 * so that the seed pairs leave in fully coalesced runs; MT_HITS independent seed pairs per thread are in flight. */
__global__ __launch_bounds__(256, 8)
void merge_emit(MergeArgs m, const MergeTile *__restrict__ tiles, const u32 *__restrict__ cnt, const u32 *__restrict__ jbg,
                const u32 *__restrict__ toff, u64 nhits, u64 *__restrict__ keys, u32 *__restrict__ vals, u32 *__restrict__ pid)
{ SEED_PRIO(g_merge_prio);
  __shared__ u32 loc[MT_A + 1];
  __shared__ u32 sjb[MT_A];
  __shared__ u32 sw4[4];
  const int l = lane_id(), w = threadIdx.x >> 6;
  const u32 tile = blockIdx.x;
  const u32 a0 = tile * (u32) MT_A, a1 = min(m.alen, a0 + (u32) MT_A), nat = a1 - a0;
  const u32 e0 = threadIdx.x * MT_PER;
  const u64 h0 = toff[tile];
  const u64 hn = (tile + 1 < gridDim.x) ? (u64) toff[tile + 1] : nhits;
  if (hn == h0)                                               /* nothing to emit here */
    return;
  u32 n[MT_PER], s = 0;
  const bool packed = tiles[tile].pad[2] != 0;                /* merge_fast's tile: hits << 16 | start inside the piece, in cnt[] */
  const u32 pb0 = tiles[tile].b0;
#pragma unroll
  for (int k = 0; k < MT_PER; k++)
    { const bool ok = e0 + k < nat;
      const u32 w = ok ? cnt[a0 + e0 + k] : 0;
      n[k] = packed ? w >> 16 : w;
      sjb[e0 + k] = !ok ? 0 : packed ? pb0 + (w & 0xffffu) : jbg[a0 + e0 + k];
      s += n[k];
    }
  u32 inc = (u32) wave_incl_scan_i((int) s);
  if (l == 63) sw4[w] = inc;
  __syncthreads();
  u32 ex = inc - s;
  for (int i = 0; i < 4; i++)
    if (i < w) ex += sw4[i];
#pragma unroll
  for (int k = 0; k < MT_PER; k++)
    { loc[e0 + k] = ex;
      ex += n[k];
    }
  if (threadIdx.x == 255)
    loc[MT_A] = ex;
  __syncthreads();
  const u32 T = loc[MT_A];
  for (u32 t0 = threadIdx.x; t0 < T; t0 += 256 * MT_HITS)
    { u32 ai[MT_HITS], bi[MT_HITS], pa[MT_HITS], pb[MT_HITS];
#pragma unroll
      for (int x = 0; x < MT_HITS; x++)                       /* (the MT_HITS chains are independent: their latencies overlap) */
        { const u32 t = t0 + (u32) x * 256;
          u32 lo = 0, hi = MT_A;                              /* last entry whose first hit is <= t */
          if (t < T)
            while (hi - lo > 1)
              { const u32 mid = (lo + hi) >> 1;
                if (loc[mid] <= t) lo = mid; else hi = mid;
              }
          ai[x] = a0 + lo;
          bi[x] = (t < T) ? sjb[lo] + (t - loc[lo]) : 0;
        }
#pragma unroll
      for (int x = 0; x < MT_HITS; x++)
        { const bool ok = t0 + (u32) x * 256 < T;
          pa[x] = ok ? m.apos[ai[x]] : 0;
          pb[x] = ok ? m.bpos[bi[x]] : 0;
        }
      u32 ra[MT_HITS], rb[MT_HITS], xa[MT_HITS], xb[MT_HITS];
      if (m.ablk.rpbits && m.bblk.rpbits)                     /* packed position words: nothing to look up */
        {
#pragma unroll
          for (int x = 0; x < MT_HITS; x++)
            { ra[x] = pa[x] >> m.ablk.rpbits;  xa[x] = pa[x] & ((1u << m.ablk.rpbits) - 1u);
              rb[x] = pb[x] >> m.bblk.rpbits;  xb[x] = pb[x] & ((1u << m.bblk.rpbits) - 1u);
            }
        }
      else
        {
#pragma unroll
          for (int x = 0; x < MT_HITS; x++)
            { pos_decode(m.ablk, pa[x], &ra[x], &xa[x]);
              pos_decode(m.bblk, pb[x], &rb[x], &xb[x]);
            }
        }
#pragma unroll
      for (int x = 0; x < MT_HITS; x++)
        { const u32 t = t0 + (u32) x * 256;
          const u64 h = h0 + t;
          if (t < T && h < nhits)
            { const u64 key = ((u64) rb[x] << (m.abits + m.pbits)) | ((u64) ra[x] << m.pbits) | (u64) xa[x];
              if (m.dbits)
                keys[h] = (key << m.dbits) | (u64) xb[x];
              else
                { keys[h] = key;
                  vals[h] = (u32) ((int) xa[x] - (int) xb[x]);
                }
              if (pid != NULL)                 /* the read pair alone, for the early cut (damar_launch_pair_cut) */
                pid[h] = (rb[x] << m.abits) | ra[x];
            }
        }
    }
}

/* workspace: tile descriptors | tile counts (scanned in place by the caller) | cnt[alen] | jb[alen] */
static size_t mw_tiles(u32 alen)  { return ((size_t) alen + MT_A - 1) / MT_A; }
static size_t mw_off_counts(u32 alen) { return (mw_tiles(alen) * sizeof(MergeTile) + 255) & ~(size_t) 255; }
static size_t mw_off_cnt(u32 alen)    { return mw_off_counts(alen) + (((mw_tiles(alen) + 1) * sizeof(u32) + 255) & ~(size_t) 255); }
static size_t mw_off_jb(u32 alen)     { return mw_off_cnt(alen) + (((size_t) alen * sizeof(u32) + 255) & ~(size_t) 255); }

size_t damar_merge_workspace_bytes(u32 alen)
{ return mw_off_jb(alen) + (((size_t) alen * sizeof(u32) + 255) & ~(size_t) 255); }

u32 *damar_merge_tile_counts(void *work, u32 alen) { return (u32 *) ((char *) work + mw_off_counts(alen)); }

u32 damar_merge_tiles(u32 alen) { return (u32) mw_tiles(alen); }

/* COUNT sweep: hits per tile and per A entry into the workspace; with gram != NULL also the histogram of mutual counts */
void damar_launch_merge_count(const MergeArgs *m, void *work, unsigned long long *gram, u32 ngram, hipStream_t st)
{ if (m->alen == 0)
    return;
  const u32 ntiles = (u32) mw_tiles(m->alen);
  MergeTile *tiles = (MergeTile *) work;
  u32 *tcount = damar_merge_tile_counts(work, m->alen);
  u32 *cnt = (u32 *) ((char *) work + mw_off_cnt(m->alen)), *jb = (u32 *) ((char *) work + mw_off_jb(m->alen));
  static int general = -1;                   /* test hook (tests/test_gpu_parity.py): every tile the general way */
  if (general < 0)
    general = getenv("DAMAR_MERGE_GENERAL") != NULL;
  const bool fast = gram == NULL && !general;  /* (the run histogram is the general kernel's) */
  MergeFastArgs f;
  f.acode = m->acode;  f.bcode = m->bcode;  f.apos = m->apos;  f.bpos = m->bpos;  f.alen = m->alen;  f.blen = m->blen;
  f.self = m->self;  f.identity = m->identity;  f.comp = m->comp;  f.rpbits = m->ablk.rpbits;
  if (m->wide)
    { hipLaunchKernelGGL(merge_tiles<u64>, dim3((ntiles + 255) / 256), dim3(256), 0, st, *m, ntiles, tiles);
      if (fast)
        hipLaunchKernelGGL(merge_fast<u64>, dim3(ntiles), dim3(256), 0, st, f, tiles, tcount, cnt);
      hipLaunchKernelGGL(merge_sweep<u64>, dim3(ntiles), dim3(256), 0, st, *m, tiles, tcount, cnt, jb, gram, ngram, fast ? 1 : 0);
    }
  else
    { hipLaunchKernelGGL(merge_tiles<u32>, dim3((ntiles + 255) / 256), dim3(256), 0, st, *m, ntiles, tiles);
      if (fast)
        hipLaunchKernelGGL(merge_fast<u32>, dim3(ntiles), dim3(256), 0, st, f, tiles, tcount, cnt);
      hipLaunchKernelGGL(merge_sweep<u32>, dim3(ntiles), dim3(256), 0, st, *m, tiles, tcount, cnt, jb, gram, ngram, fast ? 1 : 0);
    }
}

/* EMIT, after the tile counts have been scanned in place */
void damar_launch_merge_emit(const MergeArgs *m, void *work, u64 nhits, u64 *keys, u32 *vals, u32 *pid, hipStream_t st)
{ if (nhits == 0)
    return;
  const u32 ntiles = (u32) mw_tiles(m->alen);
  const u32 *toff = damar_merge_tile_counts(work, m->alen);
  const u32 *cnt = (const u32 *) ((char *) work + mw_off_cnt(m->alen)), *jb = (const u32 *) ((char *) work + mw_off_jb(m->alen));
  hipLaunchKernelGGL(merge_emit, dim3(ntiles), dim3(256), 0, st, *m, (const MergeTile *) work, cnt, jb, toff, nhits, keys, vals, pid);
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
{ SEED_PRIO(g_merge_prio);
  __shared__ u32 wsum[4];
  const int nthr = nshift < 0 ? 0 : 1 << nshift;       /* nshift < 0: no slices (the seeds went through the early cut) */
  const int l = lane_id(), w = threadIdx.x >> 6;
  const u64 base = (u64) blockIdx.x * DAMAR_SCAN_TILE;
  u32 mine = 0;
  /* the slice ends that can matter to a seed of this tile: the first one beyond the tile's start and the one after it
     (found once per workgroup; the walk over all of them per run head was most of this kernel's scalar instructions) */
  u64 e0 = ~0ull, e1 = ~0ull;
  for (int t = 0; t < nthr; t++)
    { const u64 e = send[t];
      if (e > base)
        { e0 = e;
          if (t + 1 < nthr)
            e1 = send[t + 1];
          break;
        }
    }
  /* four rounds' loads are issued before the first is looked at (from clamped addresses, so that none sits behind a
     branch): one round at a time the workgroup waited out a memory round trip per round, 16 per tile */
  for (int r0 = 0; r0 < PH_ROUNDS; r0 += 4)
    { K k0[4], km[4], kp[4];
#pragma unroll
      for (int q = 0; q < 4; q++)
        { const u64 i = base + (u64) (r0 + q) * 256u + threadIdx.x;
          const u64 ic = i < nhits ? i : nhits - 1;
          const u64 ip = ic + (u64) (minhit - 1);
          k0[q] = keys[ic];
          km[q] = keys[ic > 0 ? ic - 1 : 0];
          kp[q] = keys[ip < nhits ? ip : nhits - 1];
        }
#pragma unroll
      for (int q = 0; q < 4; q++)
        { const int r = r0 + q;
          const u64 i = base + (u64) r * 256u + threadIdx.x;
          bool f = false;
          if (i < nhits)
            { const K pr = k0[q] >> pbits;
              if ((i == 0 || (km[q] >> pbits) != pr) && i + (u64) (minhit - 1) < nhits && (kp[q] >> pbits) == pr)
                { f = true;
                  if (i < e0)
                    { if (i + (u64) minhit >= e0) f = false; }
                  else if (i < e1)
                    { if (i + (u64) minhit >= e1) f = false; }
                  else                           /* (three slice ends inside one tile: slices of a few hundred seeds) */
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
{ SEED_PRIO(g_merge_prio);
  u32 t = blockIdx.x * 256u + threadIdx.x;
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
      int dlast = 0x7fffffff;
      for (int x = 0; x < n && !ok; x++)
        { const int dx = seed_diag(keys[i + (u64) x], vals, i + (u64) x, pmask, dbits) >> binshift;
          if (dx == dlast)                       /* the sums depend on the bucket only: the seed before this one had them */
            continue;
          dlast = dx;
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

/* Run heads and their screen in ONE pass over the sorted seeds (round 6).  pair_heads_mark + pair_screen above read the
 * keys twice -- the second time one thread per head, from wherever the head's run lies -- and the heads went through a
 * list, a flag per head, a second device-wide scan and a compaction before they were the work list.  Here a workgroup
 * brings its tile of DAMAR_SCAN_TILE keys (and the 64 behind it: a screened run has at most SCREEN_MAX seeds) into LDS
 * once, finds the heads there (same predicate, same slice rule), gathers them in an LDS list so that the screen runs
 * with full wavefronts, screens each from LDS, and leaves ONE bit per seed: head AND kept.  The work list is then
 * pair_heads_expand of those bit words after the scan over the tile counts: no head list, no flags, no second scan, no
 * compaction, and one host synchronisation per comparison instead of two. */
#define PW_HALO 64
#define PW_SMALL 8                                       /* runs of up to this many seeds are screened by one lane */
#define PW_SHORT 4                                       /* ... and those of up to this many in wavefronts of their own */

/* The screen of a run of n <= N seeds of the packed layout, in registers: seed y's bucket and A position are unpacked once
 * and everything is unrolled and predicated.  First what seed y adds to its bucket's sum, min(kmer, apos - apos of the seed
 * before it in the same bucket) (two instructions per earlier seed), then for every seed x the sum of what the seeds of its
 * bucket and of the next one add (four per pair) -- the loops over x and y with their LDS reads, unpacking and branches were
 * 2 000 instructions per wavefront for the longest run among its 64 heads.  run[q] = key of the run's seed q; lanes with
 * n = 0 take part and return false. */
template <int N, bool UNS>
__device__ __forceinline__ bool pw_screen_small(const u64 *run, int n, u64 pmask, int dbits, int binshift, int kmer, int hitmin)
{ int d[N], c[N];                                        /* bucket and A position of seed q */
  const u64 dmask = (1ull << dbits) - 1;
  int top = 0;                                           /* (UNS) the run's largest A position */
#pragma unroll
  for (int q = 0; q < N; q++)
    { const u64 k = run[q < n ? q : 0];
      c[q] = (int) ((k >> dbits) & pmask);
      d[q] = (q < n) ? ((c[q] - (int) (k & dmask)) >> binshift) : 0x40000000 + 4 * q;       /* (a bucket of its own) */
      if (UNS)
        top = (q < n && c[q] > top) ? c[q] : top;
    }
  /* contributions: add[q] = min(kmer, ap[q] - ap[last y < q in the same bucket, or 0]) */
  int add[N];
#pragma unroll
  for (int q = 0; q < N; q++)
    { int prev = 0;
      if (UNS)
        { /* the seeds of a run in no particular order (the seed sort went over the read pair only): the seed before q in
             its bucket is the one with the largest A position among those that order before q (position, then place) */
#pragma unroll
          for (int y = 0; y < N; y++)
            if (y != q)
              { const bool before = (y < q) ? c[y] <= c[q] : c[y] < c[q];
                prev = (d[y] == d[q] && before && c[y] > prev) ? c[y] : prev;
              }
        }
      else
        {
#pragma unroll
          for (int y = 0; y < q; y++)
            prev = (d[y] == d[q]) ? c[y] : prev;
        }
      add[q] = min(c[q] - prev, kmer);
    }
  if (UNS && top > SCREEN_PANEL)                         /* beyond the first panel: not screened (pair_screen), kept */
    return true;
  bool ok = false;
#pragma unroll
  for (int x = 0; x < N; x++)
    { int sum = 0;
#pragma unroll
      for (int y = 0; y < N; y++)
        sum += ((u32) (d[y] - d[x]) <= 1u) ? add[y] : 0;
      ok |= (x < n) && (sum >= hitmin);
    }
  return ok;
}

template <bool UNS>                                      /* UNS: the seeds of a read pair's run are in no particular order */
__global__ __launch_bounds__(256)
void pair_work_mark(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u64 nhits, int ppos, int dbits, int abits,
                    int minhit, int nshift, const u64 *__restrict__ send, int binshift, int kmer, int hitmin,
                    u32 b_lo, u32 b_hi, u64 *__restrict__ bits, u32 *__restrict__ tcount)
{ SEED_PRIO(g_merge_prio);
  __shared__ u64 sk[DAMAR_SCAN_TILE + PW_HALO + 1];     /* sk[j] = keys[base + j - 1] */
  __shared__ u64 bb[(PH_ROUNDS + 1) * 4];               /* bit j: seed j of the tile (or of the halo) starts a read pair's run */
  __shared__ u16 hl[DAMAR_SCAN_TILE];                   /* the tile's heads (offsets into the tile), in no particular order */
  __shared__ u16 cl[DAMAR_SCAN_TILE];                   /* heads to screen, offset | seeds << 12: runs of up to PW_SHORT seeds
                                                           from the front, of up to PW_SMALL from the back */
  __shared__ u32 kb[DAMAR_SCAN_TILE / 32];              /* head AND kept, one bit per seed of the tile */
  __shared__ u32 bl[DAMAR_SCAN_TILE / PW_SMALL];        /* heads of runs of more than PW_SMALL seeds: offset | seeds << 16 */
  __shared__ u32 nh, nlo, nhi, nb;
  const int  pshift = ppos + dbits;                     /* the shift that leaves the read pair */
  const u64  pmask = (1ull << ppos) - 1;
  const int  nthr = nshift < 0 ? 0 : 1 << nshift;       /* nshift < 0: no slices (the seeds went through the early cut) */
  const int  l = lane_id(), w = threadIdx.x >> 6;
  const u64  base = (u64) blockIdx.x * DAMAR_SCAN_TILE;
  { /* all of a thread's 17 loads are issued before the first is stored (from clamped addresses, none behind a branch): one
       at a time the workgroup waited out 17 memory round trips, and three workgroups per CU do not hide that */
    constexpr int NLD = (DAMAR_SCAN_TILE + PW_HALO + 1 + 255) / 256;
    u64 v[NLD];
#pragma unroll
    for (int q = 0; q < NLD; q++)
      { u64 i = base + (u64) q * 256u + threadIdx.x;
        i = i > 0 ? i - 1 : 0;
        v[q] = keys[i < nhits ? i : nhits - 1];         /* (what lies beyond the last seed is never looked at) */
      }
#pragma unroll
    for (int q = 0; q < NLD; q++)
      { const u32 j = (u32) q * 256u + threadIdx.x;
        if (j < DAMAR_SCAN_TILE + PW_HALO + 1)
          sk[j] = v[q];
      }
  }
  if (threadIdx.x < DAMAR_SCAN_TILE / 32)
    kb[threadIdx.x] = 0;
  if (threadIdx.x == 0)
    nh = nlo = nhi = nb = 0;
  u64 e0 = ~0ull, e1 = ~0ull;                            /* the two slice ends that can matter here (pair_heads_mark) */
  for (int t = 0; t < nthr; t++)
    { const u64 e = send[t];
      if (e > base)
        { e0 = e;
          if (t + 1 < nthr)
            e1 = send[t + 1];
          break;
        }
    }
  __syncthreads();

  /* 1. where the runs start, as bits -- a run's length is then the distance to the next bit (what lies behind the last seed,
        or behind the halo, counts as a start) -- and the heads of the tile as a list, so that what follows runs with full
        wavefronts (inside this loop a wavefront has a head in one lane of eight) */
  u64 mk[PH_ROUNDS];                                     /* the heads of this wavefront's 16 x 64 seeds (scalar registers) */
  u32 tot = 0;
#pragma unroll
  for (int r = 0; r <= PH_ROUNDS; r++)
    { const u32 off = (u32) r * 256u + threadIdx.x;
      const u64 i = base + off;
      bool st = true, f = false;
      if (i < nhits && off < DAMAR_SCAN_TILE + PW_HALO)
        { const u64 pr = sk[off + 1] >> pshift;
          const u64 ip = i + (u64) (minhit - 1);
          st = (i == 0) || (sk[off] >> pshift) != pr;
          if (r < PH_ROUNDS && st && ip < nhits)
            { const u64 kp = (minhit - 1 <= PW_HALO) ? sk[off + (u32) minhit] : keys[ip];
              if ((kp >> pshift) == pr)                  /* filter.c:2215: seed i + minhit - 1 is the same pair */
                { f = true;
                  if (i < e0)
                    { if (i + (u64) minhit >= e0) f = false; }
                  else if (i < e1)
                    { if (i + (u64) minhit >= e1) f = false; }
                  else                                   /* (three slice ends inside one tile: slices of a few hundred seeds) */
                    for (int t = 0; t < nthr; t++)
                      { u64 e = send[t];
                        if (i < e)
                          { if (i + (u64) minhit >= e) f = false;
                            break;
                          }
                      }
                }
            }
        }
      const u64 ms = __ballot(st);
      if (l == 0)
        bb[r * 4 + w] = ms;
      if (r < PH_ROUNDS)
        { mk[r] = __ballot(f);
          tot += (u32) __popcll(mk[r]);
        }
    }
  if (tot != 0)                                          /* one place in the list for all of them: one LDS atomic per wavefront */
    { u32 at = 0;
      if (l == 0)
        at = atomicAdd(&nh, tot);
      at = (u32) __shfl((int) at, 0);
#pragma unroll
      for (int r = 0; r < PH_ROUNDS; r++)
        { if ((mk[r] >> l) & 1)
            hl[at + (u32) __popcll(mk[r] & lanes_below(l))] = (u16) ((u32) r * 256u + threadIdx.x);
          at += (u32) __popcll(mk[r]);
        }
    }
  __syncthreads();

  /* 2. every head's run: those the screen has nothing to say about are kept or dropped here, the others go to one of three
        lists by the length of their run -- a wavefront of the screen costs what the longest of its 64 runs costs, and 80 %
        of the runs have 3 or 4 seeds (scripts/seed_runs.py) */
  const u32 nheads = nh;
  for (u32 h0 = 0; h0 < nheads; h0 += 256)               /* (a loop every lane leaves together: the ballots below) */
    { const u32 h = h0 + threadIdx.x;
      u32 off = 0;
      int n = 0, kind = 0;                               /* 1: short run, 2: small run */
      if (h < nheads)
        { off = hl[h];
          const u32 rb = (u32) ((sk[off + 1] >> pshift) >> abits);      /* a scheduler may hand this call a B-read range only */
          if (rb >= b_lo && rb < b_hi)
            { const u32 p = off + 1;                     /* the run's seeds: 1 + the seeds from p on that start no run */
              u64 ws = bb[p >> 6] >> (p & 63);
              if (p & 63)
                ws |= bb[(p >> 6) + 1] << (64 - (p & 63));
              n = ws ? __ffsll((long long) ws) : 65;     /* (65: more than 64) */
              if (n > SCREEN_MAX || minhit > SCREEN_MAX ||
                  (!UNS && (int) ((sk[off + (u32) n] >> dbits) & pmask) > SCREEN_PANEL))    /* (UNS: the screens look at the panel) */
                atomicOr(&kb[off >> 5], 1u << (off & 31));       /* not screened (pair_screen): kept */
              else if (n > PW_SMALL)
                bl[atomicAdd(&nb, 1u)] = off | ((u32) n << 16);
              else
                kind = n > PW_SHORT ? 2 : 1;
            }
        }
      const u64 m1 = __ballot(kind == 1), m2 = __ballot(kind == 2);
      u32 a1 = 0, a2 = 0;
      if (l == 0)
        { if (m1) a1 = atomicAdd(&nlo, (u32) __popcll(m1));
          if (m2) a2 = atomicAdd(&nhi, (u32) __popcll(m2));
        }
      a1 = (u32) __shfl((int) a1, 0);
      a2 = (u32) __shfl((int) a2, 0);
      if (kind == 1)
        cl[a1 + (u32) __popcll(m1 & lanes_below(l))] = (u16) (off | ((u32) n << 12));
      else if (kind == 2)
        cl[DAMAR_SCAN_TILE - 1 - (a2 + (u32) __popcll(m2 & lanes_below(l)))] = (u16) (off | ((u32) n << 12));
    }
  __syncthreads();

  /* 3. the screen.  Runs of up to PW_SMALL seeds: one lane each, the two lists one after the other */
  const u32 n_lo = nlo, n_hi = nhi;
  const u32 lo_rounds = (n_lo + 255u) & ~255u;           /* (the second list starts at a wavefront of its own) */
  for (u32 h0 = 0; h0 < lo_rounds + n_hi; h0 += 256)
    { const u32 h = h0 + threadIdx.x;
      const bool act = h0 < lo_rounds ? h < n_lo : h - lo_rounds < n_hi;
      const u32 e = act ? (u32) cl[h0 < lo_rounds ? h : DAMAR_SCAN_TILE - 1 - (h - lo_rounds)] : 0u;
      const u32 off = e & 0xfffu;
      const int n = act ? (int) (e >> 12) : 0;
      bool keep = false;
      if (dbits == 0)                                    /* (unpacked layout: the diagonals are in vals) */
        { const u64 i = base + off;
          int dlast = 0x7fffffff;
          for (int x = 0; x < n && !keep; x++)
            { const int dx = (int) vals[i + (u64) x] >> binshift;
              if (dx == dlast)
                continue;
              dlast = dx;
              int s0 = 0, s1 = 0, p0 = 0, p1 = 0;
              for (int y = 0; y < n; y++)
                { const int dy = (int) vals[i + (u64) y] >> binshift;
                  const int ap = (int) ((sk[off + 1 + (u32) y] >> dbits) & pmask);
                  if (dy == dx)
                    { s0 += (ap - p0 >= kmer) ? kmer : ap - p0;  p0 = ap; }
                  else if (dy == dx + 1)
                    { s1 += (ap - p1 >= kmer) ? kmer : ap - p1;  p1 = ap; }
                }
              keep = s0 + s1 >= hitmin;
            }
        }
      else if (__ballot(n > 0) != 0)                     /* all of the wavefront's runs with the code for its longest */
        { if      (__ballot(n > 7) != 0) keep = pw_screen_small<8, UNS>(sk + off + 1, n, pmask, dbits, binshift, kmer, hitmin);
          else if (__ballot(n > 6) != 0) keep = pw_screen_small<7, UNS>(sk + off + 1, n, pmask, dbits, binshift, kmer, hitmin);
          else if (__ballot(n > 5) != 0) keep = pw_screen_small<6, UNS>(sk + off + 1, n, pmask, dbits, binshift, kmer, hitmin);
          else if (__ballot(n > 4) != 0) keep = pw_screen_small<5, UNS>(sk + off + 1, n, pmask, dbits, binshift, kmer, hitmin);
          else if (__ballot(n > 3) != 0) keep = pw_screen_small<4, UNS>(sk + off + 1, n, pmask, dbits, binshift, kmer, hitmin);
          else                           keep = pw_screen_small<3, UNS>(sk + off + 1, n, pmask, dbits, binshift, kmer, hitmin);
        }
      if (keep)
        atomicOr(&kb[off >> 5], 1u << (off & 31));
    }
  /* the longer runs (up to 48 x 48 steps: in one lane a wavefront waited 58 us for one of them): a wavefront each, lane x
     takes seed x's bucket and all lanes walk the run together (the same LDS word for every lane) */
  const u32 nbig = nb;
  for (u32 q = threadIdx.x >> 6; q < nbig; q += 4)
    { const u32 off = bl[q] & 0xffffu;
      const int n = (int) (bl[q] >> 16);
      const u64 i = base + off;
      const int xl = l < n ? l : n - 1;                  /* (n <= SCREEN_MAX < 64) */
      const int dx = seed_diag(sk[off + 1 + (u32) xl], vals, i + (u64) xl, pmask, dbits) >> binshift;
      if (UNS)
        { /* seeds in no particular order: what seed x adds to its bucket first (the largest A position of its bucket
             that orders before it), then the sums over the two buckets, the other lanes' figures by ds_bpermute */
          const int apx = (int) ((sk[off + 1 + (u32) xl] >> dbits) & pmask);
          int prev = 0, top = 0;
          for (int y = 0; y < n; y++)
            { const u64 ky = sk[off + 1 + (u32) y];
              const int dy = seed_diag(ky, vals, i + (u64) y, pmask, dbits) >> binshift;
              const int ap = (int) ((ky >> dbits) & pmask);
              const bool before = ap < apx || (ap == apx && y < xl);
              prev = (dy == dx && before && ap > prev) ? ap : prev;
              top = ap > top ? ap : top;
            }
          const int addx = (l < n) ? min(apx - prev, kmer) : 0;
          int sum = 0;
          for (int y = 0; y < n; y++)
            { const int ay = __shfl(addx, y), dy = __shfl(dx, y);
              sum += ((u32) (dy - dx) <= 1u) ? ay : 0;
            }
          if ((top > SCREEN_PANEL || __ballot(l < n && sum >= hitmin) != 0) && l == 0)
            atomicOr(&kb[off >> 5], 1u << (off & 31));
          continue;
        }
      int s0 = 0, s1 = 0, p0 = 0, p1 = 0;
#pragma unroll 4
      for (int y = 0; y < n; y++)
        { const u64 ky = sk[off + 1 + (u32) y];
          const int dy = seed_diag(ky, vals, i + (u64) y, pmask, dbits) >> binshift;
          const int ap = (int) ((ky >> dbits) & pmask);
          if (dy == dx)
            { s0 += (ap - p0 >= kmer) ? kmer : ap - p0;  p0 = ap; }
          else if (dy == dx + 1)
            { s1 += (ap - p1 >= kmer) ? kmer : ap - p1;  p1 = ap; }
        }
      if (__ballot(s0 + s1 >= hitmin) != 0 && l == 0)
        atomicOr(&kb[off >> 5], 1u << (off & 31));
    }
  __syncthreads();
  if (threadIdx.x < 64)
    { const u64 m = (u64) kb[2 * threadIdx.x] | ((u64) kb[2 * threadIdx.x + 1] << 32);
      bits[(base >> 6) + threadIdx.x] = m;
      const int c = wave_incl_scan_i(__popcll(m));
      if (threadIdx.x == 63)
        tcount[blockIdx.x] = (u32) c;
    }
}

/* The seed sort over the read pair only (4 passes instead of 6: 28 of the 43 key bits) leaves the seeds of a pair in index
 * order; only the runs the report kernel will walk -- the kept heads' -- have to be in order of their A positions, ties in
 * the order they have (what the stable sort over all the bits leaves).  order_sort, one workgroup per work item once the
 * work list exists (the rule of shim.hip match_front turns this on where the kept runs are few and long): the run's length (the lanes probe 64 seeds at a time), the run into LDS, a bitonic sort of
 * (A position << 11 | place in the run) -- a strict order, so any sort is the stable one -- and the run back where it was.
 * order_probe, before the host reads the number of work items: is any kept head's run longer than a wavefront sorts in
 * LDS (OR_MAX)?  Then the caller sorts that comparison over all the bits after all. */
#define OR_MAX 2048
__device__ __forceinline__ u32 run_length(const u64 *__restrict__ keys, u64 nhits, u64 i, int pshift, u32 maxrun, int l)
{ const u64 pr = keys[i] >> pshift;
  u32 n = 0;
  for (;;)                                                    /* 64 probes at a time */
    { const u64 x = i + (u64) n + (u64) l;
      const u64 same = __ballot(x < nhits && (keys[x] >> pshift) == pr);
      if (same != ~0ull)
        return n + (u32) __ffsll((long long) ~same) - 1u;
      n += 64;
      if (n > maxrun)
        return n;
    }
}

__global__ __launch_bounds__(256)
void order_probe(const u64 *__restrict__ keys, u64 nhits, int ppos, int dbits, const u64 *__restrict__ bits, u64 *__restrict__ flag,
                 u32 maxrun)
{ SEED_PRIO(g_merge_prio);
  __shared__ u16 hl[DAMAR_SCAN_TILE];
  __shared__ u32 nh;
  const int  l = lane_id();
  const u64  base = (u64) blockIdx.x * DAMAR_SCAN_TILE;
  if (threadIdx.x == 0)
    nh = 0;
  __syncthreads();
  if (threadIdx.x < DAMAR_SCAN_TILE / 64)
    { u64 w = bits[(base >> 6) + threadIdx.x];
      while (w)
        { const int b = __ffsll((long long) w) - 1;
          w &= w - 1;
          hl[atomicAdd(&nh, 1u)] = (u16) (threadIdx.x * 64u + (u32) b);
        }
    }
  __syncthreads();
  const u32 n_h = nh;
  bool over = false;
  for (u32 h = threadIdx.x >> 6; h < n_h; h += 4)             /* a wavefront per kept head */
    over |= run_length(keys, nhits, base + hl[h], ppos + dbits, maxrun, l) > maxrun;
  if (over && l == 0)
    atomicOr((unsigned long long *) flag, 1ull);
}

__global__ __launch_bounds__(256)
void order_sort(u64 *__restrict__ keys, u64 nhits, int ppos, int dbits, const u32 *__restrict__ work, u32 nwork)
{ SEED_PRIO(g_merge_prio);
  __shared__ u64 rk[OR_MAX];
  __shared__ u32 ck[OR_MAX];
  const int  l = lane_id();                                  /* (every wavefront finds the run's length for itself) */
  const u64  pmask = (1ull << ppos) - 1;
  if (blockIdx.x >= nwork)
    return;
  const u64 i = work[blockIdx.x];
  const u32 n = run_length(keys, nhits, i, ppos + dbits, OR_MAX, l);
  if (n < 2 || n > OR_MAX)                                   /* (longer: order_probe has sent the comparison the other way) */
    return;
  u32 P = 2;
  while (P < n)
    P <<= 1;
  for (u32 j = threadIdx.x; j < P; j += 256)
    { if (j < n)
        { const u64 k = keys[i + j];
          rk[j] = k;
          ck[j] = ((u32) ((k >> dbits) & pmask) << 11) | j;
        }
      else
        ck[j] = 0xffffffffu;
    }
  __syncthreads();
  for (u32 k2 = 2; k2 <= P; k2 <<= 1)
    for (u32 jj = k2 >> 1; jj > 0; jj >>= 1)
      { for (u32 t = threadIdx.x; t < (P >> 1); t += 256)
          { const u32 ix = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), px = ix | jj;
            const bool up = (ix & k2) == 0;
            const u32 a = ck[ix], c = ck[px];
            if ((a > c) == up)
              { ck[ix] = c;  ck[px] = a; }
          }
        __syncthreads();
      }
  for (u32 j = threadIdx.x; j < n; j += 256)
    keys[i + j] = rk[ck[j] & 2047u];
}

void damar_launch_order_runs(u64 *keys, u64 nhits, int ppos, int dbits, const u32 *work, u32 nwork, hipStream_t st)
{ if (nwork == 0)
    return;
  hipLaunchKernelGGL(order_sort, dim3(nwork), dim3(256), 0, st, keys, nhits, ppos, dbits, work, nwork);
}

/* first half: bit words + *total_dev = the number of work items; the caller reads the total, makes room, and calls the
   second half.  bits: 64 u64 words per tile of DAMAR_SCAN_TILE seeds; scan_work: damar_scan_workspace_bytes(nhits) */
void damar_launch_pair_work(u64 *keys, const u32 *vals, u64 nhits, int ppos, int dbits, int abits, int minhit, int nshift,
                            u64 *send, u64 *bits, void *scan_work, u64 *total_dev, int binshift, int kmer, int hitmin,
                            u32 b_lo, u32 b_hi, int unsorted, hipStream_t st)
{ HIP_CHECK(hipMemsetAsync(total_dev, 0, 2 * sizeof(u64), st));         /* [1]: order_runs met a run it does not sort */
  if (nhits == 0)
    return;
  if (nshift > 6)
    nshift = 6;
  const u32 ntiles = (u32) ((nhits + DAMAR_SCAN_TILE - 1) / DAMAR_SCAN_TILE);
  u32 *tcount = (u32 *) scan_work;
  if (nshift >= 0)
    hipLaunchKernelGGL(slice_ends<u64>, dim3(1), dim3(64), 0, st, keys, nhits, abits + ppos + dbits, nshift, send);
  if (!unsorted)
    hipLaunchKernelGGL(pair_work_mark<false>, dim3(ntiles), dim3(256), 0, st, (const u64 *) keys, vals, nhits, ppos, dbits, abits, minhit,
                       nshift, (const u64 *) send, binshift, kmer, hitmin, b_lo, b_hi, bits, tcount);
  else
    { hipLaunchKernelGGL(pair_work_mark<true>, dim3(ntiles), dim3(256), 0, st, (const u64 *) keys, vals, nhits, ppos, dbits, abits, minhit,
                         nshift, (const u64 *) send, binshift, kmer, hitmin, b_lo, b_hi, bits, tcount);
      static int maxrun = -1;                      /* test hook DAMAR_TEST_RUN_MAX: runs longer than this count as too long (<= OR_MAX) */
      if (maxrun < 0)
        { const char *e = getenv("DAMAR_TEST_RUN_MAX");
          maxrun = (e && atoi(e) >= 2 && atoi(e) <= OR_MAX) ? atoi(e) : OR_MAX;
        }
      hipLaunchKernelGGL(order_probe, dim3(ntiles), dim3(256), 0, st, (const u64 *) keys, nhits, ppos, dbits, (const u64 *) bits, total_dev + 1,
                         (u32) maxrun);
    }
  damar_scan_tile_counts(tcount, ntiles, total_dev, st);
}

void damar_launch_pair_work_expand(const u64 *bits, const void *scan_work, u64 nhits, u32 *work, hipStream_t st)
{ if (nhits == 0)
    return;
  const u32 ntiles = (u32) ((nhits + DAMAR_SCAN_TILE - 1) / DAMAR_SCAN_TILE);
  hipLaunchKernelGGL(pair_heads_expand, dim3(ntiles), dim3(64), 0, st, bits, (const u32 *) scan_work, work);
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
  if (coarse >= 0xfffffffcu)    /* cost = the expected length of the alignment instead of a seed count: the wave kernel's
                                   launch ends with its longest serial chains, so those must start first */
    { const u64 pm = (1ull << pbits0) - 1;
      const u64 ext = ((keys[a - 1] >> dbits) & pm) - ((keys[i] >> dbits) & pm);      /* extent of the seeds on A */
      const u32 ra = (u32) (pr & ((1ull << abits) - 1)), rb = (u32) (pr >> abits);
      const int alen = (int) (aboff[ra + 1] - aboff[ra]) - 1, blen = (int) (bboff[rb + 1] - bboff[rb]) - 1;
      const int d = seed_diag(keys[i], vals, i, pm, dbits);             /* diagonal a - b of the first seed */
      const int geo = min(alen, blen + d) - max(0, d);                  /* overlap of the two reads on it */
      u64 len = (coarse == 0xffffffffu) ? ext : (coarse == 0xfffffffeu ? (u64) max(geo, 0) : max(ext, (u64) max(geo, 0)));
      if (coarse == 0xfffffffcu)  /* forward extent from the first seed: the two halves of a wavefront step in lockstep through
                                     the forward and then the reverse pass, so similar forward lengths side by side matter */
        { const int xa0 = (int) ((keys[i] >> dbits) & pm), xb0 = xa0 - d;
          const int fw = min(alen - xa0, blen - xb0);
          len = (u64) max(fw, 0);
        }
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

/* loads this file's code object now (a lazy load otherwise happens at the first launch, on the launching thread): called by
   the library's start-up thread, beside the caller's first uploads (shim.hip damar_hip_init) */
void damar_preload_merge(void)
{ hipFuncAttributes fa;
  (void) hipFuncGetAttributes(&fa, (const void *) merge_emit);
}

/* kmer_index.hip -- k-mer tuple generation and index helpers for gfx950.
 *
 * Restates reference dalign/filter.c:458-547 (tuple_thread, unmasked branch) and
 * :700-751, 890-939 (the -t frequency suppression) as data-parallel passes.
 *
 * Index layout in HBM (differs from the reference's 16-byte KmerPos records on
 * purpose): codes[i] (u32 for k <= 16, u64 for k <= 32; 2 bits per base, last base in the low bits) and
 * pos[i] (u32 offset of the k-mer's LAST base in the block's base array).  The
 * reference's (read, rpos) pair is recoverable from pos through the block's read
 * offsets, and because reads are laid out in order, sorting stably by code leaves
 * entries in the reference's (code, read, rpos) order.  The merge (seed_merge.hip) streams two such
 * indexes against each other; there is no look-up structure beside the two sorted arrays.
 */
#include "dev_common.h"
#include "kernels.h"

SEED_PRIO_VAR(g_index_prio, 0)
SEED_PRIO_SETTER(damar_index_set_prio, g_index_prio)


/* One thread per base position p: the read through the coarse table (two dependent look-ups
 * instead of a search over all reads), then the k-mer ENDING at p if the read has K bases up to
 * there.  Read r owns k-mer indices [boff[r] - r*k, boff[r+1] - (r+1)*k), in position order. */
/* PACK (k <= 16): code << 32 | pos as one u64 per k-mer, the form the index sort runs on (radix_sort.hip). */
template <typename CodeT, bool PACK>
__global__ __launch_bounds__(256)
void kmer_tuples(DevBlock blk, int kmer, u32 nkmers, CodeT *__restrict__ codes, u32 *__restrict__ pos)
{ SEED_PRIO(g_index_prio);
  const u32 p = blockIdx.x * 256u + threadIdx.x;
  if (p >= blk.total)
    return;
  const u32 r = read_of_pos(blk, p), b0 = blk.boff[r];
  if (p - b0 < (u32) (kmer - 1) || p + 1 == blk.boff[r + 1])       /* too close to the start / the terminator */
    return;
  const u32 i = p - (r + 1) * (u32) kmer + 1;
  if (i >= nkmers)
    return;
  CodeT c = 0;
  if (PACK)
    { /* k <= 16: the k-mer out of the 2-bit copy of the block (two adjacent words, base 16w in the low bits of word w)
         instead of k byte loads; the code wants the FIRST base in its high bits, so the window is reversed pair-wise */
      const u32 f = p - (u32) (kmer - 1), wq = f >> 4, o = (f & 15) << 1;
      u64 win = ((u64) blk.pk[wq + 1] << 32) | (u64) blk.pk[wq];
      win >>= o;
      u64 r = __brevll(win);                                           /* base j now sits at bits 62-2j, its two bits swapped */
      r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
      c = (CodeT) (r >> (64 - 2 * kmer));
    }
  else
    { const u8 *s = blk.bases + (p - (u32) (kmer - 1));
      for (int j = 0; j < kmer; j++)
        c = (CodeT) (c << 2) | (CodeT) s[j];
    }
  const u32 pw = pos_encode(blk, r, p - b0, p);                  /* the position word (kernels.h) */
  if (PACK)
    codes[i] = (CodeT) ((u64) c << 32) | (CodeT) pw;
  else
    { codes[i] = c;
      pos[i]   = pw;
    }
}

void damar_launch_kmer_tuples(const DevBlock *blk, int kmer, u32 nkmers, void *codes, int wide, u32 *pos, hipStream_t st)
{ if (nkmers == 0)
    return;
  if (wide)
    hipLaunchKernelGGL((kmer_tuples<u64, false>), dim3((blk->total + 255) / 256), dim3(256), 0, st, *blk, kmer, nkmers, (u64 *) codes, pos);
  else if (pos == NULL)         /* packed: codes holds one u64 per k-mer */
    hipLaunchKernelGGL((kmer_tuples<u64, true>), dim3((blk->total + 255) / 256), dim3(256), 0, st, *blk, kmer, nkmers, (u64 *) codes, pos);
  else
    hipLaunchKernelGGL((kmer_tuples<u32, false>), dim3((blk->total + 255) / 256), dim3(256), 0, st, *blk, kmer, nkmers, (u32 *) codes, pos);
}

/* The masked branch of tuple_thread (filter.c:474-526): between two mask intervals of a read
 * (and before the first / after the last) lies an unmasked stretch [p, q); the reference emits
 * exactly the k-mers with p <= first base and last base < q.  The intervals of a read are sorted
 * and disjoint, so the k-mer [s, e] is kept unless the first interval that ends after s begins
 * at or before e (an empty interval [b, b) still splits a stretch, as in the reference). */
__global__ __launch_bounds__(256)
void mask_flags(DevBlock blk, int kmer, const u32 *__restrict__ pos, u32 n, u32 *__restrict__ keep)
{ u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  u32 r, x;
  pos_decode(blk, pos[i], &r, &x);
  const int e = (int) x, s = e - (kmer - 1);
  u32 lo = blk.moff[r] >> 1, hi = blk.moff[r + 1] >> 1;
  const u32 end = hi;
  while (lo < hi)                               /* first interval with end > s */
    { const u32 mid = (lo + hi) >> 1;
      if (blk.mdat[2 * mid + 1] > s) hi = mid; else lo = mid + 1;
    }
  keep[i] = (lo < end && blk.mdat[2 * lo] <= e) ? 0u : 1u;
}

void damar_launch_mask_flags(const DevBlock *blk, int kmer, const u32 *pos, u32 n, u32 *keep, hipStream_t st)
{ if (n == 0)
    return;
  hipLaunchKernelGGL(mask_flags, dim3((n + 255) / 256), dim3(256), 0, st, *blk, kmer, pos, n, keep);
}

/* -b (filter.c:549-688): the window walk is a serial state machine along a read, so one thread
 * walks one read (a block holds >= 10^4 reads; this runs once per index build).  A k-mer that
 * ends at block offset P is left at codes[P] / pos[P] with keep[P] = 1 (keep is cleared by the
 * caller); the usual compaction then puts them in (read, rpos) order.  Masked reads are walked
 * stretch by stretch, including the reference's end-of-stretch case (:602: the growing loop stops
 * at the end of a stretch but the window is still offered, with rpos = the stretch end). */
template <typename CodeT>
__global__ __launch_bounds__(64)
void biased_tuples(DevBlock blk, int kmer, int lb0, int lb1, int lb2, int lb3,
                   CodeT *__restrict__ codes, u32 *__restrict__ pos, u32 *__restrict__ keep)
{ const u32 r = blockIdx.x * 64u + threadIdx.x;
  if (r >= blk.nreads)
    return;
  const int  LogNorm = 10000 * kmer, LogThresh = 10000 * (kmer - 2);
  const u64  kmask = (kmer == 32) ? ~0ull : ((1ull << (2 * kmer)) - 1);
  const u32  b0 = blk.boff[r];
  const int  rlen = (int) (blk.boff[r + 1] - b0) - 1;
  const u8  *s = blk.bases + b0;
  const bool masked = blk.moff != NULL;
  const u32  sb = masked ? blk.moff[r] : 0, sf = masked ? blk.moff[r + 1] : 0;
#define LB(x) ((x) == 0 ? lb0 : ((x) == 1 ? lb1 : ((x) == 2 ? lb2 : lb3)))
  for (u32 sa = sb; sa <= sf; sa += 2)
    { int p = (sa == sb) ? 0 : blk.mdat[sa - 1];
      const int q = (sa == sf) ? rlen : blk.mdat[sa];
      if (p + kmer > q)
        continue;
      u64 c = 0;
      int a = 0, k = 1;
      bool stop = false;
      while (p < q)
        { int x = s[p];
          a += LB(x);
          c = (c << 2) | (u64) x;
          while (a < LogNorm && k < kmer)
            { if (++p >= q)
                { stop = !masked;
                  break;
                }
              k += 1;
              x = s[p];
              a += LB(x);
              c = (c << 2) | (u64) x;
            }
          if (stop)
            break;
          for (;;)
            { const int u = a - LB((int) s[p - k + 1]);
              if (u < LogNorm) break;
              a = u;
              k -= 1;
            }
          if (a > LogThresh)
            { const u32 P = b0 + (u32) p;
              codes[P] = (CodeT) ((c << (2 * kmer - 2 * k)) & kmask);
              pos[P]   = pos_encode(blk, r, (u32) p, P);
              keep[P]  = 1u;
            }
          p += 1;
          a -= LB((int) s[p - k]);
        }
    }
#undef LB
}

void damar_launch_biased_tuples(const DevBlock *blk, int kmer, const int *logbase, void *codes, int wide, u32 *pos,
                                u32 *keep, hipStream_t st)
{ if (blk->nreads == 0)
    return;
  if (wide)
    hipLaunchKernelGGL(biased_tuples<u64>, dim3((blk->nreads + 63) / 64), dim3(64), 0, st, *blk, kmer,
                       logbase[0], logbase[1], logbase[2], logbase[3], (u64 *) codes, pos, keep);
  else
    hipLaunchKernelGGL(biased_tuples<u32>, dim3((blk->nreads + 63) / 64), dim3(64), 0, st, *blk, kmer,
                       logbase[0], logbase[1], logbase[2], logbase[3], (u32 *) codes, pos, keep);
}

/* -t (filter.c:700-751, 890-939): keep[i] = 1 iff the run of equal codes around sorted entry i is shorter than
 * `suppress`.  Runs are short, so each entry gallops to the ends of its own run (doubling steps, then a binary search
 * over the last stride) instead of consulting an index-wide table. */
template <typename CodeT>
__global__ __launch_bounds__(256)
void suppress_flags(const CodeT *__restrict__ codes, u32 n, u32 suppress, u32 *__restrict__ keep)
{ const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  const CodeT c = codes[i];
  u32 lo = i, step = 1;                               /* codes[lo] == c throughout */
  while (lo >= step && codes[lo - step] == c)
    { lo -= step;
      step <<= 1;
    }
  { u32 a = (lo >= step) ? lo - step + 1 : 0, b = lo;  /* first index of the run lies in [a, b] */
    while (a < b)
      { const u32 mid = (a + b) >> 1;
        if (codes[mid] == c) b = mid; else a = mid + 1;
      }
    lo = a;
  }
  u32 hi = i;
  step = 1;
  while (hi + step < n && codes[hi + step] == c)
    { hi += step;
      step <<= 1;
    }
  { u32 a = hi, b = (hi + step < n) ? hi + step - 1 : n - 1;      /* last index of the run lies in [a, b] */
    while (a < b)
      { const u32 mid = (a + b + 1) >> 1;
        if (codes[mid] == c) a = mid; else b = mid - 1;
      }
    hi = a;
  }
  keep[i] = (hi - lo + 1 < suppress) ? 1u : 0u;
}

void damar_launch_suppress_flags(const void *codes, int wide, u32 n, int suppress, u32 *keep, hipStream_t st)
{ if (n == 0)
    return;
  if (wide)
    hipLaunchKernelGGL(suppress_flags<u64>, dim3((n + 255) / 256), dim3(256), 0, st, (const u64 *) codes, n, (u32) suppress, keep);
  else
    hipLaunchKernelGGL(suppress_flags<u32>, dim3((n + 255) / 256), dim3(256), 0, st, (const u32 *) codes, n, (u32) suppress, keep);
}

template <typename CodeT>
__global__ __launch_bounds__(256)
void compact_pairs(const CodeT *__restrict__ k, const u32 *__restrict__ v, const u32 *__restrict__ keep,
                   const u32 *__restrict__ off, u32 n, CodeT *__restrict__ ko, u32 *__restrict__ vo)
{ u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i < n && keep[i])
    { ko[off[i]] = k[i];
      vo[off[i]] = v[i];
    }
}

/* the same, leaving code << 32 | pos (u32 codes only) */
__global__ __launch_bounds__(256)
void compact_pack(const u32 *__restrict__ k, const u32 *__restrict__ v, const u32 *__restrict__ keep,
                  const u32 *__restrict__ off, u32 n, u64 *__restrict__ ko)
{ u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i < n && keep[i])
    ko[off[i]] = ((u64) k[i] << 32) | (u64) v[i];
}

void damar_launch_compact_pairs(const void *k, int wide, const u32 *v, const u32 *keep, const u32 *off, u32 n,
                                void *ko, u32 *vo, hipStream_t st)
{ if (n == 0)
    return;
  if (wide)
    hipLaunchKernelGGL(compact_pairs<u64>, dim3((n + 255) / 256), dim3(256), 0, st, (const u64 *) k, v, keep, off, n, (u64 *) ko, vo);
  else if (vo == NULL)
    hipLaunchKernelGGL(compact_pack, dim3((n + 255) / 256), dim3(256), 0, st, (const u32 *) k, v, keep, off, n, (u64 *) ko);
  else
    hipLaunchKernelGGL(compact_pairs<u32>, dim3((n + 255) / 256), dim3(256), 0, st, (const u32 *) k, v, keep, off, n, (u32 *) ko, vo);
}

/* datander links (scrub/tandem.c:556-589).  Sorted entry i of a run of equal codes gets the
 * distance to entry i-1 if both lie in the same read, else 0; the reference never overwrites
 * the code of sorted entry 0, so that one keeps its k-mer code.  The result is stored at the
 * k-mer's index in position order, which is what the reference's re-sort on (read, rpos)
 * produces (every position owns exactly one k-mer). */
template <typename CodeT>
__global__ __launch_bounds__(256)
void tandem_links(DevBlock blk, int kmer, const CodeT *__restrict__ codes, const u32 *__restrict__ pos, u32 n,
                  int *__restrict__ dist)
{ u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  u32 r, x;
  pos_decode(blk, pos[i], &r, &x);
  const u32 p = blk.boff[r] + x;
  int d = 0;
  if (i == 0)
    d = (int) codes[0];             /* tandem.c:571-573 leaves the first entry's (truncated) code in place */
  else if (codes[i] == codes[i - 1])
    { u32 rq, xq;
      pos_decode(blk, pos[i - 1], &rq, &xq);
      if (rq == r)                          /* same read: entries of a run are in position order */
        d = (int) (x - xq);
    }
  dist[p - (r + 1) * (u32) kmer + 1] = d;
}

void damar_launch_tandem_links(const DevBlock *blk, int kmer, const void *codes, int wide, const u32 *pos, u32 n,
                               int *dist, hipStream_t st)
{ if (n == 0)
    return;
  if (wide)       /* k > 16: 64-bit codes, as scrub/tandem.c:132-149 */
    hipLaunchKernelGGL(tandem_links<u64>, dim3((n + 255) / 256), dim3(256), 0, st, *blk, kmer, (const u64 *) codes, pos, n, dist);
  else
    hipLaunchKernelGGL(tandem_links<u32>, dim3((n + 255) / 256), dim3(256), 0, st, *blk, kmer, (const u32 *) codes, pos, n, dist);
}

/* 2-bit packed copy of a block's bases for the alignment wave (16 bases per dword), followed by the same bases in
   REVERSE order (word nwords + w holds bases total-1-16w' ... downwards, w' = w - PK_PAD): a reverse pass of the wave
   then slides along ascending addresses like a forward one and needs no bit reversal of its windows (report_packed.h) */
__global__ __launch_bounds__(256)
void pack_bases(const u8 *__restrict__ bases, long long nwords, long long total, u32 *__restrict__ pk)
{ long long w = (long long) blockIdx.x * 256 + threadIdx.x - PK_PAD;
  if (w >= nwords - PK_PAD)
    return;
  const u8 *s = bases + 16 * w;                     /* >= bases - 64: inside the padding */
  u32 v = 0, r = 0;
  for (int j = 0; j < 16; j++)
    { v |= (u32) (s[j] & 3) << (2 * j);
      const long long q = total - 1 - (16 * w + j);                  /* the base the reversed copy holds at 16 w + j */
      const u32 c = (q >= -64 && q < total + 64) ? (u32) (bases[q] & 3) : 0u;
      r |= c << (2 * j);
    }
  pk[w] = v;
  pk[nwords + w] = r;
}

/* A block as it lies in the .bps file (raw: 2 bits per base, four per byte, first base in the top bits, every read
   padded to a byte; foff[r] = where read r starts in raw) into the layout Read_All_Sequences gives it (db/DB.c:1562-1600:
   one byte per base, read r at boff[r], a 4 behind every read) -- or, with comp, into the layout of the block's reverse
   complement (daligner.c:511-570: every read reversed in place, bases 3 - x).  One thread per four output bytes; bases
   must have been set to 4 beforehand (the terminators are not written here).  The host used to do both (37 + 37 ms per
   135 Mbp block and strand on a reader thread, then 2 x 135 MB over PCIe: round 5 sends 34 MB once). */
__global__ __launch_bounds__(256)
void unpack_bps(const u8 *__restrict__ raw, const u32 *__restrict__ foff, DevBlock blk, int comp, u8 *__restrict__ bases)
{ const u32 p0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
  if (p0 >= blk.total)
    return;
  u32 r = read_of_pos(blk, p0);
  u32 b0 = blk.boff[r], len = blk.boff[r + 1] - b0 - 1, fo = foff[r];
  u32 word = 0;
#pragma unroll
  for (int j = 0; j < 4; j++)
    { const u32 p = p0 + (u32) j;
      u32 v = 4;
      if (p < blk.total)
        { if (p >= b0 + len + 1)                         /* into the next read (reads are at least one base long) */
            { r += 1;
              b0 = blk.boff[r];  len = blk.boff[r + 1] - b0 - 1;  fo = foff[r];
            }
          const u32 x = p - b0;
          if (x < len)
            { const u32 y = comp ? len - 1 - x : x;
              v = ((u32) raw[fo + (y >> 2)] >> (6 - 2 * (y & 3))) & 3u;
              if (comp)
                v = 3 - v;
            }
        }
      word |= v << (8 * j);
    }
  *(u32 *) (bases + p0) = word;
}

void damar_launch_unpack_bps(const u8 *raw, const u32 *foff, const DevBlock *blk, int comp, u8 *bases, hipStream_t st)
{ if (blk->total == 0)
    return;
  hipLaunchKernelGGL(unpack_bps, dim3((blk->total / 4 + 256) / 256), dim3(256), 0, st, raw, foff, *blk, comp, bases);
}

/* pk must hold 2 * damar_pack_words(total) words */
long long damar_pack_words(u32 total) { return (long long) (total >> 4) + 1 + 2 * PK_PAD; }      /* words -PK_PAD .. total/16 + PK_PAD */

void damar_launch_pack_bases(const u8 *bases, u32 total, u32 *pk, hipStream_t st)
{ const long long nwords = damar_pack_words(total);
  hipLaunchKernelGGL(pack_bases, dim3((u32) ((nwords + 255) / 256)), dim3(256), 0, st, bases, nwords, (long long) total, pk);
}

/* loads this file's code object now (a lazy load otherwise happens at the first launch, on the launching thread): called by
   the library's start-up thread, beside the caller's first uploads (shim.hip damar_hip_init) */
void damar_preload_index(void)
{ hipFuncAttributes fa;
  (void) hipFuncGetAttributes(&fa, (const void *) pack_bases);
}

/* radix_sort.hip -- stable LSD radix sort for gfx950 in ONE sweep per digit.
 *
 * Replaces the reference's threaded 16-byte-record radix sort (dalign/filter.c:230-435 lex_thread / lex_sort),
 * which it uses for the k-mer index (filter.c:854) and for the seed pairs (filter.c:2776).  Both orders the
 * reference produces are total orders on (key, original position), so any stable sort on the same key bits
 * yields the same sequence.
 *
 * Shape of a sort of P 8-bit digits (P + 1 launches, every key read P + 1 times and written P times):
 *
 *   onesweep_hist   one read of the keys: the 256-bin histograms of ALL P digits at once (the reference gets the
 *                   next digit's histogram out of the scatter of the current one, filter.c:245-300; here every
 *                   digit's comes out of one pass up front), 32 copies of every bin in LDS, one per bank, so that
 *                   the LDS atomics of a wavefront never conflict; also clears the look-back words
 *   onesweep_pass   x P: a workgroup takes the next tile from a ticket counter, ranks its keys (ballot match-any
 *                   per wavefront, per-wavefront digit counts in LDS), publishes the tile's 256 digit counts,
 *                   finds the counts of all earlier tiles by DECOUPLED LOOK-BACK (one thread per digit walks the
 *                   earlier tiles' words back to the first inclusive prefix), stages the tile through LDS in
 *                   output order and writes each digit's items as one contiguous run.  No per-pass histogram
 *                   kernel, no scan kernel, no second read of the keys.
 *
 * Look-back words ("granules"): one word per (tile, digit) = status (2 bits: 0 empty, 1 tile count, 2 inclusive
 * prefix) | value, written by ONE relaxed agent-scope store and polled by relaxed agent-scope loads -- the data is
 * the flag, so no fence is needed (cdna_hip_programming.md Guideline 16, form R2).  Tiles are taken in ticket order,
 * so every earlier tile belongs to a workgroup that is already running: the look-back cannot wait for a tile that
 * has not started.  Every spin is bounded; a timeout raises the error word that the host checks.
 * 32-bit granules serve sorts of fewer than 2^30 items, 64-bit ones the rest.
 */
#include <atomic>
#include <type_traits>
#include "dev_common.h"
#include "kernels.h"

SEED_PRIO_VAR(g_sort_prio, 1)
SEED_PRIO_SETTER(damar_sort_set_prio, g_sort_prio)

#ifndef OS_THREADS
#define OS_THREADS 1024                      /* threads per workgroup: 1024 (x 8 keys: 16 wavefronts of <= 64 registers, two workgroups
                                                per CU) or 512 (x 16 keys) -- both tiles of 8192 keys: long runs per digit -- or
                                                256 (x 16 keys: tiles of 4096) */
#endif
#ifndef OS_MINW
#define OS_MINW    4                         /* wavefronts per SIMD the pass kernel is compiled for */
#endif
#ifndef OS_RANK_LDS
#define OS_RANK_LDS 1                        /* 1: the lanes with my digit through a 64-bit LDS word per (wavefront, digit); 0: eight ballots */
#endif
#define OS_MINTILE (256 * 8)                 /* the smallest tile shape (workspace bound) */
#define OS_MAXPASS 8

#define OH_ITEMS   16

#define LB_COUNT   1u
#define LB_PREFIX  2u
#define LB_SPINS   (1u << 24)
#ifndef LB_AHEAD
#define LB_AHEAD   8                         /* look-back words requested per round trip */
#endif

/* workspace: [0] error word | [256..) P x 256 digit totals | [8448..) P tickets | [16384..) two look-back regions */
#define WS_HIST    256
#define WS_CTR     (256 + OS_MAXPASS * 256 * 4)
#define WS_LB      16384

template <typename GT> __device__ __forceinline__ GT lb_load(const GT *p)
{ return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename GT> __device__ __forceinline__ void lb_store(GT *p, GT v)
{ __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* exclusive scan over the first 256 threads of the workgroup (all threads call it) */
__device__ __forceinline__ u32 os_scan256(u32 v, u32 *lds4)
{ const int l = lane_id(), w = threadIdx.x >> 6;
  const u32 inc = (u32) wave_incl_scan_i((int) v);
  if (l == 63 && w < 4) lds4[w] = inc;
  __syncthreads();
  u32 base = 0;
  for (int i = 0; i < 4; i++)
    if (i < w) base += lds4[i];
  __syncthreads();
  return base + inc - v;
}

/***** all digits' histograms in one read ********************************************************************/

/* LDS bins without bank conflicts: every bin exists in 32 copies, one per bank -- a lane adds into copy (lane & 31), so the
   32 lanes an LDS instruction serves per cycle hit 32 different banks whatever their digits are (lanes l and l + 32 are
   served in different cycles) -- and two 16-bit bins share a word: word (d >> 1) * 32 + copy, half (d & 1).  16 KB per digit
   place, one workgroup of 1024 threads per CU.  A 16-bit bin receives at most 32 threads x OH_ITEMS keys per tile: the grid
   is sized so that no bin of a workgroup can pass OH_BINMAX. */
#define OH_THREADS 1024
#define OH_BINMAX  61440                    /* what a 16-bit bin may receive (< 2^16) */

template <typename KeyT>
__global__ __launch_bounds__(OH_THREADS)
void onesweep_hist(const KeyT *__restrict__ keys, u64 n, int lobit, int npass, u64 dsh /* a byte per pass: where its digit starts */,
                   u64 dwd /* a byte per pass: its mask */, u32 *__restrict__ ghist, uint4 *__restrict__ clr, u64 clr16)
{ extern __shared__ u32 sh[];                                  /* [npass][128][32] */
  constexpr u32 nthr = OH_THREADS;
  const int nw = npass * 128 * 32;
  for (int j = threadIdx.x; j < nw; j += nthr)
    sh[j] = 0;
  __syncthreads();
  const u32 c = threadIdx.x & 31u;
  const u64 ntile = (n + (u64) nthr * OH_ITEMS - 1) / ((u64) nthr * OH_ITEMS);
  for (u64 t = blockIdx.x; t < ntile; t += gridDim.x)
    { const u64 base = t * ((u64) nthr * OH_ITEMS);
      KeyT k[OH_ITEMS];
#pragma unroll
      for (int r = 0; r < OH_ITEMS; r++)
        { const u64 i = base + (u64) r * nthr + threadIdx.x;
          k[r] = (i < n) ? keys[i] : (KeyT) 0;
        }
#pragma unroll
      for (int r = 0; r < OH_ITEMS; r++)
        { const u64 i = base + (u64) r * nthr + threadIdx.x;
          if (i < n)
            { const KeyT x = k[r] >> lobit;
              for (int p = 0; p < npass; p++)
                { const u32 d = (u32) (x >> (int) ((dsh >> (8 * p)) & 0xffu)) & (u32) ((dwd >> (8 * p)) & 0xffu);
                  atomicAdd(&sh[((u32) p << 12) + ((d >> 1) << 5) + c], 1u << ((d & 1u) << 4));
                }
            }
        }
    }
  __syncthreads();
  /* bin (p, d) = sum over its 32 copies; the lanes of a wavefront start at different copies: no conflicts here either */
  for (int j = threadIdx.x; j < npass * 256; j += nthr)
    { const u32 pl = (u32) j >> 8, d = (u32) j & 0xffu, sft = (d & 1u) << 4;
      u32 sum = 0;
#pragma unroll 8
      for (u32 q = 0; q < 32; q++)
        sum += (sh[(pl << 12) + ((d >> 1) << 5) + ((q + threadIdx.x) & 31u)] >> sft) & 0xffffu;
      if (sum)
        atomicAdd(&ghist[j], sum);
    }
  /* the first look-back region starts empty */
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (u64 i = (u64) blockIdx.x * nthr + threadIdx.x; i < clr16; i += (u64) gridDim.x * nthr)
    clr[i] = z;
}

/* The histogram beside a resident report launch: 256 threads (one free wave slot per SIMD) and 8 KB per digit place --
   the bank-private layout above wants 16 KB per place in ONE piece, which the LDS of a CU shared with report workgroups
   often cannot give (measured: launches waiting up to a whole report launch, profiles/r03_sweeps.txt). */
#define OH_COPIES  8
template <typename KeyT>
__global__ __launch_bounds__(256)
void onesweep_hist8(const KeyT *__restrict__ keys, u64 n, int lobit, int npass, u64 dsh /* a byte per pass: where its digit starts */,
                   u64 dwd /* a byte per pass: its mask */, u32 *__restrict__ ghist, uint4 *__restrict__ clr, u64 clr16)
{ SEED_PRIO(g_sort_prio);
  extern __shared__ u32 sh[];                                  /* [npass][256][OH_COPIES]: the bins of a digit place, OH_COPIES copies chosen by lane & 7 */
  const int nb = npass * 256 * OH_COPIES;
  for (int j = threadIdx.x; j < nb; j += 256)
    sh[j] = 0;
  __syncthreads();
  const u32 c = threadIdx.x & (OH_COPIES - 1);
  const u64 ntile = (n + 256 * OH_ITEMS - 1) / (256 * OH_ITEMS);
  for (u64 t = blockIdx.x; t < ntile; t += gridDim.x)
    { const u64 base = t * (256 * OH_ITEMS);
      KeyT k[OH_ITEMS];
#pragma unroll
      for (int r = 0; r < OH_ITEMS; r++)
        { const u64 i = base + (u64) r * 256 + threadIdx.x;
          k[r] = (i < n) ? keys[i] : (KeyT) 0;
        }
#pragma unroll
      for (int r = 0; r < OH_ITEMS; r++)
        { const u64 i = base + (u64) r * 256 + threadIdx.x;
          if (i < n)
            { const KeyT x = k[r] >> lobit;
              for (int p = 0; p < npass; p++)
                { const u32 d = (u32) (x >> (int) ((dsh >> (8 * p)) & 0xffu)) & (u32) ((dwd >> (8 * p)) & 0xffu);
                  atomicAdd(&sh[(((u32) p << 8) + d) * OH_COPIES + c], 1u);
                }
            }
        }
    }
  __syncthreads();
  for (int j = threadIdx.x; j < npass * 256; j += 256)
    { u32 s = 0;
#pragma unroll
      for (int q = 0; q < OH_COPIES; q++)
        s += sh[j * OH_COPIES + q];
      if (s)
        atomicAdd(&ghist[j], s);
    }
  /* the first look-back region starts empty */
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (u64 i = (u64) blockIdx.x * 256 + threadIdx.x; i < clr16; i += (u64) gridDim.x * 256)
    clr[i] = z;
}

/***** one digit: rank, look back, scatter ******************************************************************/

/* Wavefront w owns items [w * OS_WSPAN, (w + 1) * OS_WSPAN) of its tile in rounds of 64 consecutive items, so
 * (wavefront, round, lane) order is input order and the ranks below make the pass stable.
 * HV: a u32 payload travels with the key.  SPLIT: the last pass of the packed k-mer index -- the key's high word
 * goes to ohi, its low word to vout (no key array is written). */
template <typename KeyT, typename GT, bool HV, bool SPLIT, int TH, int IT>
__global__ __launch_bounds__(TH, (IT <= 8 ? 8 : OS_MINW))      /* the small shape asks for <= 64 VGPRs: two wavefronts per SIMD beside a report launch */
void onesweep_pass(const KeyT *__restrict__ kin, const u32 *__restrict__ vin, KeyT *__restrict__ kout,
                   u32 *__restrict__ vout, u32 *__restrict__ ohi, u64 n, int shift, u32 mask,
                   const u32 *__restrict__ ghist, GT *__restrict__ lb, GT *__restrict__ lbclear,
                   u32 *__restrict__ ctr, u32 *__restrict__ err)
{ SEED_PRIO(g_sort_prio);
  constexpr int OS_TILE = TH * IT, OS_WAVES = TH / 64, OS_WSPAN = 64 * IT;
  /* keys and payload are staged through the SAME buffer one after the other */
  __shared__ __attribute__((aligned(16))) KeyT skey[OS_TILE];
  u32 *const sval = (u32 *) skey;
  typedef typename std::conditional<(OS_WAVES > 8), u16, u32>::type CntT;      /* (a wavefront's count of a digit is at most 64 * IT) */
  __shared__ CntT cnt[OS_WAVES][256];
  __shared__ u32  dstart[256];
  __shared__ u32  gadj[256];
  __shared__ u32  lds4[4];
  __shared__ u32  s_tile;
  constexpr int SH = (int) sizeof(GT) * 8 - 2;
  constexpr GT  VM = (((GT) 1) << SH) - 1;

  const int l = lane_id(), w = threadIdx.x >> 6;
  if (threadIdx.x == 0)
    s_tile = atomicAdd(ctr, 1u);
  for (int i = 0; i < OS_WAVES; i++)
    if (threadIdx.x < 256)
      cnt[i][threadIdx.x] = 0;
#if OS_RANK_LDS
  static_assert((size_t) OS_TILE * sizeof(KeyT) >= (size_t) OS_WAVES * 256 * sizeof(u64), "the rank words do not fit the stage");
  for (int i = threadIdx.x; i < OS_WAVES * 256; i += TH)
    ((u64 *) skey)[i] = 0;
#endif
  __syncthreads();
  const u32 tile  = (u32) __builtin_amdgcn_readfirstlane((int) s_tile);
  const u64 tbase = (u64) tile * OS_TILE;
  /* tile-local 32-bit indexes against lane-uniform tile pointers: one offset register per thread, 32-bit bound checks */
  const u32 have = (n - tbase < (u64) OS_TILE) ? (u32) (n - tbase) : (u32) OS_TILE;
  const u32 t0 = (u32) w * OS_WSPAN + (u32) l;
  const KeyT *const tin = kin + tbase;

  KeyT key[IT];
  u32  rnk[IT];
#pragma unroll
  for (int r = 0; r < IT; r++)
    { const u32 ti = t0 + (u32) r * 64;
      key[r] = tin[ti < have ? ti : have - 1];               /* (no branch around a load; what a lane beyond the end reads is never used) */
    }
#if OS_RANK_LDS
  /* Ranking through LDS (round 5): the lanes of a wavefront that hold the same digit find each other in a 64-bit word
     per (wavefront, digit) -- every lane ORs its lane bit into the word of its digit, then reads the word back: the lanes
     with my digit, in ONE round trip instead of eight ballots (4 vector instructions per digit bit, 32 of the pass's 89
     per 64 keys; the pass was bound by vector issue at 61 % of the pipes).  LDS executes the
     operations of one wavefront in order, so the read sees every lane's OR and the first lane's clear comes behind every
     lane's read.  The words live where the tile is staged later (skey): nothing is staged before every wavefront is
     through with its ranks. */
  u64 *const wmask = (u64 *) skey + (u32) w * 256u;          /* (cleared before the first barrier) */
  const u64 lanebit = 1ull << l;
#pragma unroll
  for (int r = 0; r < IT; r++)
    { const bool ok = t0 + (u32) r * 64 < have;
      const u32  d  = (u32) (key[r] >> shift) & mask;
      if (ok)
        __hip_atomic_fetch_or(&wmask[d], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const u64 peers  = __hip_atomic_load(&wmask[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const u32 before = cnt[w][d];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const u32 mine   = __builtin_amdgcn_mbcnt_hi((u32) (peers >> 32), __builtin_amdgcn_mbcnt_lo((u32) peers, 0u));
      rnk[r] = before + mine;
      if (ok && mine == 0)
        { cnt[w][d] = (CntT) (before + (u32) __popcll(peers));
          __hip_atomic_store(&wmask[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
#else
#pragma unroll
  for (int r = 0; r < IT; r++)
    { const bool ok = t0 + (u32) r * 64 < have;
      const u32  d  = (u32) (key[r] >> shift) & mask;
      /* the lanes with my digit: for every bit, the ballot of the bit or its complement -- (~m) ^ (-bit) is m where the
         bit is set and ~m where it is clear, so a bit costs a sign-extending field extract, a compare and two 3-input
         logic operations (the plain `bit ? m : ~m` compiled to nine) */
      const u64 okm = __ballot(ok);
      u32 plo = (u32) okm, phi = (u32) (okm >> 32);
#pragma unroll
      for (int b = 0; b < 8; b++)
        { const int t  = ((int) (d << (31 - b))) >> 31;
          const u64 nm = ~__ballot(t < 0);
          plo &= (u32) nm ^ (u32) t;
          phi &= (u32) (nm >> 32) ^ (u32) t;
        }
      const u32 before = cnt[w][d];
      const u32 mine   = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
      rnk[r] = before + mine;
      if (ok && mine == 0)
        cnt[w][d] = (CntT) (before + (u32) __popc(plo) + (u32) __popc(phi));
    }
#endif
  __syncthreads();

  /* thread d: digit d's count in this tile; publish it, then the places of the digits inside the tile */
  u32 tot = 0;
  if (threadIdx.x < 256)
    { u32 run = 0;
#pragma unroll
      for (int i = 0; i < OS_WAVES; i++)
        { const u32 c = cnt[i][threadIdx.x];
          cnt[i][threadIdx.x] = (CntT) run;
          run += c;
        }
      tot = run;
      lb_store<GT>(&lb[(u64) tile * 256 + threadIdx.x],
                   ((GT) (tile == 0 ? LB_PREFIX : LB_COUNT) << SH) | (GT) tot);
      lbclear[(u64) tile * 256 + threadIdx.x] = 0;            /* the other region, for the next pass */
    }
  const u32 dbase = os_scan256(threadIdx.x < 256 ? ghist[threadIdx.x] : 0u, lds4);   /* where digit d's items start */
  const u32 ex    = os_scan256(tot, lds4);
  if (threadIdx.x < 256)
    dstart[threadIdx.x] = ex;
  __syncthreads();

#pragma unroll
  for (int r = 0; r < IT; r++)
    { if (t0 + (u32) r * 64 < have)
        { const u32 d  = (u32) (key[r] >> shift) & mask;
          const u32 lp = dstart[d] + cnt[w][d] + rnk[r];
          rnk[r] = lp;                                         /* position inside the tile's output */
          skey[lp] = key[r];
        }
    }

  /* look back: items of digit d in all earlier tiles.  A word is polled by a load that goes to memory (agent scope: the
     L2s of the XCDs are not coherent), and in steady state the nearest inclusive prefix is about 20 tiles back
     (tools/sortbench -DOS_STATS) -- walked one tile per round trip that was most of a tile's time.  LB_AHEAD words are
     requested per round trip and consumed in order: a prefix ends the walk, an empty word ends the round. */
  if (threadIdx.x < 256)
    { GT before = 0;
      if (tile > 0 && threadIdx.x <= mask)                   /* (a digit the pass does not have: nothing before it anywhere) */
        { u32 t = tile - 1, spins = 0;
#ifdef OS_STATS
          u32 nstep = 0;
#endif
          bool found = false;
          while (!found)
            { GT g[LB_AHEAD];
#pragma unroll
              for (int k = 0; k < LB_AHEAD; k++)                 /* (below tile 0 the word of tile 0 again: its prefix ends the walk before that) */
                g[k] = lb_load<GT>(&lb[(u64) (t >= (u32) k ? t - (u32) k : 0u) * 256 + threadIdx.x]);
              int used = LB_AHEAD;
#pragma unroll
              for (int k = 0; k < LB_AHEAD; k++)
                if (used == LB_AHEAD)
                  { const u32 st = (u32) (g[k] >> SH);
                    if (st == 0)
                      used = k;
                    else
                      { before += g[k] & VM;
#ifdef OS_STATS
                        nstep += 1;
#endif
                        if (st == LB_PREFIX)
                          { found = true;  used = -1; }
                      }
                  }
              if (found)
                break;
              t -= (u32) used;                                   /* tile 0 publishes a prefix: t never passes it */
              if (used < LB_AHEAD)                               /* an earlier tile has not published its count yet */
                { if (++spins > LB_SPINS)
                    { atomicOr(err, 1u);
                      break;
                    }
                  __builtin_amdgcn_s_sleep(1);
                }
            }
#ifdef OS_STATS                                                /* tools/sortbench: look-back steps, polls of an empty word, tiles */
          if (threadIdx.x == 0)
            { atomicAdd(err + 1, nstep);  atomicAdd(err + 2, spins);  atomicAdd(err + 3, 1u); }
#endif
          lb_store<GT>(&lb[(u64) tile * 256 + threadIdx.x], ((GT) LB_PREFIX << SH) | ((before + tot) & VM));
        }
      gadj[threadIdx.x] = dbase + (u32) before - ex;
    }
  __syncthreads();

  u32 gdst[HV ? IT : 1];                                 /* where this thread's output positions go (n < 2^32) */
#pragma unroll
  for (int q = 0; q < IT; q++)
    { const u32 i = threadIdx.x + (u32) q * TH;
      if (i < have)
        { const KeyT k = skey[i];
          const u32  d = (u32) (k >> shift) & mask;
          const u32  g = gadj[d] + i;
          if (HV) gdst[q] = g;
          if (SPLIT)
            { ohi[g]  = (u32) ((u64) k >> 32);
              vout[g] = (u32) k;
            }
          else
            kout[g] = k;
        }
    }
  if (HV)
    { const u32 *const tvin = vin + tbase;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < IT; r++)
        { const u32 ti = t0 + (u32) r * 64;
          if (ti < have)                                       /* (the payload is loaded only now: registers) */
            sval[rnk[r]] = tvin[ti];
        }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < IT; q++)
        { const u32 i = threadIdx.x + (u32) q * TH;
          if (i < have)
            vout[gdst[q]] = sval[i];
        }
    }
}

/***** host side **********************************************************************************************/

static int G_sort_threads = 0;              /* 0: not chosen yet */

/* tile shape of the sorts to come: 512 or 256 threads per workgroup, 16 keys per thread (a 256 x 8 shape within 64 VGPRs
   was measured: slower alone and beside a report launch, profiles/r03_sweeps.txt) */
void damar_sort_set_threads(int threads) { G_sort_threads = (threads == 256 || threads == 512) ? threads : 1024; }

static int sort_threads(void)
{ if (G_sort_threads == 0)
    { const char *e = getenv("DAMAR_SORT_THREADS");
      damar_sort_set_threads(e ? atoi(e) : OS_THREADS);
    }
  return G_sort_threads;
}

size_t damar_sort_workspace_bytes(u64 n)
{ const u64 ntiles = (n + OS_MINTILE - 1) / OS_MINTILE;
  return (size_t) WS_LB + 2 * (((size_t) ntiles * 256 * sizeof(u64) + 255) & ~(size_t) 255) + 256;
}

const u32 *damar_sort_error_word(const void *work) { return (const u32 *) work; }

template <typename KeyT, typename GT, bool HV, int TH, int IT>
static void onesweep_passes(KeyT *k0, u32 *v0, KeyT *k1, u32 *v1, u64 n, int lobit, int hibit, u32 *ohi, u32 *olo,
                            char *ws, hipStream_t st)
{ constexpr int OS_TILE = TH * IT;
  const u32 ntiles = (u32) ((n + OS_TILE - 1) / OS_TILE);
  const int npass  = (hibit - lobit + 7) / 8;
  /* The digits of a sort are as EVEN as its bits allow -- 28 bits are four digits of 7, 43 bits one of 8 and five of 7 -- not
     8, 8, ... and a last one of what is left: the same number of passes, but a pass over a 7-bit digit has 128 look-back words
     per tile to publish and poll instead of 256 and its scatter leaves the tile in runs twice as long (round 6). */
  int dshift[OS_MAXPASS], dwidth[OS_MAXPASS];
  u64 dsh = 0, dwd = 0;
  { const int bits = hibit - lobit, base = bits / npass, extra = bits % npass;
    int at = 0;
    static int even = -1;                       /* DAMAR_SORT_EVEN=0: digits of 8 bits and a short last one (rounds 2-5) */
    if (even < 0)
      { const char *e = getenv("DAMAR_SORT_EVEN");
        even = e ? atoi(e) : 1;
      }
    for (int p = 0; p < npass; p++)
      { dshift[p] = at;
        dwidth[p] = even ? base + (p < extra ? 1 : 0) : (p < npass - 1 ? 8 : bits - 8 * (npass - 1));
        at += dwidth[p];
        dsh |= (u64) dshift[p] << (8 * p);
        dwd |= (u64) ((1u << dwidth[p]) - 1u) << (8 * p);
      }
  }
  const size_t region = ((size_t) ntiles * 256 * sizeof(GT) + 255) & ~(size_t) 255;
  u32 *err = (u32 *) ws, *ghist = (u32 *) (ws + WS_HIST), *ctr = (u32 *) (ws + WS_CTR);
  GT  *lbr[2] = { (GT *) (ws + WS_LB), (GT *) (ws + WS_LB + region) };
  HIP_CHECK(hipMemsetAsync(ws, 0, WS_LB, st));
  if (sort_threads() >= 512)
    { const u64 nt = (n + (u64) OH_THREADS * OH_ITEMS - 1) / ((u64) OH_THREADS * OH_ITEMS);
      const u64 maxit = OH_BINMAX / ((OH_THREADS / 32) * OH_ITEMS);   /* tiles one workgroup may see */
      u64 grid = nt < 512 ? nt : 512;                                 /* two rounds of one workgroup per CU */
      if ((nt + grid - 1) / grid > maxit)
        grid = (nt + maxit - 1) / maxit;
      const size_t lds = (size_t) npass * 128 * 32 * sizeof(u32);
      { /* up to 128 KB of dynamic LDS (8 digit places): a function attribute is kept per DEVICE, so it is set once for
           every ordinal a thread of this process sorts on (the library itself is one GPU per process) */
        static std::atomic<unsigned long long> big_lds(0ull);
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(big_lds.load(std::memory_order_acquire) & bit))
          { HIP_CHECK(hipFuncSetAttribute((const void *) onesweep_hist<u32>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384));
            HIP_CHECK(hipFuncSetAttribute((const void *) onesweep_hist<u64>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384));
            big_lds.fetch_or(bit, std::memory_order_release);
          }
      }
      hipLaunchKernelGGL(onesweep_hist<KeyT>, dim3((u32) grid), dim3(OH_THREADS), lds, st,
                         k0, n, lobit, npass, dsh, dwd, ghist, (uint4 *) lbr[0], (u64) (region / 16));
      HIP_CHECK(hipGetLastError());
    }
  else
    { const u64 nt = (n + 256 * OH_ITEMS - 1) / (256 * OH_ITEMS);
      const u32 grid = (u32) (nt < 2048 ? nt : 2048);
      hipLaunchKernelGGL(onesweep_hist8<KeyT>, dim3(grid), dim3(256), (size_t) npass * 256 * OH_COPIES * sizeof(u32), st,
                         k0, n, lobit, npass, dsh, dwd, ghist, (uint4 *) lbr[0], (u64) (region / 16));
    }
  for (int p = 0; p < npass; p++)
    { const int  side = p & 1;
      const bool last = (p == npass - 1);
      KeyT *ki = side ? k1 : k0, *ko = side ? k0 : k1;
      u32  *vi = side ? v1 : v0, *vo = side ? v0 : v1;
      const u32 mask = (1u << dwidth[p]) - 1u;
      if (last && ohi != NULL)
        hipLaunchKernelGGL((onesweep_pass<KeyT, GT, false, true, TH, IT>), dim3(ntiles), dim3(TH), 0, st,
                           ki, (const u32 *) NULL, (KeyT *) NULL, olo, ohi, n, lobit + dshift[p], mask,
                           ghist + 256 * p, lbr[side], lbr[side ^ 1], ctr + p, err);
      else
        hipLaunchKernelGGL((onesweep_pass<KeyT, GT, HV, false, TH, IT>), dim3(ntiles), dim3(TH), 0, st,
                           ki, vi, ko, vo, (u32 *) NULL, n, lobit + dshift[p], mask,
                           ghist + 256 * p, lbr[side], lbr[side ^ 1], ctr + p, err);
    }
}

/* Sorts on key bits [lobit, hibit).  Ping-pongs between (k0,v0) and (k1,v1); returns 0 if the result is in
 * (k0,v0), 1 if in (k1,v1).  With ohi / olo the last pass writes the two halves of the keys there instead. */
template <typename KeyT, bool HV>
static int onesweep_impl(KeyT *k0, u32 *v0, KeyT *k1, u32 *v1, u64 n, int lobit, int hibit, u32 *ohi, u32 *olo,
                         void *work, hipStream_t st)
{ const int npass = (hibit - lobit + 7) / 8;
  if (n == 0 || npass <= 0)
    return 0;
  if (npass > OS_MAXPASS || n >= 0xfffffff0ull)
    { fprintf(stderr, "damar: internal error, radix sort of %llu items on %d bits\n", (unsigned long long) n, hibit - lobit);
      fflush(NULL);
      _exit(1);
    }
  const int shape = sort_threads();
  static int lb64 = -1;                    /* test hook (tests/test_gpu_sort.py): the 64-bit look-back words at any size */
  if (lb64 < 0)
    lb64 = getenv("DAMAR_SORT_LB64") != NULL;
  if (n < (1ull << 30) && !lb64)
    { if (shape == 1024)     onesweep_passes<KeyT, u32, HV, 1024, 8>(k0, v0, k1, v1, n, lobit, hibit, ohi, olo, (char *) work, st);
      else if (shape == 512) onesweep_passes<KeyT, u32, HV, 512, 16>(k0, v0, k1, v1, n, lobit, hibit, ohi, olo, (char *) work, st);
      else              onesweep_passes<KeyT, u32, HV, 256, 16>(k0, v0, k1, v1, n, lobit, hibit, ohi, olo, (char *) work, st);
    }
  else                  /* 2^30 items and more: 64-bit look-back words, one shape */
    onesweep_passes<KeyT, u64, HV, 256, 16>(k0, v0, k1, v1, n, lobit, hibit, ohi, olo, (char *) work, st);
  return npass & 1;
}

int damar_radix_sort_u32(u32 *k0, u32 *v0, u32 *k1, u32 *v1, u64 n, int nbits, void *work, hipStream_t st)
{ return onesweep_impl<u32, true>(k0, v0, k1, v1, n, 0, nbits, NULL, NULL, work, st); }

/* keys only */
int damar_radix_sort_keys_u32(u32 *k0, u32 *k1, u64 n, int nbits, void *work, hipStream_t st)
{ return onesweep_impl<u32, false>(k0, NULL, k1, NULL, n, 0, nbits, NULL, NULL, work, st); }

int damar_radix_sort_u64(u64 *k0, u32 *v0, u64 *k1, u32 *v1, u64 n, int nbits, void *work, hipStream_t st)
{ return onesweep_impl<u64, true>(k0, v0, k1, v1, n, 0, nbits, NULL, NULL, work, st); }

/* keys only, on bits [lobit, hibit): whatever sits below lobit rides along as payload */
int damar_radix_sort_keys_u64(u64 *k0, u64 *k1, u64 n, int lobit, int hibit, void *work, hipStream_t st)
{ return onesweep_impl<u64, false>(k0, NULL, k1, NULL, n, lobit, hibit, NULL, NULL, work, st); }

/* as damar_radix_sort_keys_u64, but the last pass writes the keys' high words to ohi and their low words to olo
 * (neither may overlap the buffer that pass reads: k0 for an odd number of passes, k1 for an even one) */
void damar_radix_sort_split_u64(u64 *k0, u64 *k1, u64 n, int lobit, int hibit, u32 *ohi, u32 *olo, void *work,
                                hipStream_t st)
{ onesweep_impl<u64, false>(k0, NULL, k1, NULL, n, lobit, hibit, ohi, olo, work, st); }

/* loads this file's code object now (a lazy load otherwise happens at the first launch, on the launching thread): called by
   the library's start-up thread, beside the caller's first uploads (shim.hip damar_hip_init) */
void damar_preload_sort(void)
{ hipFuncAttributes fa;
  (void) hipFuncGetAttributes(&fa, (const void *) onesweep_hist<u64>);
}

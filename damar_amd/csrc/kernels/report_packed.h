/* report_packed.h -- the report loop with TWO read pairs per wavefront whose halves run INDEPENDENTLY (included by report.hip).
 *
 * Why two: the live band of a Local_Alignment wave is 12 diagonals wide on average and 99.9 % of the wave steps fit 30
 * lanes (band histogram of the oracle, profiles/), so one alignment per 64-lane wavefront leaves five lanes in six idle.
 * Each 32-lane half of a wavefront owns a read pair; everything that is uniform per alignment lives in VGPRs (the same
 * value in the 32 lanes of a half) or, as a predicate, in a 64-bit lane mask on the scalar side.
 * Why independently: rounds 2-3 ran the two halves through the same direction of Local_Alignment at the same time (the
 * direction was a template parameter of the wave loop), so a round cost max(f1, f2) + max(r1, r2) wave steps and 1.58 of
 * the 2 halves stepped per iteration (now 1.97).  Here the direction of a half is a run-time value and the wave loop is
 * written ONCE, in coordinates in which both directions look the same:
 *
 *     sigma = +1 forward, -1 reverse (kept as the xor mask m = 0 / -1: sigma * x == (x ^ m) - m)
 *     K = sigma * k,  V = sigma * v,  X = sigma * x,  Y = sigma * y            (k = x - y diagonal, v = x + y anti-diagonal)
 *
 * In these coordinates the reverse wave of align.c:1126-1898 IS the forward wave of align.c:409-1122: furthest point =
 * maximum V, the sweep runs from the highest K down (hgh..low forward, low..hgh reverse: align.c:781, 1490), the
 * predecessor rule, its tie breaks, the clipping at sequence ends, the lag rules and the pruning are literally the same
 * expressions (shown case by case in DESIGN.md section 4).  What is left of the direction: the base compared at (X, Y) is
 * a[X ^ m], b[Y ^ m] (the reverse wave compares a[x-1], b[y-1]) -- read 16 at a time off the 2-bit packed bases, a
 * reverse pass off their REVERSED copy (DevBlock.rbias), so that its windows ascend like a forward pass's -- and the
 * trace grid is indexed by G = sigma * (grid index) (+ 2^14 for m = -1, so that it stays positive).
 * So a half steps through whatever it has to do next -- forward pass, trace walk, reverse pass, emission, seed scan --
 * while the other half does the same on its own: the wave loop is left when EITHER half has an event, the event is
 * served (only that half's lanes are live), and the loop is entered again.
 *
 * What else keeps the wave step short (158 vector instructions per step of two halves, 266 in round 3):
 *   - the next trace marks NA/NB of a diagonal (align.c:861-909) are not carried at all: at every use NA[k] is at most one
 *     spacing beyond the mark of the inherited chain head (the predecessor's x is never behind the diagonal's own last
 *     x, and a new edge diagonal inherits its neighbour's NA), so "push every mark in (head mark, x]" is what the
 *     reference's loop does; the head's mark rides in the head word as before.  Checked with an assertion in a copy of
 *     the oracle over config 1 (31 262 records, several trace spacings) and eleven golden cases: no violation;
 *   - the band is kept in LANE coordinates (ls..hs = lanes of the highest..lowest K) so that widening, pruning, clipping
 *     and the recentring test need no conversion; lanes outside the band always hold V = EDGE (re-established after the
 *     pruning of every step), so the neighbours read by DPP need no range tests; the band plus the two lanes it may grow
 *     into stays within lanes 1..30, which keeps the two halves' DPP rotations apart;
 *   - T, HA, HB are committed unconditionally (a lane outside the band is never anybody's predecessor).
 *   - new best / last / trim point (align.c:911-928) by a prefix maximum in sweep order instead of a serial replay;
 *     TABLE/SCORE (2 x 64 KB in HBM) are replaced by one 1 KB table in LDS (pk_trim_ok, report.hip);
 *   - the popcount M of the match history is not carried: M == popcount(T & (2^61 - 1)) at all times.
 * A band that needs more than 30 lanes borrows the whole wavefront (duo_solo) and comes back when it is narrow again.
 * Reference semantics and citations are those of report.hip: dalign/filter.c:2128-2432 report_thread,
 * dalign/align.c:409-1122 forward_wave, :1126-1898 reverse_wave, :1904-2097 Local_Alignment.
 */

#ifndef DUO_WAVES
#define DUO_WAVES 8        /* resident wavefronts per SIMD the kernel and its pieces are compiled for (VGPR budget 512 / DUO_WAVES; the
                              launch bound of the kernel is handed down to the functions it calls): the wave loop fits 64 VGPRs
                              without a spill.  Report ms per config-2 step, every kernel alone on the machine: 180 / 167 / 153 at
                              5 / 6 / 8 wavefronts per SIMD (profiles/r04_sweeps.txt) */
#endif

/* experiment switches (scripts/build_exp.sh; results are WRONG with any of them -- they only price one part of the kernel):
   DAMAR_EXP_SCANONLY seeds are scanned but no alignment is started, DAMAR_EXP_NOWALK the pebble chains are not walked,
   DAMAR_EXP_NOPEBBLE (only with NOWALK) the pebbles are not stored */
#ifdef DAMAR_EXP_NOPEBBLE
#ifndef DAMAR_EXP_NOWALK
#error "DAMAR_EXP_NOPEBBLE needs DAMAR_EXP_NOWALK: a walk over stale cells need not end"
#endif
#define DUO_EXP_PEBBLE(x)
#else
#define DUO_EXP_PEBBLE(x) x
#endif
int damar_report2_waves_per_simd(void) { return DUO_WAVES; }
int damar_report2_slots_per_wave(void) { return 2; }

__device__ __forceinline__ u32 hmask(u64 m, int hb) { return (u32) (m >> hb); }           /* this half's 32 bits */
__device__ __forceinline__ int hget(int v, int hb, int s) { return __builtin_amdgcn_ds_bpermute((hb + s) << 2, v); }
/* Inclusive prefix maximum in lane order inside each 32-lane half: Kogge-Stone inside the rows of 16 by DPP row_shr,
   then the last lane of rows 0 / 2 into rows 1 / 3.  In place: a lane whose DPP source does not exist is simply not written
   (bound_ctrl off), which is the identity here -- one instruction per step instead of the mov-identity / mov_dpp / max
   triple the update_dpp builtin compiles to.  (s_nop 1: a DPP operand needs two wait states after the VALU write of its
   register, and the assembler does not add them inside inline asm.) */
__device__ __forceinline__ int pk_prefix_max(int x)
{ asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1" : "+v"(x));
  return x;
}

#define DUO_PIECE __device__ __noinline__
#define DUO_PART  __device__ __forceinline__   /* a part of the one piece duo_run: a call costs its callee-saved registers in scratch */

#define DUO_EDGE   (-BIG)
#define DUO_GREV   (1 << 14)                    /* G = DUO_GREV - grid index in a reverse pass */
#define DUO_LIMK   (1 << 29)

enum { MD_SCAN = 0, MD_TASK, MD_RUN, MD_END, MD_OVF, MD_DONE };

/* per-half event record in LDS (in the loop's coordinates): what a direction touches only
   at events.  tip = a (V of the point), k (its K), d, ha, hb */
enum { DC_REACHM = 0, DC_ACLIP, DC_BCLIP, DC_TRIM, DC_REACH = DC_TRIM + 5, DC_WORDS = 16 };
__shared__ int duo_cold[2 * DC_WORDS];

/* Lane predicates are kept as 64-bit lane masks in scalar registers: a ballot of ONE comparison is the comparison
   itself (v_cmp writes the mask), masks combine on the scalar unit, a branch on "any lane" is s_cmp of the mask, and
   inv() hands a mask back to the vector side as a condition for free.  (Left to bool expressions the compiler turns
   every compound predicate into 0/1 per lane and compares it again: two vector instructions per wany(), measured 16 per
   wave step.) */
__device__ __forceinline__ u64  bal(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool inv(u64 m)  { return __builtin_amdgcn_inverse_ballot_w64(m); }
/* find-first-bit from the top / from the bottom as the hardware has them: -1 for 0 */
__device__ __forceinline__ int ffbh_raw(u32 x) { int r;  asm("v_ffbh_u32_e32 %0, %1" : "=v"(r) : "v"(x));  return r; }
__device__ __forceinline__ int ffbl_raw(u32 x) { int r;  asm("v_ffbl_b32_e32 %0, %1" : "=v"(r) : "v"(x));  return r; }

/* Everything a half carries between the pieces below (noinline functions with their own register allocation) lives in
   LDS: one record per half (every lane of the half reads and writes the same words: broadcast reads, one write) and the
   band state of the 64 lanes.  (Round 4's first version kept it in a struct in private memory handed to the pieces by
   pointer: 6 M piece calls per config-2 step, each loading and storing ~11 KB of scratch per wavefront, were most of the
   20 GB per launch the report kernel moved through L2 -- profiles/r04_traffic_summary.txt.) */
struct DuoCtx
{ int md;
  int m;                                        /* direction: 0 forward, -1 reverse */
  int ls, hs, kbase;                            /* band = lanes ls..hs (highest..lowest K) */
  int dif, besta, bestk, lasta, more, ncell, bad;
  int mlo, mhi;                                 /* the band may not grow below lane mlo / above lane mhi (minp, maxp) */
  int alim, blim;                               /* bases left: alim - X in A, blim - Y in B */
  int offa, offb;                               /* the mark after head index G is crossed when X >= G * TS + offa */
  int va0, vb0, alen, blen;                     /* the reads: offsets in the packed bases (biased by the padding), lengths */
  int pa0, pb0;                                 /* where the pass's packed bases start: window of (X, Y) at pa0 + X, pb0 + Y */
  /* the task and what its passes have produced */
  int diag, anti;
  u32 item;                                     /* the work item (read pair) of the half: named when its pebbles outgrow the packed format */
  int roota, rootb;                             /* trace-grid index the A / B chain of the pass starts from (its root) */
  int aepos, bepos, abpos, bbpos, diffs, atlen, btlen, aback, bback;
  /* what the wavefront has stepped through so far (the same in every lane): SURVEY 8(d)'s secondary unit of K6 */
  u32 n_cells_lo, n_cells_hi;                   /* band cells = diagonals computed, summed over the wave steps */
  u32 n_iter, n_half;                           /* iterations of the wave loop, and halves that stepped in them (counted by half 0's record) */
};
__shared__ DuoCtx duo_half[2];
/* band state of the lanes: lane s of a half owns K = kbase - s */
__shared__ int duo_V[64], duo_HA[64], duo_HB[64], duo_acc[64];
__shared__ u32 duo_Tlo[64], duo_Thi[64];
#define DUO_CX()  DuoCtx &cx = duo_half[lane_id() >> 5]

/* Trim points of the running pass (align.c:911-928 / 1620-1637), one slot per LANE: the wave loop only notes the event --
   a record breaker with a good history and a good tail writes (V, K, head A, head B | dif) into the slot of the lane it
   sits in -- and who was last is asked once, when the pass is over or the band leaves the lanes (duo_trim_settle): the V
   of record breakers grows strictly from event to event, so the slot with the largest V is the reference's trim point,
   whatever diagonal lived in its lane at the time.  (Round 5 elected the last such lane of a half in every step and
   wrote the half's record from it: three find-first-bits, three gathers and four LDS writes in two steps of three.) */
typedef int duo_v4i __attribute__((ext_vector_type(4)));
__shared__ duo_v4i duo_tq[64];                  /* V, K, chain heads A / B as they ride in the lanes */
__shared__ int     duo_td[64];                  /* dif */

/* the trim events noted since the last call -> the half's record in duo_cold (every lane of the halves with `which`) */
__device__ __forceinline__ void duo_trim_settle(bool which)
{ const int lane = lane_id();
  int *const cold = duo_cold + ((lane & 32) >> 1);
  const int tv = duo_tq[lane].x;
  int mx = tv;
  for (int o = 1; o < 32; o <<= 1)
    { const int t = __shfl_xor(mx, o);  mx = t > mx ? t : mx; }
  if (which)
    { if (tv == mx && tv != -BIG)
        { const duo_v4i q = duo_tq[lane];
          cold[DC_TRIM] = q.x;  cold[DC_TRIM + 1] = q.y;  cold[DC_TRIM + 2] = duo_td[lane];
          cold[DC_TRIM + 3] = q.z & PK_HMASK;  cold[DC_TRIM + 4] = q.w & PK_HMASK;
        }
      duo_tq[lane].x = -BIG;
    }
}

struct DuoSnake { int Y, na, nb;  u64 b; };

/* The snake (align.c:832-856 / 1542-1566) of diagonal K from Y, in the loop's coordinates: 16 bases per step off the
   2-bit packed reads -- forward off the packed bases, reverse off their REVERSED copy (DevBlock.rbias), so that both
   slide along ascending addresses: the window starts at biased position pa0 + X / pb0 + Y -- bounded by the bases left
   in either read; a lane that is past an end takes the byte path, which reads what the reference reads there. */
__device__ __forceinline__ DuoSnake duo_snake(const u32 *apk, const u32 *bpk, const u8 *abase, const u8 *bbase,
                                              int m, int alim, int blim, int pa0, int pb0, int va0, int vb0, int alen, int blen,
                                              int K, int Y, u64 b)
{ DuoSnake o;
  const int X = Y + K;
  int na = alim - X, nb = blim - Y;
  if ((u32) na > (u32) alen || (u32) nb > (u32) blen)
    { const int k = (K ^ m) - m, y = (Y ^ m) - m;
      const u8 *ar = abase + (va0 - 16 * PK_PAD), *br = bbase + (vb0 - 16 * PK_PAD);
      SnakeOut so;
      if (m)
        so = snake<1>(ar - 1 + k, br - 1, y, 0, b);
      else
        so = snake<0>(ar + k, br, y, 0, b);
      o.Y = (so.y ^ m) - m;  o.b = so.b;  o.na = so.na;  o.nb = so.nb;
      return o;
    }
  u32 pa = 2u * (u32) (pa0 + X), pb = 2u * (u32) (pb0 + Y);         /* as bit positions: the window shift needs no doubling (and is the
                                                                       same for every 16 bases: only the word offsets move on) */
  asm("" : "+v"(pa), "+v"(pb));                                      /* (one add-and-shift each; the offsets from them, not from the sums again) */
  u32 oa = (pa >> 3) & ~3u, ob = (pb >> 3) & ~3u;
  for (;;)
    { u32 wa, wb;
      { typedef u32 v2u __attribute__((ext_vector_type(2)));
        v2u ra, rb;                                                  /* both loads in flight together, ONE wait (as load16x2) */
        asm volatile("global_load_dwordx2 %0, %2, %4\n\tglobal_load_dwordx2 %1, %3, %5\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(ra), "=&v"(rb) : "v"(oa), "v"(ob), "s"(apk - PK_PAD), "s"(bpk - PK_PAD) : "memory");
        wa = __builtin_amdgcn_alignbit(ra.y, ra.x, pa);
        wb = __builtin_amdgcn_alignbit(rb.y, rb.x, pb);
      }
      const u32 run = (u32) ffbl_raw(wa ^ wb) >> 1;            /* equal bases at the head of the window; huge if all 16 are */
      const int lim = na < nb ? na : nb;
      const int n = (int) __builtin_elementwise_min(__builtin_elementwise_min(run, 16u), (u32) lim);
      b = (b << n) | (u64) ((1u << n) - 1);
      Y += n;  na -= n;  nb -= n;
      if (n < 16 || lim == 16)
        break;
      oa += 4;  ob += 4;
    }
  o.Y = Y;  o.b = b;  o.na = na;  o.nb = nb;
  return o;
}

/* An alignment of the half has been abandoned because its pebbles do not fit the pool.  While the pool can still grow the
   launch is repeated with a larger one (DAMAR_ERR_CELLS); at the packed format's limit of 2^18 the read pair is left to the
   wide kernel (report.hip: report_wide_kernel) -- its bit in widemap, one count per pair -- and what this kernel has
   written or will still write for the pair is dropped by the host.  Called by every lane of the half. */
__device__ __forceinline__ void duo_pebbles_over(const ReportArgs &a, u32 item)
{ if ((lane_id() & 31) != 0)
    return;
  atomicOr(&a.counters[3], DAMAR_ERR_CELLS);
  if (a.widemap != NULL && a.cell_cap >= a.cell_max)
    { const u32 bit = 1u << (item & 31);
      if (!(atomicOr(&a.widemap[item >> 5], bit) & bit))
        atomicAdd(&a.counters[DAMAR_CNT_WIDE], 1u);
    }
}

/* clipping at sequence ends (align.c:628-658 / 943-975) for the halves with `on`, in lane coordinates: the A-side clip
   lane (the highest sweep index that reached A's end) cuts the band's low lanes, the B-side one its high lanes */
#define DUO_CLIP()                                                                                     \
  if (onm & bal(more == 0))                                                                            \
    { const bool cl_ = on && more == 0;                                                                \
      const int  mp_ = pk_popc61(rT);                                                                  \
      if (cl_)                                                                                         \
        { const int by_ = (besta - bestk) >> 1, bx_ = besta - by_;                                     \
          if (bbase[(vb0 - 16 * PK_PAD) + (by_ ^ m)] != 4 && abase[(va0 - 16 * PK_PAD) + (bx_ ^ m)] != 4) \
            more = 1;                                                                                  \
        }                                                                                              \
      const int acl_ = cold[DC_ACLIP], bcl_ = cold[DC_BCLIP];                                          \
      { const bool ca_ = cl_ && ls <= acl_;                                                            \
        const int  sl_ = ca_ ? acl_ : 0;                                                               \
        const int  mm_ = hget(mp_, hb, sl_), vv_ = hget(rV, hb, sl_);                                  \
        const int  ha_ = hget(rHA, hb, sl_), hb2_ = hget(rHB, hb, sl_);                                \
        if (ca_)                                                                                       \
          { ls = acl_ + 1;                                                                             \
            if (cold[DC_REACHM] <= mm_)                                                                \
              { cold[DC_REACHM] = mm_;  cold[DC_REACH] = vv_;  cold[DC_REACH + 1] = kbase - acl_;      \
                cold[DC_REACH + 2] = dif;                                                              \
                cold[DC_REACH + 3] = ha_ & PK_HMASK;  cold[DC_REACH + 4] = hb2_ & PK_HMASK; }          \
          }                                                                                            \
      }                                                                                                \
      { const bool cb_ = cl_ && hs >= bcl_;                                                            \
        const int  sl_ = cb_ ? bcl_ : 0;                                                               \
        const int  mm_ = hget(mp_, hb, sl_), vv_ = hget(rV, hb, sl_);                                  \
        const int  ha_ = hget(rHA, hb, sl_), hb2_ = hget(rHB, hb, sl_);                                \
        if (cb_)                                                                                       \
          { hs = bcl_ - 1;                                                                             \
            if (cold[DC_REACHM] <= mm_)                                                                \
              { cold[DC_REACHM] = mm_;  cold[DC_REACH] = vv_;  cold[DC_REACH + 1] = kbase - bcl_;      \
                cold[DC_REACH + 2] = dif;                                                              \
                cold[DC_REACH + 3] = ha_ & PK_HMASK;  cold[DC_REACH + 4] = hb2_ & PK_HMASK; }          \
          }                                                                                            \
      }                                                                                                \
      if (cl_)                                                                                         \
        { cold[DC_ACLIP] = -1;  cold[DC_BCLIP] = 64; }                                                 \
    }

/* the names the pieces use for a job's constants */
#define DUO_NAMES()                                                                                  \
  const ReportArgs &a = g_jobs[uni(job)];                                                            \
  const int lane = lane_id(), hb = lane & 32, s = lane & 31;                                         \
  const int TS = uni(a.tspace);                                                                      \
  const u32 *apk = uni_ptr(a.ablk.pk), *bpk = uni_ptr(a.bblk.pk);                                    \
  const u8 *abase = uni_ptr(a.ablk.bases), *bbase = uni_ptr(a.bblk.bases);                           \
  GLOBAL_AS v2u32 *const gcell = (GLOBAL_AS v2u32 *) uni_ptr((Cell *) a.cells);                      \
  const int cell_cap = (int) uni((int) a.cell_cap);                                                  \
  u32 *const errw = uni_ptr(&a.counters[3]);                                                         \
  int *const cold = duo_cold + (hb >> 1);                                                            \
  (void) lane; (void) hb; (void) s; (void) TS; (void) apk; (void) bpk; (void) abase; (void) bbase;   \
  (void) gcell; (void) cell_cap; (void) errw; (void) cold;

/* Wave 0 on the seed diagonal (align.c:491-626 / 1203-1340) and the clipping behind it, for the halves with
   md == MD_TASK: sets up direction cx.m of the task (cx.diag, cx.anti).  Every lane of a half computes the same. */
DUO_PART void duo_begin(int job, u32 cbase)
{ DUO_NAMES()
  DUO_CX();
  const bool on = cx.md == MD_TASK;
  const u64 onm = bal(on);
  const int m = cx.m;
  const int va0 = cx.va0, vb0 = cx.vb0, alen = cx.alen, blen = cx.blen;
  const int diag = cx.diag, anti = cx.anti;
  const int guard = 4 * (alen + blen) + 1024;
  const int boff = (a.comp & 1) ? (blen % TS) : 0;
  const int offa0 = -PK_BIAS * TS, offb0 = boff - PK_BIAS * TS;          /* mark = grid index * TS + off */
  int ls = 15, hs = 15, kbase = 0, dif = 0, besta = 0, bestk = 0, lasta = 0, more = 1, ncell = 2, bad = 0;
  int mlo = 0, mhi = 0, alim = 0, blim = 0, offa = 0, offb = 0, pa0 = 0, pb0 = 0, roota = 0, rootb = 0;
  int rV = DUO_EDGE, rHA = 0, rHB = 0;
  u64 rT = 0;
  int md = cx.md;

  if (on)
    { const int K0 = (diag ^ m) - m, V0 = (anti ^ m) - m;
      int Y = (V0 - K0) >> 1;
      const int y = (anti - diag) >> 1, x = y + diag;
      int ga, gb;
      if (m == 0)
        { ga = (x + TS) / TS - 1 + PK_BIAS;  gb = (y + (TS - boff)) / TS - 1 + PK_BIAS;
          if (s == 0)
            { gcell[cbase] = cell_root(ga * TS + offa0, diag);
              gcell[cbase + 1] = cell_root(gb * TS + offb0, diag);
            }
          offa = offa0 + TS;  offb = offb0 + TS;
          alim = alen;  blim = blen;
          pa0 = va0;  pb0 = vb0;
          roota = ga;  rootb = gb;
        }
      else
        { const int hai = (x + TS - 1) / TS + PK_BIAS, hbi = (y + (TS - boff) - 1) / TS + PK_BIAS;     /* the true start, rounded up to the grid */
          if (s == 0)
            { gcell[cbase] = cell_root(x, diag);
              gcell[cbase + 1] = cell_root(y, diag);
            }
          ga = DUO_GREV - hai;  gb = DUO_GREV - hbi;
          roota = hai;  rootb = hbi;
          offa = -offa0 - DUO_GREV * TS + TS;  offb = -offb0 - DUO_GREV * TS + TS;
          alim = 0;  blim = 0;
          /* base x - 1 of the read = block base a0 + x - 1 = reversed base total - a0 - x, at X = -x */
          pa0 = (int) (a.ablk.rbias + a.ablk.total) - (va0 - 16 * PK_PAD);
          pb0 = (int) (a.bblk.rbias + a.bblk.total) - (vb0 - 16 * PK_PAD);
        }
      { const bool selfie = (abase + va0 == bbase + vb0);
        const int minp = (selfie && diag >= 0) ? 1 : -DUO_LIMK, maxp = (selfie && diag <= 0) ? -1 : DUO_LIMK;
        const int minK = m ? -maxp : minp, maxK = m ? -minp : maxp;
        kbase = K0 + 15;
        mlo = kbase - maxK;  mhi = kbase - minK;
      }
      cold[DC_REACHM] = -1;  cold[DC_ACLIP] = -1;  cold[DC_BCLIP] = 64;
      { const DuoSnake so = duo_snake(apk, bpk, abase, bbase, m, alim, blim, pa0, pb0, va0, vb0, alen, blen, K0, Y, 0ull);
        Y = so.Y;
        if (so.nb == 0)      { more = 0;  cold[DC_BCLIP] = 15; }
        else if (so.na == 0) { more = 0;  cold[DC_ACLIP] = 15; }
      }
      const int v = (Y << 1) + K0, X = Y + K0;
      int ha = 0, hb_ = 1, g0 = 0;
      for (;;)
        { if (!(X >= ga * TS + offa))
            break;
          GUARD(g0, guard, 2)
          ga += 1;
          if (s == 0 && ncell < cell_cap)
            gcell[cbase + (u32) ncell] = cell_pack(ha, diag, 0, (ga ^ m) - m);
          ha = ncell++;
        }
      for (;;)
        { if (!(Y >= gb * TS + offb))
            break;
          GUARD(g0, guard, 3)
          gb += 1;
          if (s == 0 && ncell < cell_cap)
            gcell[cbase + (u32) ncell] = cell_pack(hb_, diag, 0, (gb ^ m) - m);
          hb_ = ncell++;
        }
      besta = lasta = V0;  bestk = K0;
      cold[DC_TRIM] = V0;  cold[DC_TRIM + 1] = K0;  cold[DC_TRIM + 2] = 0;  cold[DC_TRIM + 3] = 0;  cold[DC_TRIM + 4] = 1;
      cold[DC_REACH] = V0;  cold[DC_REACH + 1] = K0;  cold[DC_REACH + 2] = 0;  cold[DC_REACH + 3] = 0;  cold[DC_REACH + 4] = 1;
      if (v > besta)
        { besta = lasta = v;
          cold[DC_TRIM] = v;  cold[DC_TRIM + 3] = ha;  cold[DC_TRIM + 4] = hb_;
        }
      if (s == 15)
        rV = v;
      rT = HIST_FULL;
      rHA = ha | (ga << PK_HBITS);  rHB = hb_ | (gb << PK_HBITS);
      md = MD_RUN;
      if (ncell > cell_cap)              /* a seed diagonal that slides over more marks than the pool holds */
        { duo_pebbles_over(a, cx.item);
          more = 0;  ncell = 2;  bad = 1;  md = MD_END;
        }
    }
  DUO_CLIP()
  if (on)
    { duo_V[lane] = rV;  duo_HA[lane] = rHA;  duo_HB[lane] = rHB;  duo_Tlo[lane] = (u32) rT;  duo_Thi[lane] = (u32) (rT >> 32);
      duo_tq[lane].x = -BIG;                       /* no trim event of this pass yet (duo_cold holds the seed diagonal's point) */
      cx.md = md;
      cx.ls = ls;  cx.hs = hs;  cx.kbase = kbase;  cx.dif = dif;  cx.besta = besta;  cx.bestk = bestk;
      cx.lasta = lasta;  cx.more = more;  cx.ncell = ncell;  cx.bad = bad;
      cx.mlo = mlo;  cx.mhi = mhi;  cx.alim = alim;  cx.blim = blim;  cx.offa = offa;  cx.offb = offb;
      cx.pa0 = pa0;  cx.pb0 = pb0;  cx.roota = roota;  cx.rootb = rootb;
    }
}

/* The wave steps (align.c:667-999 / 1378-1697) of the halves with md == MD_RUN, until one of them has an event: its
   direction is over (or failed), or its band no longer fits lanes 1..30.  The caller tells which from the state
   (duo_classify); on entry every such half can step (duo_classify has been through).

   Round 6: the step is written in the ONE coordinate the wave carries, v = 2 Y + K, against per-lane constants that only
   change when the band is moved to the middle of its lanes (K = kbase - lane is fixed in between):
     - the window of the packed bases starts at bit 2 (pa0 + X) = v + (2 pa0 + K) in A and at v + (2 pb0 - K) in B;
     - bases left: 2 (alim - X) = (2 alim - K) - v, 2 (blim - Y) = (2 blim + K) - v, so the snake's bound is
       cmin - v with cmin the smaller of the two, "a read's end was reached" is v == cmin, and "this lane lies outside
       the reads" (the byte path) is ONE unsigned comparison of cmin - v with a per-lane span;
     - the next trace mark is crossed when v >= G * 2 TS + (2 offa - K)  /  G * 2 TS + (2 offb + K);
   Y, X, na, nb are not formed at all.  The band is a lane MASK: lanes outside it hold V = EDGE, so a lane takes part in a
   step iff its own or a neighbour's V is real -- the new V says so by itself (v > EDGE + 2) -- and pruning leaves the
   span of the lanes within reach of the best point; ls / hs exist only where a number is needed (recentring, clipping,
   the record).  What a half needs to know about its best / last points only when something happens is left in the
   lanes: R = V of the lane's last record, L = V of its last record with a good history; bestk and lasta are read out of
   them when the loop is left, when a sequence end is clipped, and (lasta) when the cheap test against the last KNOWN
   lasta fails; trim points go to per-lane slots (duo_tq above).  dif is the entry value plus the scalar iteration
   count, `more` is 1 unless the rare block behind a sequence end says otherwise.
   Per iteration (static, scripts/asm_blocks.py): see DESIGN.md section 4. */
#define DUO_EDGES 0xC0000003C0000003ull                    /* lanes 0, 1, 30, 31 of either half */
#ifndef DUO_DBG
#define DUO_DBG 0
#endif
/* -DDAMAR_LOOPC (scripts/build_prof_var.sh loopc -DDAMAR_LOOPC, scripts/loop_blocks.py): how often the wave loop enters its
   conditional blocks, counted in scalar registers and added to g_loopc when the loop is left.  Without the switch: nothing. */
#ifdef DAMAR_LOOPC
__device__ unsigned long long g_loopc[16];
#define DUO_LC_DECL()   u32 lc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, lcv = 0;
#define DUO_LC(i)       lc[i] += 1u;
#define DUO_LCV()       lcv += 1u;               /* (inside divergent code: a count per lane, summed at the end) */
#define DUO_LC_FLUSH()  { if (lane == 0) { _Pragma("unroll") for (int q_ = 0; q_ < 16; q_++) if (lc[q_]) atomicAdd(&g_loopc[q_], (unsigned long long) lc[q_]); } \
                          if (lcv) atomicAdd(&g_loopc[3], (unsigned long long) lcv); }
#else
#define DUO_LC_DECL()
#define DUO_LC(i)
#define DUO_LCV()
#define DUO_LC_FLUSH()
#endif

DUO_PART void duo_loop(int job, const u32 *trimtab, u32 cbase)
{ DUO_NAMES()
  DUO_CX();
  DUO_LC_DECL()
  const int ave = uni(a.ave_path);
  const u64 onm = bal(cx.md == MD_RUN);
  if (!onm)
    return;
  const bool on = inv(onm);
  const int m = cx.m;
  const int TS2 = 2 * TS;
  const int lane4 = lane << 2, top4 = (hb + 31) << 2, s31 = 31 - s;
  int dif = cx.dif;
  int guard;                                      /* (one bound for both halves, in a scalar register: a lane-varying one makes every exit
                                                     of the pebble loops a divergent one) */
  { const int gv = 4 * (cx.alen + cx.blen) + 1024;
    const int g0 = __builtin_amdgcn_readlane(gv, 0), g1 = __builtin_amdgcn_readlane(gv, 32);
    guard = g0 > g1 ? g0 : g1;
  }
  int besta = cx.besta, ncell = cx.ncell;
  int lastlim = cx.lasta + MAX_TRIM_LAG;          /* the pass goes on while lasta + lag >= besta; this is the last KNOWN lasta's */
  int R = -BIG, L = -BIG;
  int K = cx.kbase - s;
  int rV = duo_V[lane], rHA = duo_HA[lane], rHB = duo_HB[lane];
  u64 rT = ((u64) duo_Thi[lane] << 32) | duo_Tlo[lane];
  u32 st_cells = 0;                               /* (scalar: one s_bcnt1 + one s_add per step; the steps themselves: left0 - left) */
  int left;                                       /* steps until the first stepping half reaches the loop bound (dif <= alen + blen + 64:
                                                     cannot happen; leaving early for the other half's sake only re-enters the loop) */
  { const int lf = on ? cx.alen + cx.blen + 64 - dif : BIG;
    const int l0 = __builtin_amdgcn_readlane(lf, 0), l1 = __builtin_amdgcn_readlane(lf, 32);
    left = l0 < l1 ? l0 : l1;
  }
  const int left0 = left;
  int cpa, cpb, cmin, cspan, cpA, cpB;
  u64 allowm;
  u64 spanm = onm & bal(rV > DUO_EDGE + 2);       /* the band: lanes outside it hold EDGE */
  u64 stopm = 0;                                  /* lanes of the halves that cannot go on */
  u64 edges = DUO_EDGES;
  asm("" : "+s"(edges));                        /* (in a register pair: as a literal it is two 32-bit ands) */

  for (;;)
    { /* (every few dozen steps) keep the band and the two lanes it may grow into within lanes 1 .. 30 of the half; a band
         that has outgrown the lanes, or a pebble pool that has run over (its stores are bounded), leaves */
      if (spanm & edges)
        { DUO_LC(1)
          const u32 hk = hmask(spanm, hb);
          const int ls = ffbl_raw(hk), hs = 31 ^ ffbh_raw(hk);
          if (onm & (bal(hs - ls > 27) | bal(ncell > cell_cap)))
            break;
          const u64 mvm = onm & (bal(ls < 2) | bal(hs > 29));
          const int dl = inv(mvm) ? ((31 - (hs - ls)) >> 1) - ls : 0;
          const int src = (hb + ((s - dl) & 31)) << 2;
          rV  = __builtin_amdgcn_ds_bpermute(src, rV);
          rHA = __builtin_amdgcn_ds_bpermute(src, rHA);
          rHB = __builtin_amdgcn_ds_bpermute(src, rHB);
          R   = __builtin_amdgcn_ds_bpermute(src, R);
          { const u32 tl = (u32) __builtin_amdgcn_ds_bpermute(src, (int) (u32) rT);
            const u32 th = (u32) __builtin_amdgcn_ds_bpermute(src, (int) (u32) (rT >> 32));
            rT = ((u64) th << 32) | tl;
          }
          K += dl;
          if (on)                                 /* the growth limits move with the lanes: through the record, they are needed here only */
            { cx.mlo += dl;  cx.mhi += dl; }
          spanm = onm & bal(rV > DUO_EDGE + 2);
        }
      /* the lane's constants (see above) */
      { const int ca2 = 2 * cx.alim - K, cb2 = 2 * cx.blim + K;
        const int clo = max(ca2 - 2 * cx.alen, cb2 - 2 * cx.blen);
        cmin = min(ca2, cb2);  cspan = cmin - clo;
        if (cspan < 0)                            /* a diagonal that misses a read altogether: never on the window path */
          { cmin = -(1 << 30);  cspan = 0; }
        cpa = 2 * cx.pa0 + K;  cpb = 2 * cx.pb0 - K;
        cpA = 2 * cx.offa - K;  cpB = 2 * cx.offb + K;
        const int alo = max(cx.mlo, 1), ahi = min(cx.mhi, 30);
        allowm = onm & bal(s >= alo) & bal(s <= ahi);
      }

      do
        { /* widen (align.c:675-776) and pick the predecessor (align.c:793-825): K - 1 sits one lane up, K + 1 one lane down */
          int  v, ha, hb_;
          u64  b;
          typedef u32 v2u __attribute__((ext_vector_type(2)));
          const int am = lane_up(rV), ap = lane_dn(rV), ac = rV;
          const int nbv = am > ap ? am : ap;
          const u64 takem = bal(ac < nbv), upm = bal(am < ap);
          v = inv(takem) ? nbv + 1 : ac + 2;
          int dsel = inv(upm) ? -4 : 4;
          dsel = inv(takem) ? dsel : 0;
          const int src = lane4 + dsel;
          const u64 actm = allowm & bal(v > DUO_EDGE + 2);
          dif += 1;
          st_cells += (u32) __popcll(actm);
          asm("s_sub_u32 %0, %0, 1\n\ts_cselect_b64 %1, -1, %1" : "+s"(left), "+s"(stopm) : : "scc");     /* (the bound: cannot happen) */
          /* the snake (align.c:832-856 / 1542-1566): 16 bases per step off the packed reads; a lane outside the reads takes the
             byte path, which compares what the reference compares there and says in `ef` whether it met an end (1: A's, 2: B's) */
          const u64 bytem = actm & bal((u32) (cmin - v) > (u32) cspan);
          const u64 fastm = actm & ~bytem;
          const u32 pa = (u32) v + (u32) cpa, pb = (u32) v + (u32) cpb;          /* bit positions of the two windows */
          u32 oa = (pa >> 3) & ~3u, ob = (pb >> 3) & ~3u;
          v2u ra, rb;
          { ha  = __builtin_amdgcn_ds_bpermute(src, rHA);
            hb_ = __builtin_amdgcn_ds_bpermute(src, rHB);
            const u32 tlo = (u32) __builtin_amdgcn_ds_bpermute(src, (int) (u32) rT);
            const u32 thi = (u32) __builtin_amdgcn_ds_bpermute(src, (int) (u32) (rT >> 32));
            b = ((u64) thi << 32) | tlo;
          }
          int ef;
          asm volatile("" : "=v"(ef));              /* (only the byte path's lanes define it; every use is behind bytem) */
          if (bytem)
            { DUO_LC(2)
              if (inv(bytem))
                { const int Y = (v - K) >> 1;
                  const int k = (K ^ m) - m, y = (Y ^ m) - m;
                  const u8 *ar = abase + (cx.va0 - 16 * PK_PAD), *br = bbase + (cx.vb0 - 16 * PK_PAD);
                  SnakeOut so;
                  if (m)
                    so = snake<1>(ar - 1 + k, br - 1, y, 0, b << 1);
                  else
                    so = snake<0>(ar + k, br, y, 0, b << 1);
                  v = (((so.y ^ m) - m) << 1) + K;  b = so.b;
                  ef = so.nb == 0 ? 2 : (so.na == 0 ? 1 : 0);
                }
            }
          if (inv(fastm))
            { u32 m2;
#define DUO_LOADS()                                              /* both loads in flight together, ONE wait */  \
                asm volatile("global_load_dwordx2 %0, %2, %4\n\tglobal_load_dwordx2 %1, %3, %5\n\ts_waitcnt vmcnt(0)" \
                             : "=&v"(ra), "=&v"(rb) : "v"(oa), "v"(ob), "s"(apk - PK_PAD), "s"(bpk - PK_PAD) : "memory");
#define DUO_WINDOW(FIRST)                                                                                     \
              { const u32 wa = __builtin_amdgcn_alignbit(ra.y, ra.x, pa);                                     \
                const u32 wb = __builtin_amdgcn_alignbit(rb.y, rb.x, pb);                                     \
                const u32 run2 = (u32) ffbl_raw(wa ^ wb) & ~1u;  /* twice the equal bases at the head of the window; huge if all 16 are */ \
                m2 = __builtin_elementwise_min(run2, (u32) (cmin - v));                                       \
                const u32 n2 = __builtin_elementwise_min(m2, 32u), n = n2 >> 1;                               \
                u32 ones;                                                                                     \
                asm("v_bfm_b32 %0, %1, 0" : "=v"(ones) : "v"(n));                                             \
                b = (b << (n + (FIRST))) | (u64) ones;         /* (the step's own 0 rides on the first window's shift) */ \
                v += (int) n2;                                                                                \
              }
              DUO_LOADS()
              DUO_WINDOW(1)
              const u64 contm = bal(m2 > 32u);               /* all 16 equal and more than 16 left: one window in eight hundred */
              if (contm)
                { if (inv(contm))
                    do
                      { DUO_LCV()
                        oa += 4;  ob += 4;
                        DUO_LOADS()
                        DUO_WINDOW(0)
                      }
                    while (m2 > 32u);
                }
#undef DUO_WINDOW
#undef DUO_LOADS
            }

          /* What the rest of the step asks of the new V: is a trace mark crossed (align.c:859-909 / 1569-1618: a pebble is due
             when v reaches the mark after the inherited head's), is it a new best point (align.c:911-928 / 1620-1637: the
             candidates' prefix maximum in sweep order), has the history enough matches, is it a read's end.  Some lane
             passes the old best in every step (profiles/r05_loop_blocks.txt): no branch around the maximum */
          DUO_LC(0)
          const u64 candm = actm & bal(v > besta);
#ifdef DAMAR_LOOPC
          if (candm) { DUO_LC(12) }
#endif
          int x = inv(candm) ? v : -BIG, e;
          u64 nam, nbm, mokc, endc;
          { const int ga = (int) ((u32) ha >> PK_HBITS), gb = (int) ((u32) hb_ >> PK_HBITS);
            nam = bal(v >= __mul24(ga, TS2) + cpA);  nbm = bal(v >= __mul24(gb, TS2) + cpB);
            x = pk_prefix_max(x);
            /* the maximum over the lanes before this one; what lane 0 of a half receives does not matter: the band lives
               in lanes 1 .. 30, lane 0 is never a candidate */
            e = __builtin_amdgcn_mov_dpp(x, 0x138, 0xf, 0xf, true);                               /* wave_shr:1 */
            mokc = bal(pk_popc61(b) >= ave);
            endc = bal(v == cmin);
          }
          nam &= actm;  nbm &= actm;
          if (nam | nbm)
            { DUO_LC(4)
              /* ONE round for both chains: a slide crosses at most one mark of either grid unless it is longer than a trace
                 spacing, so the lanes of the A list and of the B list take their cells out of the half's pool together (A's
                 first) and each writes its pebble; only what is left after that goes round the loops.  In a packed head the
                 grid index sits above the cell index: + 1 << PK_HBITS steps it, and the cell's own index field is that of
                 the reference's coordinates, sigma * G = (G ^ m) - m, formed on the field where it sits */
              const int kk = (K ^ m) - m;
              const u32 w1 = ((u32) kk & 0xffffu) | ((u32) (dif) << 16);
              const u32 below = (1u << s) - 1u;
              const u32 M18 = (u32) m << PK_HBITS;
              { const u32 hmA = hmask(nam, hb), hmB = hmask(nbm, hb);
                const int idxA = ncell + __popc(hmA & below), tA = ncell + __popc(hmA);
                const int idxB = tA + __popc(hmB & below);
                ncell = tA + __popc(hmB);
                if (inv(nam))
                  { const u32 t = (u32) ha + (1u << PK_HBITS);
                    if (idxA < cell_cap)
                      { const v2u32 c = { (((t & ~(u32) PK_HMASK) ^ M18) - M18) | ((u32) ha & (u32) PK_HMASK), w1 };
                        DUO_EXP_PEBBLE(gcell[cbase + (u32) idxA] = c;)
                      }
                    ha = (int) ((t & ~(u32) PK_HMASK) | (u32) idxA);
                  }
                if (inv(nbm))
                  { const u32 t = (u32) hb_ + (1u << PK_HBITS);
                    if (idxB < cell_cap)
                      { const v2u32 c = { (((t & ~(u32) PK_HMASK) ^ M18) - M18) | ((u32) hb_ & (u32) PK_HMASK), w1 };
                        DUO_EXP_PEBBLE(gcell[cbase + (u32) idxB] = c;)
                      }
                    hb_ = (int) ((t & ~(u32) PK_HMASK) | (u32) idxB);
                  }
              }
              nam &= bal(v >= __mul24((int) ((u32) ha >> PK_HBITS), TS2) + cpA);
              nbm &= bal(v >= __mul24((int) ((u32) hb_ >> PK_HBITS), TS2) + cpB);
              if (nam | nbm)
                { DUO_LC(5)
                  int ga = (int) ((u32) ha >> PK_HBITS), gb = (int) ((u32) hb_ >> PK_HBITS);
                  int hax = ha & PK_HMASK, hbx = hb_ & PK_HMASK;
                  int g2 = 0;
                  while (nam)
                    { GUARD(g2, guard, 5)
                      DUO_LC(10)
                      const u32 hm = hmask(nam, hb);
                      const int idx = ncell + __popc(hm & below);
                      if (inv(nam))
                        { ga += 1;
                          if (idx < cell_cap)
                            { const v2u32 c = { (u32) hax | ((u32) ((ga ^ m) - m) << PK_HBITS), w1 };
                              DUO_EXP_PEBBLE(gcell[cbase + (u32) idx] = c;)
                            }
                          hax = idx;
                        }
                      ncell += __popc(hm);
                      nam &= bal(v >= __mul24(ga, TS2) + cpA);
                    }
                  while (nbm)
                    { GUARD(g2, guard, 6)
                      DUO_LC(11)
                      const u32 hm = hmask(nbm, hb);
                      const int idx = ncell + __popc(hm & below);
                      if (inv(nbm))
                        { gb += 1;
                          if (idx < cell_cap)
                            { const v2u32 c = { (u32) hbx | ((u32) ((gb ^ m) - m) << PK_HBITS), w1 };
                              DUO_EXP_PEBBLE(gcell[cbase + (u32) idx] = c;)
                            }
                          hbx = idx;
                        }
                      ncell += __popc(hm);
                      nbm &= bal(v >= __mul24(gb, TS2) + cpB);
                    }
                  ha = hax | (ga << PK_HBITS);  hb_ = hbx | (gb << PK_HBITS);
                }
            }

          /* commit the new wave (lanes outside the band get V = EDGE again behind the pruning) */
          rV = v;  rT = b;  rHA = ha;  rHB = hb_;

          /* new best / last / trim point in sweep order: the record breakers of the prefix maximum; their V is strictly
             monotone, so the LAST breaker with the wanted property is the one the serial sweep leaves behind, and the new
             best is the maximum itself */
          { const u64 rbm = candm & bal(v > e);
            R = inv(rbm) ? v : R;
            const u64 mokm = rbm & mokc;
            L = inv(mokm) ? v : L;
            const int xl = __builtin_amdgcn_ds_bpermute(top4, x);                                 /* the maximum of the half's candidates */
            besta = xl > besta ? xl : besta;
            if (mokm)
              { DUO_LC(6)
                const u64 tokm = mokm & bal(pk_trim_ok(trimtab, b));
#ifdef DAMAR_LOOPC
                if (tokm) { DUO_LC(7) }
#endif
                if (inv(tokm))
                  { const duo_v4i q = { v, K, ha, hb_ };
                    duo_tq[lane] = q;  duo_td[lane] = dif;
                  }
              }
          }

          /* sequence ends reached (the largest sweep index for A, the smallest for B) and the clipping behind them
             (align.c:628-658 / 943-975): rare; the band goes there as lane bounds and comes back as a mask */
          u64 wband = actm;                                   /* the band as widened: what the pruning starts from */
          { const u64 endm = (fastm & endc) | bytem;
            if (endm)
              { DUO_LC(8)
                const int ca2 = 2 * cx.alim - K, cb2 = 2 * cx.blim + K;
                const bool byl = inv(bytem), act = inv(actm);
                const bool isB = act && (byl ? ef == 2 : v == cb2);
                const bool isA = act && !isB && (byl ? ef == 1 : v == ca2);
                const u32 am_ = hmask(bal(isA), hb), bm_ = hmask(bal(isB), hb);
                if (am_) cold[DC_ACLIP] = 31 - __clz((int) am_);
                if (bm_) cold[DC_BCLIP] = __ffs((int) bm_) - 1;
                int more = (am_ | bm_) ? 0 : 1;
                const int va0 = cx.va0, vb0 = cx.vb0, kbase = K + s;
                int bestk = cx.bestk;
                { const u32 hr = hmask(bal(R == besta), hb);
                  if (hr) bestk = kbase - ffbl_raw(hr);
                }
                const u32 hw = hmask(actm, hb);
                int ls = ffbl_raw(hw), hs = 31 ^ ffbh_raw(hw);
                DUO_CLIP()
                wband = onm & bal(s >= ls) & bal(s <= hs);
                if (on && more == 0)
                  cx.more = 0;
                stopm |= onm & bal(more == 0);
              }
          }

          /* prune (align.c:977-986 / 1686-1695): the band becomes the span of its lanes that are within reach of the best
             point, and V = EDGE again in every lane outside it (an empty band: the find-first-bit instructions return
             -1 for 0, which no lane number reaches as an unsigned) */
          { const u64 keepm = wband & bal(rV >= besta - MAX_WAVE_LAG);
            const u32 hk = hmask(keepm, hb);
            const u32 lo_ = (u32) ffbl_raw(hk), hi_ = (u32) ffbh_raw(hk);
            spanm = bal((u32) s >= lo_) & bal((u32) s31 >= hi_);
            rV = inv(spanm) ? rV : DUO_EDGE;
            /* may every half go on?  lasta is at least the last known one */
            const u64 badm = onm & (bal((int) lo_ < 0) | bal(lastlim < besta));
            if (badm)
              { DUO_LC(9)
                int lm = L;
                for (int o = 1; o < 32; o <<= 1)
                  { const int t = __shfl_xor(lm, o);  lm = t > lm ? t : lm; }
                lm += MAX_TRIM_LAG;
                lastlim = lm > lastlim ? lm : lastlim;
                stopm |= onm & (bal((int) lo_ < 0) | bal(lastlim < besta));
              }
          }
        }
      while ((stopm | (spanm & edges)) == 0);
      if (stopm)
        break;
    }
  DUO_LC_FLUSH()
  { DuoCtx &c0 = duo_half[0];                   /* (every lane adds the same: one record counts for the wavefront) */
    const u32 lo = c0.n_cells_lo + st_cells, st_iter = (u32) (left0 - left);
    c0.n_cells_hi += (lo < st_cells) ? 1u : 0u;  c0.n_cells_lo = lo;
    c0.n_iter += st_iter;  c0.n_half += st_iter * ((u32) __popcll(onm) >> 5);
  }
  { int lm = L;                                  /* lasta and bestk out of the lanes */
    for (int o = 1; o < 32; o <<= 1)
      { const int t = __shfl_xor(lm, o);  lm = t > lm ? t : lm; }
    const u32 hr = hmask(bal(R == besta), hb), hk = hmask(spanm, hb);
    if (on)
      { duo_V[lane] = rV;  duo_HA[lane] = rHA;  duo_HB[lane] = rHB;  duo_Tlo[lane] = (u32) rT;  duo_Thi[lane] = (u32) (rT >> 32);
        cx.ls = ffbl_raw(hk);  cx.hs = 31 ^ ffbh_raw(hk);  cx.kbase = K + s;  cx.dif = dif;  cx.besta = besta;
        if (hr)
          cx.bestk = K + s - ffbl_raw(hr);
        lastlim -= MAX_TRIM_LAG;
        cx.lasta = lm > lastlim ? lm : lastlim;
        cx.ncell = ncell;
      }
  }
}

/* What the wave loop left for the halves with md == MD_RUN (the reference's loop conditions, in their order) */
__device__ __forceinline__ void duo_classify(const ReportArgs &a)
{ DUO_CX(); const int s = lane_id() & 31;
  u32 *const errw = &a.counters[3];
  if (cx.md != MD_RUN)
    return;
  if (cx.ncell > (int) a.cell_cap)
    { duo_pebbles_over(a, cx.item);
      cx.more = 0;  cx.ncell = 2;  cx.bad = 1;  cx.md = MD_END;
    }
  else if (!(cx.more && cx.lasta >= cx.besta - MAX_TRIM_LAG))
    cx.md = MD_END;
  else if (cx.hs < cx.ls)
    { if (s == 0) atomicAdd(errw + 2, 1u);
      cx.md = MD_END;
    }
  else if (cx.dif > cx.alen + cx.blen + 64)
    { if (s == 0) atomicOr(errw, DAMAR_ERR_BAND);
      cx.md = MD_END;
    }
  else if (cx.hs - cx.ls > 27)
    cx.md = MD_OVF;
}

/* A half whose band outgrew its lanes borrows the whole wavefront: the band goes to the one-alignment-per-wavefront
 * register path in the reference's own coordinates (wave_reg_cont<REV>: lane (k & 63) owns diagonal k, marks as values,
 * NA = the mark after the head's) and comes back as soon as it fits a half again (hgh - low + 3 <= PK_NARROW) or
 * finishes the direction there (through wave_mem<REV> if it outgrows the wavefront too).  Called for one half at a time
 * (hsel = its lane base) with every lane active. */
template <int REV>
DUO_PIECE void duo_solo(int job, const u32 *trimtab, SlotScratch sc, int hsel)
{ DuoCtx &cx = duo_half[hsel >> 5];              /* the record of the half that borrows the wavefront (every lane reads it) */ const ReportArgs &a = g_jobs[uni(job)];
  const int lane = lane_id();
  const int TS = a.tspace;
  const int m = REV ? -1 : 0;
  const int edge = REV ? BIG : -1;
  const int src = hsel;
  int *const cold = duo_cold + (hsel >> 1);
  WaveCtx c;
  WaveState ws;
#define DUO_PTR_OF(T, ptr) ((T) (uintptr_t) (((u64) (u32) bcast_i((int) (u32) ((u64) (uintptr_t) (ptr) >> 32), src) << 32) | \
                                             (u32) bcast_i((int) (u32) (u64) (uintptr_t) (ptr), src)))
#define DUO_SG(x) (REV ? -(x) : (x))
  c.a0 = (u32) (uni(cx.va0) - 16 * PK_PAD);  c.b0 = (u32) (uni(cx.vb0) - 16 * PK_PAD);
  c.aseq = a.ablk.bases + c.a0;  c.bseq = a.bblk.bases + c.b0;
  c.apk = a.ablk.pk;  c.bpk = a.bblk.pk;
  c.alen = uni(cx.alen);  c.blen = uni(cx.blen);
  c.ts = TS;  c.ave = a.ave_path;  c.reach = a.reach;
  c.score = a.score;  c.table = a.table;  c.trim8 = trimtab;
  const int diag = uni(cx.diag), mida = uni(cx.anti);
  { const bool selfie = (c.aseq == c.bseq);
    c.minp = (selfie && diag >= 0) ? 1 : -BIG;
    c.maxp = (selfie && diag <= 0) ? -1 : BIG;
  }
  c.aoff = 0;  c.boff = (a.comp & 1) ? (c.blen % TS) : 0;
  c.st0 = DUO_PTR_OF(DState *, sc.st0);  c.st1 = DUO_PTR_OF(DState *, sc.st1);
  c.NA = DUO_PTR_OF(int *, sc.NA);  c.NB = DUO_PTR_OF(int *, sc.NB);
  c.koff = c.blen + 8;  c.ring = a.span;
  c.cells = DUO_PTR_OF(Cell *, sc.cells);  c.cell_cap = a.cell_cap;
  c.err = &a.counters[3];
  c.atr = DUO_PTR_OF(u16 *, sc.atr);  c.btr = DUO_PTR_OF(u16 *, sc.btr);
#undef DUO_PTR_OF
  const int kbase = uni(cx.kbase), ls = uni(cx.ls), hs = uni(cx.hs);
  /* the band in the reference's coordinates: K = kbase - s, k = sigma * K */
  ws.low = REV ? ls - kbase : kbase - hs;  ws.hgh = REV ? hs - kbase : kbase - ls;
  ws.dif = uni(cx.dif);
  { const int besta = uni(cx.besta), bestk = uni(cx.bestk);
    ws.besta = DUO_SG(besta);  ws.besty = DUO_SG((besta - bestk) >> 1);
  }
  ws.lasta = DUO_SG(uni(cx.lasta));
  ws.more = uni(cx.more);  ws.reachm = uni(cold[DC_REACHM]);
  ws.aclip = REV ? -BIG : BIG;  ws.bclip = REV ? BIG : -BIG;             /* (consumed by the clipping of the last step) */
  ws.ncell = (u32) uni(cx.ncell);
  { const int ta = uni(cold[DC_TRIM]), tk = uni(cold[DC_TRIM + 1]), ra = uni(cold[DC_REACH]), rk = uni(cold[DC_REACH + 1]);
    ws.trim.a = DUO_SG(ta);  ws.trim.y = DUO_SG((ta - tk) >> 1);  ws.trim.d = uni(cold[DC_TRIM + 2]);
    ws.trim.ha = uni(cold[DC_TRIM + 3]);  ws.trim.hb = uni(cold[DC_TRIM + 4]);
    ws.reach.a = DUO_SG(ra);  ws.reach.y = DUO_SG((ra - rk) >> 1);  ws.reach.d = uni(cold[DC_REACH + 2]);
    ws.reach.ha = uni(cold[DC_REACH + 3]);  ws.reach.hb = uni(cold[DC_REACH + 4]);
  }
  ws.stopped = 0;  ws.bad = 0;  ws.narrow = 0;
  /* the half's band into the 64-lane layout (lane (k & 63) owns diagonal k) */
  LaneRegs r;
  { const int k = ws.low + ((lane - ws.low) & 63);
    const bool in = k <= ws.hgh;
    const int sl = (hsel + (in ? kbase - DUO_SG(k) : 0)) << 2;
    const int nV = duo_V[sl >> 2];
    const int nHA = duo_HA[sl >> 2], nHB = duo_HB[sl >> 2];
    r.V = in ? DUO_SG(nV) : edge;
    { const int ga = (int) ((u32) nHA >> PK_HBITS), gb = (int) ((u32) nHB >> PK_HBITS);
      const int hai = REV ? DUO_GREV - ga : ga, hbi = REV ? DUO_GREV - gb : gb;
      r.HA = (nHA & PK_HMASK) | (hai << PK_HBITS);  r.HB = (nHB & PK_HMASK) | (hbi << PK_HBITS);
      r.NA = REV ? hai - 1 : hai + 1;  r.NB = REV ? hbi - 1 : hbi + 1;
    }
    { const u32 tl = duo_Tlo[sl >> 2];
      const u32 th = duo_Thi[sl >> 2];
      r.T = ((u64) th << 32) | tl;
    }
  }
  wave_mem_sync();
  wave_reg_cont<REV>(c, mida, ws, &r);
  const bool mine = (lane & 32) == hsel;
  if (ws.narrow)
    { /* back into the half, centred */
      const int w = ws.hgh - ws.low + 1, slo = (32 - w) >> 1;
      const int nls = slo, nhs = slo + w - 1;
      const int nkbase = REV ? nls - ws.low : ws.hgh + nls;           /* K of lane nls is the highest: sigma * (REV ? low : hgh) */
      const int k = DUO_SG(nkbase - (lane & 31));
      const bool in = k >= ws.low && k <= ws.hgh;
      const int sl = (k & 63) << 2;
      const int nV = __builtin_amdgcn_ds_bpermute(sl, r.V);
      const int nHA = __builtin_amdgcn_ds_bpermute(sl, r.HA), nHB = __builtin_amdgcn_ds_bpermute(sl, r.HB);
      const u32 tl = (u32) __builtin_amdgcn_ds_bpermute(sl, (int) (u32) r.T);
      const u32 th = (u32) __builtin_amdgcn_ds_bpermute(sl, (int) (u32) (r.T >> 32));
      if (mine)
        { const int hai = (int) ((u32) nHA >> PK_HBITS), hbi = (int) ((u32) nHB >> PK_HBITS);
          duo_V[lane] = in ? DUO_SG(nV) : DUO_EDGE;
          duo_HA[lane] = (nHA & PK_HMASK) | ((REV ? DUO_GREV - hai : hai) << PK_HBITS);
          duo_HB[lane] = (nHB & PK_HMASK) | ((REV ? DUO_GREV - hbi : hbi) << PK_HBITS);
          duo_Tlo[lane] = tl;  duo_Thi[lane] = th;
          cx.mlo += nkbase - cx.kbase;  cx.mhi += nkbase - cx.kbase;
          cx.kbase = nkbase;  cx.ls = nls;  cx.hs = nhs;
          cx.md = MD_RUN;
        }
    }
  else
    { if (!ws.stopped)
        wave_mem<REV>(c, mida, ws);
      if (mine)
        { cx.md = MD_END;  cx.bad = ws.bad;
          if (ws.bad)                                /* (the excursion's own stages raised DAMAR_ERR_CELLS or DAMAR_ERR_WIDE) */
            { if (a.widemap != NULL && a.cell_cap >= a.cell_max && (lane & 31) == 0)
                { const u32 bit = 1u << (cx.item & 31);
                  if (!(atomicOr(&a.widemap[cx.item >> 5], bit) & bit))
                    atomicAdd(&a.counters[DAMAR_CNT_WIDE], 1u);
                }
            }
        }
    }
  if (mine)
    { cx.dif = ws.dif;  cx.besta = DUO_SG(ws.besta);  cx.bestk = DUO_SG(ws.besta) - 2 * DUO_SG(ws.besty);
      cx.lasta = DUO_SG(ws.lasta);  cx.more = ws.narrow ? ws.more : 0;
      cx.ncell = (int) ws.ncell;
    }
  cold[DC_REACHM] = ws.reachm;  cold[DC_ACLIP] = -1;  cold[DC_BCLIP] = 64;
  cold[DC_TRIM] = DUO_SG(ws.trim.a);  cold[DC_TRIM + 1] = DUO_SG(ws.trim.a) - 2 * DUO_SG(ws.trim.y);  cold[DC_TRIM + 2] = ws.trim.d;
  cold[DC_TRIM + 3] = ws.trim.ha;  cold[DC_TRIM + 4] = ws.trim.hb;
  cold[DC_REACH] = DUO_SG(ws.reach.a);  cold[DC_REACH + 1] = DUO_SG(ws.reach.a) - 2 * DUO_SG(ws.reach.y);  cold[DC_REACH + 2] = ws.reach.d;
  cold[DC_REACH + 3] = ws.reach.ha;  cold[DC_REACH + 4] = ws.reach.hb;
#undef DUO_SG
}

/* One chain of a pass as trace values: chain_to_trace (report.hip) with two differences that halve and halve again what a
 * wavefront waits for here (the walk is a chain of dependent loads; it was 12 % of the wavefronts' time):
 *   - the pebbles of a chain sit on CONSECUTIVE trace marks (every mark a path crosses gets one: the invariant that made
 *     NA/NB redundant), so the number of pebbles is the distance between the head's and the root's grid index and the
 *     counting walk is not needed;
 *   - `side` (0: the A chain, 1: the B chain) is a run-time value: lanes 0 and 1 of a half walk the two chains of a pass
 *     at the same time, their loads in flight together.
 * Forward: returns the number of values written to T[0 ...).  Reverse: values are prepended (T[-1], T[-2], ...), the first
 * partial segment goes into the forward trace's first pair if there is one (f0 > 0), and the number of prepended values is
 * returned (align.c:1001-1118 / 1699-1898; the `(b - a), (b - a)` quirk of align.c:1843 on the B side included). */
template <int REV>
__device__ __forceinline__ int duo_walk(const Cell *cells, int side, int head, int rootidx, int TS, int off, int mida,
                                        int ex, int ey, int ed, u16 *T, int f0, int guard, u32 *errw)
{ const int sg = side ? 1 : -1;
  const int P = side ? ey : ex, Q = side ? ex : ey;              /* coordinate tested / coordinate contributed by the end point */
  const int goff = off - PK_BIAS * TS;                           /* mark = grid index * TS + goff */
  const int k0 = (int) cells[side].w1, m0 = (int) cells[side].w0;   /* cells 0 / 1: the exact starts of the A / B chain */
  int L = 0, kc = k0, dc = 0, ac, h = head, gw = 0;
  if (head >= 2)
    { const Cell c = cells[h];
      const int hidx = (int) (c.w0 >> PK_HBITS);
      L = REV ? rootidx - hidx : hidx - rootidx;
      kc = (ex - ey) + (int) (short) (u16) ((c.w1 & 0xffffu) - (u32) (ex - ey));
      dc = ed - (int) (((u32) ed - (c.w1 >> 16)) & 0xffffu);
      ac = hidx * TS + goff + sg * kc;
      if (L < 1 || L > DAMAR_MAX_MARKS)                          /* cannot happen: a chain off the grid */
        { atomicOr(errw, DAMAR_ERR_BAND);
          atomicMax(errw + 3, 11u);
          return 0;
        }
    }
  else
    ac = REV ? m0 + sg * k0 : (mida + sg * k0) / 2;
  if (!REV)
    { int n = 2 * L;
      if (L > 0)
        { T[2 * L - 1] = 0;  T[2 * L - 2] = 0; }
      if (ac - sg * kc != P)
        { T[n] = (u16) (ed - dc);
          T[n + 1] = (u16) (Q - ac);
          n += 2;
        }
      else if (ac != Q && L > 0)
        { T[2 * L - 1] = (u16) (Q - ac);
          T[2 * L - 2] = (u16) (ed - dc);
        }
      for (int j = L; j >= 1; j--)
        { GUARD(gw, guard, 8)
          int kp, dp, ap;
          h = (int) (cells[h].w0 & PK_HMASK);
          if (j > 1)
            { const Cell c = cells[h];
              kp = kc + (int) (short) (u16) ((c.w1 & 0xffffu) - (u32) kc);
              dp = dc - (int) (((u32) dc - (c.w1 >> 16)) & 0xffffu);
              ap = (int) (c.w0 >> PK_HBITS) * TS + goff + sg * kp;
            }
          else
            { kp = k0;  dp = 0;  ap = (mida + sg * k0) / 2; }
          if (j == L)
            { T[2 * j - 2] = (u16) (T[2 * j - 2] + (dc - dp));
              T[2 * j - 1] = (u16) (T[2 * j - 1] + (ac - ap));
            }
          else
            { T[2 * j - 2] = (u16) (dc - dp);
              T[2 * j - 1] = (u16) (ac - ap);
            }
          kc = kp;  dc = dp;  ac = ap;
        }
      return n;
    }
  else
    { const int a0 = m0 + sg * k0;
      const bool partial = (m0 % TS) != off;
      const bool merged = partial && f0 > 0;
      if (partial && L == 0)
        { if (f0 == 0)
            { T[-1] = (u16) (a0 - Q);
              T[-2] = side ? (u16) (a0 - Q) : (u16) (ed - 0);
              return 2;
            }
          T[1] = (u16) (T[1] + (a0 - Q));
          T[0] = (u16) (T[0] + (ed - 0));
          return 0;
        }
      const int npush = merged ? L - 1 : L;
      int n = 2 * npush;
      if (npush > 0)
        { T[-2 * npush + 1] = 0;  T[-2 * npush] = 0; }
      if (ac - sg * kc != P)
        { T[-n - 1] = (u16) (ac - Q);
          T[-n - 2] = (u16) (ed - dc);
          n += 2;
        }
      else if (ac != Q && (f0 + n) > 0)
        { if (npush > 0)
            { T[-2 * npush + 1] = (u16) (ac - Q);
              T[-2 * npush]     = (u16) (ed - dc);
            }
          else
            { T[1] = (u16) (T[1] + (ac - Q));
              T[0] = (u16) (T[0] + (ed - dc));
            }
        }
      for (int j = L; j >= 1; j--)
        { GUARD(gw, guard, 10)
          int kp, dp, ap;
          h = (int) (cells[h].w0 & PK_HMASK);
          if (j > 1)
            { const Cell c = cells[h];
              kp = kc + (int) (short) (u16) ((c.w1 & 0xffffu) - (u32) kc);
              dp = dc - (int) (((u32) dc - (c.w1 >> 16)) & 0xffffu);
              ap = (int) (c.w0 >> PK_HBITS) * TS + goff + sg * kp;
            }
          else
            { kp = k0;  dp = 0;  ap = a0; }
          if (j == 1 && merged)
            { T[1] = (u16) (T[1] + (ap - ac));
              T[0] = (u16) (T[0] + (dc - dp));
            }
          else if (j == 1 && partial && side)
            { const u16 add0 = (L == 1) ? T[-2] : (u16) 0, add1 = (L == 1) ? T[-1] : (u16) 0;
              T[-1] = (u16) (add1 + (ap - ac));
              T[-2] = (u16) (add0 + (ap - ac));
            }
          else
            { const int jj = merged ? j - 1 : j;
              if (j == L)
                { T[-2 * jj + 1] = (u16) (T[-2 * jj + 1] + (ap - ac));
                  T[-2 * jj]     = (u16) (T[-2 * jj] + (dc - dp));
                }
              else
                { T[-2 * jj + 1] = (u16) (ap - ac);
                  T[-2 * jj]     = (u16) (dc - dp);
                }
            }
          kc = kp;  dc = dp;  ac = ap;
        }
      return n;
    }
}

/* End point and trace points of the direction that is over (align.c:1001-1118 / 1699-1898) for the halves with
 * md == MD_END: the first lane of the half walks the two pebble chains (chain_to_trace).  Leaves the pass's results in
 * cx and the half in MD_TASK with m = -1 (the reverse pass is next) or, after the reverse pass, in MD_END with m = 1
 * as the sign that the alignment is complete. */
DUO_PART void duo_finish(int job)
{ DUO_CX(); const ReportArgs &a = g_jobs[uni(job)];
  const int lane = lane_id(), hb = lane & 32, s = lane & 31;
  const SlotScratch sc = slot_scratch(a, 2 * (int) blockIdx.x + (hb >> 5));      /* (derived here: nothing of it is live across the wave loop) */
  const bool fin = cx.md == MD_END;
  const int m = cx.m;
  const int *const cold = duo_cold + (hb >> 1);
  const int TS = a.tspace;
  const int boff = (a.comp & 1) ? (cx.blen % TS) : 0;
  const int guard = 4 * (cx.alen + cx.blen) + 1024;
  u32 *const errw = &a.counters[3];
  int rx = 0, ry = 0, rd = 0, nt = 0;
  duo_trim_settle(fin);
  wave_mem_sync();
  if (fin && !cx.bad && s < 2)                      /* lane 0: the A chain, lane 1: the B chain */
    { int ta = cold[DC_TRIM], tk = cold[DC_TRIM + 1], td = cold[DC_TRIM + 2], tha = cold[DC_TRIM + 3], thb = cold[DC_TRIM + 4];
      if (cold[DC_REACHM] >= 0 && a.reach)
        { ta = cold[DC_REACH];  tk = cold[DC_REACH + 1];  td = cold[DC_REACH + 2];  tha = cold[DC_REACH + 3];  thb = cold[DC_REACH + 4]; }
      const int ty_ = (ta - tk) >> 1;
      const int trimy = (ty_ ^ m) - m, trimx = ((ta - ty_) ^ m) - m;
      const int head = s ? thb : tha, rootidx = s ? cx.rootb : cx.roota, off = s ? boff : 0;
      u16 *const T = s ? sc.btr : sc.atr;
#ifndef DAMAR_EXP_NOWALK
      if (m == 0)
        nt = duo_walk<0>(sc.cells, s, head, rootidx, TS, off, cx.anti, trimx, trimy, td, T, 0, guard, errw);
      else
        nt = duo_walk<1>(sc.cells, s, head, rootidx, TS, off, cx.anti, trimx, trimy, td, T, s ? cx.btlen : cx.atlen, guard, errw);
#else
      (void) head; (void) rootidx; (void) off; (void) T; (void) guard; (void) errw;
#endif
      rx = trimx;  ry = trimy;  rd = td;
    }
  wave_mem_sync();
  rx = hget(rx, hb, 0);  ry = hget(ry, hb, 0);  rd = hget(rd, hb, 0);
  const int at = hget(nt, hb, 0), bt = hget(nt, hb, 1);
  if (fin)
    { if (m == 0)
        { cx.aepos = rx;  cx.bepos = ry;  cx.diffs = rd;  cx.atlen = at;  cx.btlen = bt;
          cx.aback = 0;  cx.bback = 0;
          cx.m = -1;  cx.md = MD_TASK;  cx.bad = 0;
        }
      else
        { cx.abpos = rx;  cx.bbpos = ry;  cx.diffs += rd;
          cx.aback = at;  cx.bback = bt;  cx.atlen += at;  cx.btlen += bt;
          cx.m = 1;
        }
    }
}

/* One turn of the Local_Alignment machine for the two halves, as ONE call: start the passes that are due, step the
 * halves that can step until one of them has an event, and finish the passes that are over -- unless a half left the
 * lanes (MD_OVF): then the caller sends it through duo_solo first and finishes with duo_finish_piece.  (begin, loop and
 * finish were three calls; each call writes and re-reads the callee-saved registers it uses -- 4 KB per wavefront -- and
 * that was most of the kernel's write traffic, profiles/r04_sweeps.txt.) */
DUO_PIECE void duo_run(int job, const u32 *trimtab, u32 cbase)
{ DUO_CX(); const ReportArgs &a = g_jobs[uni(job)];
  if (wany(cx.md == MD_TASK))
    { duo_begin(job, cbase);
      duo_classify(a);                               /* (the seed diagonal may already have ended the pass) */
    }
  if (wany(cx.md == MD_RUN))
    {
#ifdef DAMAR_PROF
      const unsigned long long pf0 = wall_clock64();
#endif
      duo_loop(job, trimtab, cbase);                 /* every half in MD_RUN can step: the loop tests behind a step */
      duo_classify(a);
#ifdef DAMAR_PROF
      PROF_ADD(15, wall_clock64() - pf0);  PROF_ADD(29, 1);
#endif
    }
  if (wany(cx.md == MD_OVF))
    return;
  if (wany(cx.md == MD_END))
    {
#ifdef DAMAR_PROF
      const unsigned long long pf0 = wall_clock64();
#endif
      duo_finish(job);
#ifdef DAMAR_PROF
      PROF_ADD(14, wall_clock64() - pf0);
#endif
    }
}

DUO_PIECE void duo_finish_piece(int job) { duo_finish(job); }

/***** the per-half state machine of the report loop ******************************************************/

/* emit one alignment per half with `keep` (emit_record for 32 lanes): both traces to the pool (B trace reversed
 * pairwise for COMP, align.c:2033-2056) and the record */
__device__ __forceinline__ void pk_emit(const ReportArgs &a, const SlotScratch &sc, bool keep, const LaResult &r,
                                        int ar, int br, u32 item, u32 seq)
{ const int lane = lane_id(), hb = lane & 32, s = lane & 31;
  const int nval = r.atlen + r.btlen;
  u32 ri = 0, to = 0;
  if (keep && s == 0)
    { ri = atomicAdd(&a.counters[1], 1u);
      to = atomicAdd(&a.counters[2], (u32) nval);
    }
  ri = (u32) hget((int) ri, hb, 0);
  to = (u32) hget((int) to, hb, 0);
  if (keep)
    { u32 bad = 0;
      if (ri >= a.rec_cap)
        bad |= DAMAR_ERR_RECS;
      if ((u64) to + (u64) nval > (u64) a.tpool_cap || to > 0xf0000000u)       /* (the 32-bit counter must never wrap) */
        bad |= DAMAR_ERR_TPOOL;
      if (bad == 0)
        { const u16 *at = sc.atr - r.aback, *bt = sc.btr - r.bback;
          for (int i = s; i < nval; i += 32)
            { u16 v;
              if (i < r.atlen)
                v = at[i];
              else
                { int j = i - r.atlen;
                  if (a.comp)                 /* B trace pairs in reverse order, align.c:2043-2055 */
                    j = (r.btlen - 2 - 2 * (j >> 1)) + (j & 1);
                  v = bt[j];
                }
              if (a.t8)
                { ((u8 *) a.tpool)[to + i] = (u8) v;
                  if ((int) v > a.t8max)
                    atomicOr(&a.counters[3], DAMAR_ERR_T8);
                }
              else
                a.tpool[to + i] = v;
            }
        }
      if (s == 0)
        { if (bad)
            atomicOr(&a.counters[3], bad);
          else
            { LaRecord rec;
              rec.abpos = r.abpos;  rec.bbpos = r.bbpos;  rec.aepos = r.aepos;  rec.bepos = r.bepos;
              rec.diffs = r.diffs;  rec.atlen = r.atlen;  rec.btlen = r.btlen;
              rec.aread = ar;  rec.bread = br;  rec.item = item;  rec.seq = seq | ((u32) a.job << DAMAR_SEQ_BITS);  rec.toff = to;
              a.recs[ri] = rec;
            }
        }
    }
}

/* The seeds of a 32-lane group that fall into one bucket (`peers`) add to its score through the LAST of them: the group's
 * sums are formed in LDS and one lane per bucket does a plain read-modify-write.  (An atomic per seed is executed at the
 * memory side on this machine -- TCC_EA0_ATOMIC == TCC_ATOMIC -- one 32-byte transaction each, and the slot's bucket
 * arrays are private to its half anyway.)  sv = score[d] as read before the group. */
__device__ __forceinline__ void duo_bucket_add(const SlotScratch &sc, bool in, bool last, u32 peers, int d, int sv, int add, int ap)
{ const int lane = lane_id();
  __hip_atomic_store(&duo_acc[lane], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                     /* (LDS executes one wavefront's operations in order) */
  if (in)
    __hip_atomic_fetch_add(&duo_acc[(lane & 32) + (31 - __clz((int) peers))], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (last)
    { sc.score[d] = sv + __hip_atomic_load(&duo_acc[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      sc.lastp[d] = ap;
    }
}

/* pass 3: a bucket is reset by the first lane of every run of seeds in it (every store is a line written through) */
__device__ __forceinline__ void duo_bucket_reset(const SlotScratch &sc, bool in, int d)
{ const int lane = lane_id(), hb = lane & 32, s = lane & 31;
  const int dd = in ? d : BIG;
  const int dp = hget(dd, hb, (s + 31) & 31);
  if (in && (s == 0 || dp != dd))
    { sc.score[d] = 0;
      sc.lastp[d] = 0;
    }
}

enum { PK_ITEM = 0, PK_PANEL, PK_FIRE, PK_DONE };


/* one job of the launch: the two halves pull read pairs (or batch tasks) from its queue until it is empty */
/* dist != NULL: datander (scrub/tandem.c:895-1175 report_thread) -- a work item is a READ, its "seeds" are the positions
   apos of the read whose k-mer has an equal k-mer earlier in the same read, dist[...] back (damar_launch_tandem_links), and
   the alignment is the read against itself (aseq == bseq: the band may not cross the main diagonal, align.c:1949-1968) */
__device__ __forceinline__ void report2_job(const ReportArgs &a, const u32 *trimtab, const LaTask *tasks, u32 ntasks,
                                            const int *dist)
{ const int lane = lane_id(), hb = lane & 32, s = lane & 31;
  const int slot = 2 * (int) blockIdx.x + (hb >> 5);
  const SlotScratch sc = slot_scratch(a, slot);
  const u32 cbase = (u32) slot * a.cell_cap;
  const u64 *keys = a.keys;
  const u32 *vals = a.vals;
  const u64 pmask = (1ull << a.pbits) - 1;
  const int dbits = a.dbits, pshift = a.pbits + a.dbits;        /* key = pair | apos | bpos (dbits) */
  const int K = a.kmer, H = a.hitmin, W = a.binshift, minhit = a.minhit;
  const int mind = (-a.bblk.maxlen) >> W, maxd = a.ablk.maxlen >> W;
  const bool batch = tasks != NULL, tandem = dist != NULL;

  int  phase = PK_ITEM;
  u32  item = 0, seq = 0;
  u64  nidx = 0, cpair = 0, lidx = 0, end = 0, h2 = 0, fp = 0;
  int  ar = 0, br = 0, amark2 = 0, clo = BIG, chi = -BIG, sd = 0;
  int  tmb = 0, tme = 0;              /* datander: the panel of positions [tmb, tme) being scanned */
  DUO_CX();
  cx.md = MD_SCAN;  cx.m = 0;  cx.bad = 0;
  cx.va0 = cx.vb0 = 16 * PK_PAD;  cx.alen = cx.blen = 0;
  duo_V[lane] = DUO_EDGE;  duo_HA[lane] = duo_HB[lane] = 0;  duo_Tlo[lane] = duo_Thi[lane] = 0;
  duo_tq[lane].x = -BIG;
  cx.ls = cx.hs = 15;  cx.kbase = 0;  cx.dif = 0;  cx.besta = cx.bestk = cx.lasta = 0;  cx.more = 0;  cx.ncell = 2;
  cx.mlo = cx.mhi = 0;  cx.alim = cx.blim = 0;  cx.offa = cx.offb = 0;  cx.pa0 = cx.pb0 = 0;
  cx.diag = cx.anti = 0;  cx.roota = cx.rootb = 0;  cx.item = 0;
  cx.aepos = cx.bepos = cx.abpos = cx.bbpos = cx.diffs = cx.atlen = cx.btlen = cx.aback = cx.bback = 0;
  cx.n_cells_lo = cx.n_cells_hi = 0;  cx.n_iter = cx.n_half = 0;

  for (;;)
    { /* A: the halves without an alignment in hand advance their scan until they have one or have run out of work */
      while (wany(cx.md == MD_SCAN))
        { const bool sc_ = cx.md == MD_SCAN;
          if (tandem)
            { /* nidx = where code[apos] of this read sits in dist: apos = index of a k-mer's last base + 1, in [K, alen] */
              if (sc_ && phase == PK_ITEM)
                { u32 it = 0;
                  if (s == 0)
                    it = atomicAdd(a.cursor, 1u);
                  it = (u32) hget((int) it, hb, 0);
                  if (it >= a.nwork)
                    { phase = PK_DONE;  cx.md = MD_DONE; }
                  else
                    { item = it;  seq = 0;
                      ar = br = (int) it;
                      cx.va0 = cx.vb0 = (int) a.ablk.boff[ar] + 16 * PK_PAD;
                      cx.alen = cx.blen = (int) read_len(a.ablk, ar);
                      cx.item = item;
                      nidx = (u64) a.ablk.boff[ar] - (u64) ar * (u64) K - (u64) K;
                      clo = BIG;  chi = -BIG;
                      tmb = K;  tme = PANEL_SIZE;
                      if (tme >= cx.alen)
                        tme = cx.alen + 1;
                      if (a.widemap != NULL && cx.alen / a.tspace + 8 > DAMAR_MAX_MARKS)
                        { /* a read of more trace spacings than a packed chain head can name: the wide kernel's read */
                          if (s == 0)
                            atomicAdd(&a.counters[DAMAR_CNT_WIDE], 1u);
                        }
                      else
                        phase = PK_PANEL;
                    }
                }
              else if (sc_ && phase == PK_PANEL)
                { /* pass 1 (tandem.c:986-996): bucket scores of the panel */
                  for (int base = tmb; base < tme; base += 32)
                    { const int  apos = base + s;
                      const int  dg = (apos < tme) ? dist[nidx + (u64) apos] : 0;
                      const bool in = dg != 0;
                      const int  d = dg >> W;
                      int  prev = in ? sc.lastp[d] : 0;
                      const int sv = in ? sc.score[d] : 0;
                      u32  peers = hmask(wballot(in), hb);
                      { const u32 db = (u32) (d - mind);
                        for (int bit = 0; bit < a.bucket_bits; bit++)
                          { const bool one = (db >> bit) & 1;
                            const u32  mk = hmask(wballot(one), hb);
                            peers &= one ? mk : ~mk;
                          }
                      }
                      const u32  below = peers & ((1u << s) - 1u);
                      const int  pl = below ? 31 - __clz((int) below) : s;
                      const int  pap = hget(apos, hb, pl);
                      if (below) prev = pap;
                      const bool last = in && ((peers >> s) >> 1) == 0;
                      duo_bucket_add(sc, in, last, peers, d, sv, (apos - prev >= K) ? K : apos - prev, apos);
                      wave_mem_sync();
                    }
                  fp = (u64) tmb;
                  phase = PK_FIRE;
                }
              else if (sc_ && phase == PK_FIRE)
                { /* pass 2 (tandem.c:1000-1099): the next position with enough score that lies beyond lasta */
                  bool found = false;
                  int  sap = 0, sdg = 0;
                  for (int base = (int) fp; base < tme; base += 32)
                    { const int  apos = base + s;
                      const int  dg = (apos < tme) ? dist[nidx + (u64) apos] : 0;
                      const int  d = dg >> W;
                      bool fire = false;
                      if (dg != 0)
                        { const int scv = sc.score[d];
                          fire = ((scv + sc.score[d + 1] >= H) || (scv + sc.score[d - 1] >= H)) && apos > sc.lasta[d];
                        }
                      const u32 fm = hmask(wballot(fire), hb);
                      if (fm)
                        { const int l = __ffs((int) fm) - 1;
                          sap = base + l;  sdg = hget(dg, hb, l);  sd = sdg >> W;
                          fp = (u64) (base + l + 1);
                          found = true;
                          break;
                        }
                    }
                  if (found)
                    { cx.diag = sdg;  cx.anti = sap + (sap - sdg);
                      cx.m = 0;  cx.md = MD_TASK;
                      if (s == 0)
                        atomicAdd(a.nfilt, 1u);
                    }
                  else
                    { /* pass 3 (tandem.c:1103-1109), then the next panel of the read or the next read */
                      for (int base = tmb; base < tme; base += 32)
                        { const int apos = base + s;
                          const int dg = (apos < tme) ? dist[nidx + (u64) apos] : 0;
                          duo_bucket_reset(sc, dg != 0, dg >> W);
                        }
                      wave_mem_sync();
                      if (tme > cx.alen)
                        { if (clo <= chi)
                            for (int q = clo + s; q <= chi; q += 32)
                              sc.lasta[q] = 0;
                          wave_mem_sync();
                          phase = PK_ITEM;
                        }
                      else
                        { tmb = tme - PANEL_OVERLAP;
                          tme = tmb + PANEL_SIZE;
                          if (tme > cx.alen)
                            tme = cx.alen + 1;
                          phase = PK_PANEL;
                        }
                    }
                }
            }
          else if (sc_ && phase == PK_ITEM)
            { u32 it = 0;
              if (s == 0)
                it = atomicAdd(a.cursor, 1u);
              it = (u32) hget((int) it, hb, 0);
              if (it >= (batch ? ntasks : a.nwork))
                { phase = PK_DONE;  cx.md = MD_DONE; }
              else if (batch)
                { const LaTask tk = tasks[it];
                  item = it;  seq = 0;
                  ar = tk.aread;  br = tk.bread;
                  cx.va0 = (int) a.ablk.boff[ar] + 16 * PK_PAD;  cx.vb0 = (int) a.bblk.boff[br] + 16 * PK_PAD;
                  cx.alen = (int) read_len(a.ablk, ar);  cx.blen = (int) read_len(a.bblk, br);
                  cx.diag = tk.diag;  cx.anti = tk.anti;
                  cx.item = item;
                  if (a.widemap != NULL && (cx.alen > cx.blen ? cx.alen : cx.blen) / a.tspace + 8 > DAMAR_MAX_MARKS)
                    { if (s == 0)                       /* (the wide kernel's task: see the pair branch below) */
                        atomicAdd(&a.counters[DAMAR_CNT_WIDE], 1u);
                    }
                  else
                    { cx.m = 0;  cx.md = MD_TASK; }
                }
              else
                { item = a.order ? a.order[it] : it;
                  nidx = a.work[item];
                  cpair = keys[nidx] >> pshift;
                  ar = (int) (cpair & ((1ull << a.abits) - 1));  br = (int) (cpair >> a.abits);
                  cx.va0 = (int) a.ablk.boff[ar] + 16 * PK_PAD;  cx.vb0 = (int) a.bblk.boff[br] + 16 * PK_PAD;
                  cx.alen = (int) read_len(a.ablk, ar);  cx.blen = (int) read_len(a.bblk, br);
                  cx.item = item;
                  seq = 0;  amark2 = 0;  clo = BIG;  chi = -BIG;
                  if (a.widemap != NULL && (cx.alen > cx.blen ? cx.alen : cx.blen) / a.tspace + 8 > DAMAR_MAX_MARKS)
                    { /* a read of more trace spacings than a packed chain head can name: the wide kernel's pair */
                      if (!(cx.alen < a.hgap_min && cx.blen < a.hgap_min) && s == 0)
                        atomicAdd(&a.counters[DAMAR_CNT_WIDE], 1u);
                    }
                  else if (!(cx.alen < a.hgap_min && cx.blen < a.hgap_min))
                    phase = PK_PANEL;
                }
            }
          else if (sc_ && phase == PK_PANEL)
            { if (!(nidx < a.nhits && (keys[nidx] >> pshift) == cpair))
                { /* the pair is done: filter.c:2417-2432 leaves lasta all zero again */
                  if (clo <= chi)
                    for (int q = clo + s; q <= chi; q += 32)
                      sc.lasta[q] = 0;
                  phase = PK_ITEM;
                }
              else
                { /* one A-panel (filter.c:2251-2266): hits while the pair continues and the hit just consumed has apos <= amark */
                  const int amark = amark2 + PANEL_SIZE;
                  amark2 = amark - PANEL_OVERLAP;
                  lidx = nidx;  end = lidx;  h2 = lidx;
                  for (u64 base = lidx; ; base += 32)
                    { const u64  f = base + s;
                      const bool in = f < a.nhits && (keys[f] >> pshift) == cpair;
                      const int  ap = in ? (int) ((keys[f] >> dbits) & pmask) : 0;
                      const bool nextsame = (f + 1 < a.nhits) && ((keys[f + 1] >> pshift) == cpair);
                      const bool stop = in && !(nextsame && ap <= amark);
                      u32 le = hmask(wballot(in && ap <= amark2), hb);
                      const u32 sm = hmask(wballot(stop), hb);
                      if (sm)
                        { const int l = __ffs((int) sm) - 1;
                          end = base + l + 1;
                          le &= (l == 31) ? ~0u : ((1u << (l + 1)) - 1);
                          if (le) h2 = base + (31 - __clz((int) le)) + 1;
                          break;
                        }
                      if (le) h2 = base + (31 - __clz((int) le)) + 1;
                      if (hmask(wballot(in), hb) == 0)          /* cannot happen: a run always ends with a stop */
                        { end = base; break; }
                    }
                  nidx = end;
                  if (end - lidx >= (u64) minhit)
                    { /* pass 1: bucket scores (filter.c:2268-2277) */
                      for (u64 base = lidx; base < end; base += 32)
                        { const u64  f = base + s;
                          const bool in = f < end;
                          const int  ap = in ? (int) ((keys[f] >> dbits) & pmask) : 0;
                          const int  d  = in ? (seed_diag(keys[f], vals, f, pmask, dbits) >> W) : BIG;
                          int  prev = in ? sc.lastp[d] : 0;
                          const int sv = in ? sc.score[d] : 0;
                          u32  peers = hmask(wballot(in), hb);
                          { const u32 db = (u32) (d - mind);
                            for (int bit = 0; bit < a.bucket_bits; bit++)
                              { const bool one = (db >> bit) & 1;
                                const u32  mk = hmask(wballot(one), hb);
                                peers &= one ? mk : ~mk;
                              }
                          }
                          const u32  below = peers & ((1u << s) - 1u);
                          const int  pl = below ? 31 - __clz((int) below) : s;
                          const int  pap = hget(ap, hb, pl);
                          if (below) prev = pap;
                          const bool last = in && ((peers >> s) >> 1) == 0;
                          duo_bucket_add(sc, in, last, peers, d, sv, (ap - prev >= K) ? K : ap - prev, ap);
                          wave_mem_sync();
                        }
                      fp = lidx;
                      phase = PK_FIRE;
                    }
                  else
                    nidx = h2;
                }
            }
          else if (sc_ && phase == PK_FIRE)
            { /* pass 2 (filter.c:2283-2405): the next seed in order with enough score whose apos is beyond lasta */
              bool found = false;
              int  sap = 0, sdg = 0;
              for (u64 base = fp; base < end; base += 32)
                { const u64  f = base + s;
                  const bool in = f < end;
                  const int  ap = in ? (int) ((keys[f] >> dbits) & pmask) : 0;
                  const int  dg = in ? seed_diag(keys[f], vals, f, pmask, dbits) : 0;
                  const int  d  = dg >> W;
                  bool fire = false;
                  if (in)
                    { const int scv = sc.score[d];
                      fire = ((scv + sc.score[d + 1] >= H) || (scv + sc.score[d - 1] >= H)) && ap > sc.lasta[d];
                    }
                  const u32 fm = hmask(wballot(fire), hb);
                  if (fm)
                    { const int l = __ffs((int) fm) - 1;
                      sap = hget(ap, hb, l);  sdg = hget(dg, hb, l);  sd = sdg >> W;
                      fp = base + l + 1;
                      found = true;
                      break;
                    }
                }
              if (found)
                { cx.diag = sdg;  cx.anti = sap + (sap - sdg);
#ifndef DAMAR_EXP_SCANONLY
                  cx.m = 0;  cx.md = MD_TASK;
#endif
                  if (s == 0)
                    atomicAdd(a.nfilt, 1u);
                }
              else
                { /* pass 3: reset the touched buckets (filter.c:2407-2411) */
                  for (u64 base = lidx; base < end; base += 32)
                    { const u64 f = base + s;
                      const bool in = f < end;
                      duo_bucket_reset(sc, in, in ? (seed_diag(keys[f], vals, f, pmask, dbits) >> W) : 0);
                    }
                  wave_mem_sync();
                  nidx = h2;
                  phase = PK_PANEL;
                }
            }
        }

      /* B: Local_Alignment (align.c:1904-2097 for low == hgh == diag), one pass at a time per half */
      if (wany(cx.md == MD_TASK || cx.md == MD_RUN))
        duo_run(a.job, trimtab, cbase);
      { const u64 ov = wballot(cx.md == MD_OVF);
        if (ov)
          {
#ifdef DAMAR_PROF
            const unsigned long long pf0 = wall_clock64();
            PROF_ADD(28, __popcll(ov) >> 5);
#endif
            duo_trim_settle(cx.md == MD_OVF);           /* the excursion carries the trim point as a record */
            for (int h = 0; h < 64; h += 32)
              if ((ov >> h) & 1)
                { if (uni(duo_half[h >> 5].m))
                    duo_solo<1>(a.job, trimtab, sc, h);
                  else
                    duo_solo<0>(a.job, trimtab, sc, h);
                }
            duo_classify(a);
#ifdef DAMAR_PROF
            PROF_ADD(13, wall_clock64() - pf0);
#endif
            if (wany(cx.md == MD_END))               /* (duo_run left them for behind the excursion) */
              duo_finish_piece(a.job);
          }
      }

      /* C: what the reference does with the path (filter.c:2318-2380), for the halves whose reverse pass is over */
      if (wany(cx.md == MD_END))
        { const bool task = cx.md == MD_END;
          LaResult r;
          r.abpos = cx.abpos;  r.bbpos = cx.bbpos;  r.aepos = cx.aepos;  r.bepos = cx.bepos;  r.diffs = cx.diffs;
          r.atlen = cx.atlen;  r.btlen = cx.btlen;  r.aback = cx.aback;  r.bback = cx.bback;
          if (batch)
            { pk_emit(a, sc, task, r, ar, br, item, 0);
              if (task)
                { phase = PK_ITEM;  cx.md = MD_SCAN; }
            }
          else
            { int lo = 0, hi = 0;
              if (task && s == 0)                         /* Diagonal_Span (filter.c:2079-2110) on the A-view path */
                { const u16 *pt = sc.atr - r.aback;
                  int dd, tlen = r.atlen - 2;
                  lo = hi = r.abpos - r.bbpos;
                  dd = r.aepos - r.bepos;
                  if (dd < lo) lo = dd; else if (dd > hi) hi = dd;
                  dd = (r.abpos / a.tspace) * a.tspace - r.bbpos;
                  for (int i = 1; i < tlen; i += 2)
                    { dd += a.tspace - pt[i];
                      if (dd < lo) lo = dd; else if (dd > hi) hi = dd;
                    }
                  lo = (lo >> W) - 1;
                  hi = (hi >> W) + 1;
                }
              lo = hget(lo, hb, 0);  hi = hget(hi, hb, 0);
              if (task)
                { if (sd < lo) lo = sd; else if (sd > hi) hi = sd;
                  if (lo < mind - 1) lo = mind - 1;
                  if (hi > maxd + 1) hi = maxd + 1;
                  for (int q = lo + s; q <= hi; q += 32)
                    if (r.aepos > sc.lasta[q])
                      sc.lasta[q] = r.aepos;
                  if (lo < clo) clo = lo;
                  if (hi > chi) chi = hi;
                }
              wave_mem_sync();
              const bool keep = task && (r.aepos - r.abpos) + (r.bepos - r.bbpos) >= a.minover;
              pk_emit(a, sc, keep, r, ar, br, item, seq);
              if (keep)
                seq += 1;
              if (task)
                cx.md = MD_SCAN;                          /* (phase is still PK_FIRE: the panel's next seed) */
            }
        }
      if (!wany(cx.md != MD_DONE))
        break;
    }
  if (lane == 0)
    { const DuoCtx &c0 = duo_half[0];
      atomicAdd((unsigned long long *) &a.counters[DAMAR_CNT_CELLS], ((unsigned long long) c0.n_cells_hi << 32) | c0.n_cells_lo);
      atomicAdd((unsigned long long *) &a.counters[DAMAR_CNT_HALFSTEPS], (unsigned long long) c0.n_half);
      atomicAdd((unsigned long long *) &a.counters[DAMAR_CNT_ITERS], (unsigned long long) c0.n_iter);
    }
}

__global__ __launch_bounds__(64, DUO_WAVES)
void report2_kernel(int njobs, const LaTask *tasks, u32 ntasks, const int *dist)
{ __shared__ u32 trimtab[256];
  pk_fill_trimtab(trimtab, g_jobs[0].mscore, g_jobs[0].dscore);
  __syncthreads();
#ifdef DAMAR_PROF
  struct PfExit { unsigned long long t0; __device__ ~PfExit() { unsigned long long d = wall_clock64() - t0;
    if (lane_id() == 0) { atomicAdd(&g_prof[23], d); atomicAdd(&g_prof[25], 1ull); atomicMax(&g_prof[24], d); } } } pf_exit = { (unsigned long long) wall_clock64() };
#endif
  for (int turn = 0; turn < njobs; turn++)
    report2_job(g_jobs[((int) blockIdx.x + turn) % njobs], trimtab, tasks, ntasks, dist);
}

void damar_launch_report2(const ReportArgs *jobs, int njobs, const LaTask *tasks, u32 ntasks, int nslots, hipStream_t st)
{ jobs_upload(jobs, njobs, st);
  hipLaunchKernelGGL(report2_kernel, dim3(nslots / 2), dim3(64), 0, st, njobs, tasks, ntasks, (const int *) NULL);
}

/* datander: one work item per read of a->ablk, dist as produced by damar_launch_tandem_links */
void damar_launch_tandem_report2(const ReportArgs *a, const int *dist, int nslots, hipStream_t st)
{ if (a->nwork == 0)
    return;
  jobs_upload(a, 1, st);
  hipLaunchKernelGGL(report2_kernel, dim3(nslots / 2), dim3(64), 0, st, 1, (const LaTask *) NULL, 0u, dist);
}

/* report_packed.h -- the report loop with TWO read pairs per wavefront (included by report.hip).
 *
 * Why: the live band of a Local_Alignment wave is 12 diagonals wide on average and 99.96 % of the wave
 * steps fit 32 lanes (profiles/: band histogram of the oracle), so one alignment per 64-lane wavefront
 * leaves five lanes in six idle -- and round 1's kernel kept every per-alignment scalar (band bounds, best
 * and trim points, loop control) in SGPRs: 139 scalar instructions per wave step on the ONE scalar unit the
 * four SIMDs of a CU share (76 % of its calibrated issue rate, tools/roofcal.hip), 51 % on the vector side.
 * Here each 32-lane half of a wavefront owns a read pair: everything that is uniform per alignment lives in
 * VGPRs (the same value in the 32 lanes of a half), computed by vector instructions whose cost does not
 * depend on the number of alignments they serve, and the scalar unit only steers the loops.
 *
 * Reference semantics are those of report.hip (same citations: dalign/filter.c:2128-2432 report_thread,
 * dalign/align.c:409-1122 forward_wave, :1126-1898 reverse_wave, :1904-2097 Local_Alignment); what differs is
 * the mapping onto the machine:
 *   - lane s of a half owns diagonal k = kbase + s (reverse) or kbase - s (forward): the reference's sweep
 *     order (align.c:781 hgh..low, :1490 low..hgh) is ascending lane order in both directions, no ring
 *     wrap; the band is re-centred in the half (7 lane shuffles) when it drifts to an edge;
 *   - the half is a state machine (scan the pair's seeds -> Local_Alignment task -> lasta update -> ...),
 *     so that the two halves run their wave steps in lockstep whatever their seeds look like;
 *   - the popcount M of the match history is not carried: M == popcount(T & (2^61 - 1)) at all times
 *     (align.c:827-829, 853-855 keep exactly that invariant), taken where it is needed;
 *   - the mark of the pebble at a chain head rides in the top 12 bits of the head index (as a trace-grid
 *     index), NA/NB are grid indexes too: no cell is read back inside the wave loop;
 *   - new best / last / trim point (align.c:911-928) by a prefix maximum in sweep order instead of a serial
 *     replay; TABLE/SCORE (2 x 64 KB in HBM) are replaced by one 1 KB table in LDS: the test
 *     "TABLE[lo15] >= 0 && TABLE[hi15] + SCORE[lo15] >= 0" says that every suffix of the last 30 columns
 *     scores >= 0, and the minimum suffix score of 30 columns composes from 8-bit chunks.
 * A band that needs more than the 32 lanes leaves the packed loop: its state goes to the slot's memory
 * buffers and the full-wave memory path (wave_mem) finishes that direction.
 */

#ifndef PK_WINDOWS
#define PK_WINDOWS 0                      /* 1: the snake reads per-lane sliding windows of the packed bases (8 more VGPRs; measured: no gain) */
#endif
__device__ __forceinline__ u32 hmask(u64 m, int hb) { return (u32) (m >> hb); }           /* this half's 32 bits */
__device__ __forceinline__ int hget(int v, int hb, int s) { return __builtin_amdgcn_ds_bpermute((hb + s) << 2, v); }
__device__ __forceinline__ int upd_dpp_shr(int old, int v, int n)      /* lane i <- lane i-n within a row of 16 */
{ switch (n)
    { case 1:  return __builtin_amdgcn_update_dpp(old, v, 0x111, 0xf, 0xf, false);
      case 2:  return __builtin_amdgcn_update_dpp(old, v, 0x112, 0xf, 0xf, false);
      case 4:  return __builtin_amdgcn_update_dpp(old, v, 0x114, 0xf, 0xf, false);
      default: return __builtin_amdgcn_update_dpp(old, v, 0x118, 0xf, 0xf, false);
    }
}

/* Inclusive prefix maximum (minimum for REV) in lane order inside each 32-lane half: Kogge-Stone inside the rows of
   16 by DPP row_shr, then the last lane of rows 0 / 2 into rows 1 / 3.  In place: a lane whose DPP source does not
   exist is simply not written (bound_ctrl off), which is the identity here -- one instruction per step instead of
   the mov-identity / mov_dpp / max triple the update_dpp builtin compiles to.  (s_nop 1: a DPP operand needs two wait
   states after the VALU write of its register, and the assembler does not add them inside inline asm.) */
template <int REV>
__device__ __forceinline__ int pk_prefix_best(int x)
{ if (!REV)
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1" : "+v"(x));
  else
    asm("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1" : "+v"(x));
  return x;
}

/* per-half pair context and direction bookkeeping: every field holds the same value in the 32 lanes of a half */
struct PkPair
{ int  a0, b0;              /* offsets of the two reads in the blocks' base arrays */
  int  alen, blen;
  int  minp, maxp, boff;    /* (aoff is 0 on this path: filter.c:2318 aligns A reads forward) */
};

/* What a direction touches only at events (a new trim point, a sequence end reached, clipping) lives in LDS, one
   16-word record per half, not in registers: 13 VGPRs less across the wave loop.  A workgroup is one wavefront and
   every lane of a half executes the same stores with the same values, so every lane reads back what it wrote itself. */
__shared__ int pk_cold[2 * 16];
enum { PKC_REACHM = 0, PKC_ACLIP, PKC_BCLIP, PKC_TRIM, PKC_REACH = PKC_TRIM + 5 };
struct LdsInt
{ int at;
  __device__ __forceinline__ operator int() const { return pk_cold[at]; }
  __device__ __forceinline__ LdsInt &operator=(int v) { pk_cold[at] = v; return *this; }
  __device__ __forceinline__ LdsInt &operator=(const LdsInt &o) { pk_cold[at] = (int) o; return *this; }
};
struct LdsTip { LdsInt a, y, d, ha, hb; };
__device__ __forceinline__ LdsTip pk_cold_tip(int at) { LdsTip t = { { at }, { at + 1 }, { at + 2 }, { at + 3 }, { at + 4 } }; return t; }

struct PkDir
{ int low, hgh, dif, besta, besty, lasta, more, kbase;
  int ncell;
  int ovf;                  /* the band outgrew the half: continue on the full-wave path */
  int bad;
  int fin;                  /* the direction is over for this half (ended, failed, or finished on the full-wave path) */
};

/* the names the wave code uses for the half's bookkeeping (fields of D, kept in registers) and constants */
#define PK_NAMES()                                                                                   \
  const int lane = lane_id(), hb = lane & 32, s = lane & 31;                                         \
  const int KS = REV ? 1 : -1, S = REV ? -1 : 1;                                                     \
  const int edge = REV ? BIG : -1;                                                                   \
  const int TS = uni(a.tspace), ave = uni(a.ave_path);                                               \
  const u32 *apk = uni_ptr(a.ablk.pk), *bpk = uni_ptr(a.bblk.pk);                                    \
  const u8 *abase = uni_ptr(a.ablk.bases), *bbase = uni_ptr(a.bblk.bases);                           \
  GLOBAL_AS v2u32 *const gcell = (GLOBAL_AS v2u32 *) uni_ptr((Cell *) a.cells);                          \
  const int cell_cap = (int) uni((int) a.cell_cap);                                                  \
  u32 *const errw = uni_ptr(&a.counters[3]);                                                         \
  const int offa = -PK_BIAS * TS, offb = p.boff - PK_BIAS * TS;          /* mark = index * TS + off */ \
  int &low = D.low, &hgh = D.hgh, &dif = D.dif, &besta = D.besta, &besty = D.besty, &lasta = D.lasta; \
  int &more = D.more, &kbase = D.kbase;                                                              \
  int &ncell = D.ncell, &ovf = D.ovf, &bad = D.bad;                                                  \
  const int cold0 = hb >> 1;                               /* this half's record in pk_cold */         \
  LdsInt reachm = { cold0 + PKC_REACHM }, aclip = { cold0 + PKC_ACLIP }, bclip = { cold0 + PKC_BCLIP }; \
  LdsTip trim = pk_cold_tip(cold0 + PKC_TRIM), reach = pk_cold_tip(cold0 + PKC_REACH);               \
  (void) lane; (void) hb; (void) s; (void) KS; (void) S; (void) edge; (void) TS; (void) ave; (void) apk; (void) bpk; \
  (void) gcell; (void) cell_cap; (void) errw; (void) offa; (void) offb; (void) abase; (void) bbase;
/* derived on use, so that they do not sit in registers across the wave loop */
#define aseq      (abase + p.a0 + (REV ? -1 : 0))
#define bseq      (bbase + p.b0 + (REV ? -1 : 0))
#define va0       (p.a0 + 16 * PK_PAD)
#define vb0       (p.b0 + 16 * PK_PAD)
#define valen     (p.alen)
#define vblen     (p.blen)
#define steplimit (p.alen + p.blen + 64)
#define guard     (4 * (p.alen + p.blen) + 1024)

/* Wave 0 on the seed diagonal (align.c:491-626 / 1203-1340) for the halves with `on`: every lane of the half
 * computes the same values.  Sets up D and the lane registers of the direction. */
template <int REV>
__device__ __forceinline__ void pk_init(const ReportArgs &a, bool on, const PkPair &p, u32 cbase, int diag, int mida, PkDir &D,
                                        int &rV, u64 &rT, int &rHA, int &rHB, int &rNA, int &rNB)
{ PK_NAMES()
  low = diag;  hgh = diag;  dif = 0;
  besta = mida;  lasta = mida;  besty = (mida - diag) >> 1;  more = 1;  reachm = -1;
  aclip = REV ? -BIG : BIG;  bclip = REV ? BIG : -BIG;
  ncell = 2;
  kbase = diag - KS * 15;
  trim.a = mida;  reach.a = mida;  trim.y = besty;  reach.y = besty;  trim.d = 0;  reach.d = 0;
  trim.ha = 0;  reach.ha = 0;  trim.hb = 1;  reach.hb = 1;
  ovf = 0;  bad = 0;
  D.fin = 0;

  rV = edge;  rT = 0;  rHA = 0;  rHB = 0;  rNA = 0;  rNB = 0;

  if (on)
    { const int k = diag;
      int y = (mida - k) >> 1, nai, nbi, hai, hbi, ha = 0, hb_ = 1, v;
      int qa, qb;
      if (!REV)
        { qa = ((y + k) + TS) / TS;  qb = (y + (TS - p.boff)) / TS; }
      else
        { qa = ((y + k) + TS - 1) / TS;  qb = (y + (TS - p.boff) - 1) / TS; }
      nai = qa - 1 + PK_BIAS;  nbi = qb - 1 + PK_BIAS;
      hai = REV ? nai + 1 : nai;  hbi = REV ? nbi + 1 : nbi;       /* reverse: the true start, rounded up to the grid */
      if (s == 0)
        { gcell[cbase] = cell_root(REV ? y + k : nai * TS + offa, k);
          gcell[cbase + 1] = cell_root(REV ? y : nbi * TS + offb, k);
        }
      if (!REV) { nai += 1;  nbi += 1; }
      { const SnakeOut so = SNAKE_AT(k, y, 0, 0ull);
        y = so.y;
        if (so.nb == 0)      { more = 0; bclip = k; }
        else if (so.na == 0) { more = 0; aclip = k; }
      }
      v = (y << 1) + k;
      int g0 = 0;
      for (;;)
        { const int na = nai * TS + offa;
          if (!(REV ? (y + k <= na) : (y + k >= na)))
            break;
          GUARD(g0, guard, 2)
          if (s == 0 && ncell < cell_cap)
            gcell[cbase + (u32) ncell] = cell_pack(ha, k, 0, nai);
          ha = ncell++;  hai = nai;  nai += S;
        }
      for (;;)
        { const int nb = nbi * TS + offb;
          if (!(REV ? (y <= nb) : (y >= nb)))
            break;
          GUARD(g0, guard, 3)
          if (s == 0 && ncell < cell_cap)
            gcell[cbase + (u32) ncell] = cell_pack(hb_, k, 0, nbi);
          hb_ = ncell++;  hbi = nbi;  nbi += S;
        }
      if (REV ? (v < besta) : (v > besta))
        { besta = lasta = v;  trim.a = v;
          besty = y;  trim.y = y;
          trim.ha = ha;  trim.hb = hb_;
        }
      if (s == 15)
        { rV = v;  rT = HIST_FULL;
          rHA = ha | (hai << PK_HBITS);  rHB = hb_ | (hbi << PK_HBITS);
        }
      rNA = nai;  rNB = nbi;
      if (ncell > cell_cap)              /* a seed diagonal that slides over more marks than the pool holds */
        { if (s == 0) atomicOr(errw, DAMAR_ERR_CELLS);
          more = 0;  ncell = 2;  bad = 1;  D.fin = 1;
        }
    }
}

/* The wave steps of one direction for the two halves of the wavefront (align.c:667-999 / 1378-1697), until
 * every half has finished the direction or outgrown its 32 lanes (D.ovf). */
template <int REV>
__device__ __forceinline__ void pk_loop(const ReportArgs &a, const u32 *trimtab, bool on, bool first, const PkPair &p, u32 cbase, PkDir &D,
                                        int &rV, u64 &rT, int &rHA, int &rHB, int &rNA, int &rNB)
{ PK_NAMES()
  on = on && !D.fin && !ovf;
#if PK_WINDOWS
  typedef u32 v2u __attribute__((ext_vector_type(2)));
  u64 wina = 0, winb = 0;                   /* bases [16 wd, 16 wd + 32) of the two reads, 2 bits each (biased dword index wd) */
  u32 nxta = 0, nxtb = 0;                   /* the 16 bases that follow in the direction of the wave */
  int wda = -4096, wdb = -4096;
#endif

  /* clipping at sequence ends (align.c:628-658 / 943-975), per half */
#define PK_CLIP()                                                                                      \
  if (wany(on && more == 0))                                                                           \
    { const bool cl_ = on && more == 0;                                                                \
      int m_ = pk_popc61(rT);                                                                          \
      if (cl_)                                                                                         \
        { if (bseq[besty] != 4 && aseq[besta - besty] != 4)                                            \
            more = 1;                                                                                  \
        }                                                                                              \
      const int acl_ = aclip, bcl_ = bclip;                                                            \
      { const bool ca_ = cl_ && (REV ? (low <= acl_) : (hgh >= acl_));                                 \
        const int  sl_ = ca_ ? KS * (acl_ - kbase) : 0;                                                \
        const int  mm_ = hget(m_, hb, sl_), vv_ = hget(rV, hb, sl_);                                   \
        const int  ha_ = hget(rHA, hb, sl_), hb2_ = hget(rHB, hb, sl_);                                \
        if (ca_)                                                                                       \
          { if (REV) low = acl_ + 1; else hgh = acl_ - 1;                                              \
            if (reachm <= mm_)                                                                         \
              { reachm = mm_; reach.a = vv_; reach.y = (vv_ - acl_) / 2; reach.d = dif;                \
                reach.ha = ha_ & PK_HMASK; reach.hb = hb2_ & PK_HMASK; }                               \
          }                                                                                            \
      }                                                                                                \
      { const bool cb_ = cl_ && (REV ? (hgh >= bcl_) : (low <= bcl_));                                 \
        const int  sl_ = cb_ ? KS * (bcl_ - kbase) : 0;                                                \
        const int  mm_ = hget(m_, hb, sl_), vv_ = hget(rV, hb, sl_);                                   \
        const int  ha_ = hget(rHA, hb, sl_), hb2_ = hget(rHB, hb, sl_);                                \
        if (cb_)                                                                                       \
          { if (REV) hgh = bcl_ - 1; else low = bcl_ + 1;                                              \
            if (reachm <= mm_)                                                                         \
              { reachm = mm_; reach.a = vv_; reach.y = (vv_ - bcl_) / 2; reach.d = dif;                \
                reach.ha = ha_ & PK_HMASK; reach.hb = hb2_ & PK_HMASK; }                               \
          }                                                                                            \
      }                                                                                                \
      if (cl_)                                                                                         \
        { aclip = REV ? -BIG : BIG;                                                                    \
          bclip = REV ? BIG : -BIG;                                                                    \
        }                                                                                              \
    }

  if (first)                 /* (a re-entry after the other half's excursion resumes behind the clipping of its last step) */
    { PK_CLIP() }

#ifdef DAMAR_PROF
  unsigned long long pf_iters = 0, pf_half = 0;
#endif
  for (;;)
    { if (on && !(more && (REV ? (lasta <= besta + MAX_TRIM_LAG) : (lasta >= besta - MAX_TRIM_LAG))))
        { D.fin = 1;  on = false; }           /* this direction is over for the half: never stepped again */
      if (on && hgh < low)
        { if (s == 0) atomicAdd(errw + 2, 1u);  D.fin = 1;  on = false; }
      if (on && dif > steplimit)
        { if (s == 0) atomicOr(errw, DAMAR_ERR_BAND);  D.fin = 1;  on = false; }
      if (on && hgh - low + 3 > 32)                 /* would not fit the half: continue on the full-wave path */
        { ovf = 1;  on = false; }
      if (!wany(on))
        break;
#ifdef DAMAR_PROF
      pf_iters += 1;  pf_half += (unsigned long long) __popcll(wballot(on)) >> 5;
#endif

      /* keep the band (plus the two lanes it may grow by) inside the half */
      { const int slo = REV ? low - kbase : kbase - hgh, shi = REV ? hgh - kbase : kbase - low;
        const bool mv = on && (slo < 1 || shi > 30);
        if (wany(mv))
          { const int dl = mv ? ((32 - (shi - slo + 1)) >> 1) - slo : 0;
            const int src = (hb + ((s - dl) & 31)) << 2;
            rV  = __builtin_amdgcn_ds_bpermute(src, rV);
            rHA = __builtin_amdgcn_ds_bpermute(src, rHA);
            rHB = __builtin_amdgcn_ds_bpermute(src, rHB);
            rNA = __builtin_amdgcn_ds_bpermute(src, rNA);
            rNB = __builtin_amdgcn_ds_bpermute(src, rNB);
            { const u32 tl = (u32) __builtin_amdgcn_ds_bpermute(src, (int) (u32) rT);
              const u32 th = (u32) __builtin_amdgcn_ds_bpermute(src, (int) (u32) (rT >> 32));
              rT = ((u64) th << 32) | tl;
            }
            kbase -= KS * dl;
          }
      }

      const int k = kbase + KS * s;
      bool act;
      int  v, y = 0, ha, hb_, nai, nbi;
      u64  b;
      int  ena = 1, enb = 1;

      /* widen (align.c:675-776 / 1386-1486) and pick the predecessor (align.c:793-825 / 1502-1534).  Computed by every
         lane, also those of a half that is not stepping (their values go nowhere: act is false, the bookkeeping is
         selected by `on`): no divergent region, no defaults to set up for it */
      { const int upV = lane_up(rV), dnV = lane_dn(rV);
        const int upNA = lane_up(rNA), dnNA = lane_dn(rNA), upNB = lane_up(rNB), dnNB = lane_dn(rNB);
        int nlow = low - 1, nhgh = hgh + 1;
        if (nlow < p.minp) nlow += 1;
        if (nhgh > p.maxp) nhgh -= 1;
        const bool newlo = on && (nlow < low) && k == nlow, newhi = on && (nhgh > hgh) && k == nhgh;
        /* the value of diagonal k+1 sits in lane s+KS, that of k-1 in lane s-KS */
        const int kpNA = REV ? upNA : dnNA, kmNA = REV ? dnNA : upNA;
        const int kpNB = REV ? upNB : dnNB, kmNB = REV ? dnNB : upNB;
        if (newlo || newhi) rV = edge;
        rNA = newlo ? kpNA : (newhi ? kmNA : rNA);
        rNB = newlo ? kpNB : (newhi ? kmNB : rNB);
        nai = rNA;  nbi = rNB;
        /* (the neighbours' V was fetched before the new edge lanes were set: an edge lane's own old V is never
           a neighbour of an active diagonal's predecessor choice except as `edge`, enforced below) */
        act = on && k >= nlow && k <= nhgh;
        int am = REV ? dnV : upV, ap = REV ? upV : dnV;              /* V[k-1], V[k+1] of the previous wave */
        if (k - 1 < low || k - 1 > hgh) am = edge;                   /* outside the previous band */
        if (k + 1 > hgh || k + 1 < low) ap = edge;
        const int ac = (k < low || k > hgh) ? edge : rV;
        if (on)
          { low = nlow;  hgh = nhgh;  dif += 1; }
        int  nbv;
        bool take, upk;                                              /* predecessor = a neighbour? diagonal k+1? */
        if (!REV)
          { nbv = am > ap ? am : ap;  take = ac < nbv;  upk = am < ap;
            v = take ? nbv + 1 : ac + 2;
          }
        else
          { nbv = am < ap ? am : ap;  take = ac > nbv;  upk = !(ap > am);
            v = take ? nbv - 1 : ac - 2;
          }
        /* lane of the predecessor: k+1 -> s+KS, k-1 -> s-KS */
        const int ds = take ? (upk ? KS : -KS) : 0;
        const int src = (lane + ds) << 2;
        ha  = __builtin_amdgcn_ds_bpermute(src, rHA);
        hb_ = __builtin_amdgcn_ds_bpermute(src, rHB);
        const u32 tlo = (u32) __builtin_amdgcn_ds_bpermute(src, (int) (u32) rT);
        const u32 thi = (u32) __builtin_amdgcn_ds_bpermute(src, (int) (u32) (rT >> 32));
        b = ((u64) thi << 32) | tlo;
      }

      if (act)
        { b <<= 1;
          y = (v - k) >> 1;
#if !PK_WINDOWS
          { const SnakeOut so = SNAKE_AT(k, y, 0, b);
            y = so.y;  b = so.b;
            ena = so.na;  enb = so.nb;
          }
#else
          if ((u32) (y + k) > (u32) valen || (u32) y > (u32) vblen)          /* past an end: what the reference reads there */
            { const SnakeOut so = snake<REV>(aseq + k, bseq, y, 0, b);
              y = so.y;  b = so.b;
              ena = so.na;  enb = so.nb;
            }
          else
            { /* the snake (align.c:832-856 / 1542-1566) on this lane's sliding windows of the 2-bit bases: 32 bases of
                 each read in registers, the next 16 already on their way, so that a step waits for memory only
                 after a slide longer than a window */
              int pa = va0 + k + y - (REV ? 16 : 0), pb = vb0 + y - (REV ? 16 : 0);   /* first base of the 16 to compare */
              ena = REV ? y + k : valen - (y + k);
              enb = REV ? y : vblen - y;
              for (;;)
                { const int qa = pa >> 4, qb = pb >> 4;
                  if (qa != wda || qb != wdb)
                    { if (qa == wda + (REV ? -1 : 1))
                        { wina = REV ? ((wina << 32) | nxta) : ((wina >> 32) | ((u64) nxta << 32));
                          wda = qa;
                        }
                      else if (qa != wda)
                        { const v2u w = *(const GLOBAL_AS v2u *) ((const GLOBAL_AS char *) (apk - PK_PAD) + 4 * qa);
                          __builtin_amdgcn_s_waitcnt(0x0f70);      /* vmcnt(0) here, so that the windows are never "in flight"
                                                                      where the paths join and only a slide waits for its prefetch */
                          wina = ((u64) w.y << 32) | w.x;
                          wda = qa;
                        }
                      if (qb == wdb + (REV ? -1 : 1))
                        { winb = REV ? ((winb << 32) | nxtb) : ((winb >> 32) | ((u64) nxtb << 32));
                          wdb = qb;
                        }
                      else if (qb != wdb)
                        { const v2u w = *(const GLOBAL_AS v2u *) ((const GLOBAL_AS char *) (bpk - PK_PAD) + 4 * qb);
                          __builtin_amdgcn_s_waitcnt(0x0f70);
                          winb = ((u64) w.y << 32) | w.x;
                          wdb = qb;
                        }
                      nxta = *(const GLOBAL_AS u32 *) ((const GLOBAL_AS char *) (apk - PK_PAD) + 4 * (wda + (REV ? -1 : 2)));
                      nxtb = *(const GLOBAL_AS u32 *) ((const GLOBAL_AS char *) (bpk - PK_PAD) + 4 * (wdb + (REV ? -1 : 2)));
                    }
                  const u32 wa = __builtin_amdgcn_alignbit((u32) (wina >> 32), (u32) wina, (u32) pa * 2);
                  const u32 wb = __builtin_amdgcn_alignbit((u32) (winb >> 32), (u32) winb, (u32) pb * 2);
                  const u32 x = wa ^ wb;
                  const u32 run = (REV ? (u32) __builtin_clzll(((u64) x << 32) | 0x80000000ull)
                                       : (u32) __builtin_ctzll((u64) x | (1ull << 32))) >> 1;
                  const int lim = ena < enb ? ena : enb;
                  const int n = (int) run < lim ? (int) run : lim;
                  b = (b << n) | (u64) ((1u << n) - 1);
                  y  += REV ? -n : n;
                  pa += REV ? -n : n;
                  pb += REV ? -n : n;
                  ena -= n;  enb -= n;
                  if (n < 16 || lim == 16)
                    break;
                }
            }
#endif
          v = (y << 1) + k;
        }
      const bool bhit = act && enb == 0, ahit = act && enb != 0 && ena == 0;

      /* pebbles (align.c:859-909 / 1569-1618): marks as grid indexes, the head's mark in the head */
      { int na = nai * TS + offa, nb = nbi * TS + offb;
        bool needa = act && (REV ? (y + k <= na) : (y + k >= na));
        bool needb = act && (REV ? (y <= nb) : (y >= nb));
        if (wany(needa || needb))
          { int hai = (int) ((u32) ha >> PK_HBITS), hbi = (int) ((u32) hb_ >> PK_HBITS);
            int hax = ha & PK_HMASK, hbx = hb_ & PK_HMASK;
            int g2 = 0;
            for (;;)
              { if (!wany(needa))
                  break;
                GUARD(g2, guard, 5)
                const bool dropit = needa && (REV ? (hai > nai) : (hai < nai));
                const u64  mask = wballot(dropit);
                if (mask)
                  { const u32 hm = hmask(mask, hb);
                    const int idx = ncell + __popc(hm & ((1u << s) - 1u));
                    if (dropit)
                      { if (idx < cell_cap)
                          gcell[cbase + (u32) idx] = cell_pack(hax, k, dif, nai);
                        hax = idx;  hai = nai;
                      }
                    ncell += __popc(hm);
                  }
                if (needa)
                  { nai += S;  na += S * TS; }
                needa = act && (REV ? (y + k <= na) : (y + k >= na));
              }
            for (;;)
              { if (!wany(needb))
                  break;
                GUARD(g2, guard, 6)
                const bool dropit = needb && (REV ? (hbi > nbi) : (hbi < nbi));
                const u64  mask = wballot(dropit);
                if (mask)
                  { const u32 hm = hmask(mask, hb);
                    const int idx = ncell + __popc(hm & ((1u << s) - 1u));
                    if (dropit)
                      { if (idx < cell_cap)
                          gcell[cbase + (u32) idx] = cell_pack(hbx, k, dif, nbi);
                        hbx = idx;  hbi = nbi;
                      }
                    ncell += __popc(hm);
                  }
                if (needb)
                  { nbi += S;  nb += S * TS; }
                needb = act && (REV ? (y <= nb) : (y >= nb));
              }
            ha = hax | (hai << PK_HBITS);  hb_ = hbx | (hbi << PK_HBITS);
          }
      }

      /* commit the new wave */
      if (on)
        { rV = act ? v : edge;
          if (act) { rT = b;  rHA = ha;  rHB = hb_;  rNA = nai;  rNB = nbi; }
        }

      /* sequence ends reached: the largest sweep index for A, the smallest for B (as wave_mem's chunks) */
      { const u64 amw = wballot(ahit), bmw = wballot(bhit);
        if (amw | bmw)
          { const u32 am_ = hmask(amw, hb), bm_ = hmask(bmw, hb);
            if (on && (am_ | bm_))
              { more = 0;
                if (am_) aclip = kbase + KS * (31 - __clz((int) am_));
                if (bm_) bclip = kbase + KS * (__ffs((int) bm_) - 1);
              }
          }
      }

      /* new best / last / trim point in sweep order (align.c:911-928 / 1620-1637): record breakers of a
         prefix maximum; their v is strictly monotone, so the LAST breaker with the wanted property is the one the
         serial sweep leaves behind */
      { const bool cand = act && (REV ? (v < besta) : (v > besta));
        if (wany(cand))
          { const int worst = REV ? BIG : -BIG;
            const int x = pk_prefix_best<REV>(cand ? v : worst);
            int e = __builtin_amdgcn_update_dpp(worst, x, 0x138, 0xf, 0xf, false);             /* wave_shr:1 */
            if (s == 0) e = worst;
            const bool rb = cand && (REV ? (v < e) : (v > e));
            bool mok = false, tok = false;
            if (rb)
              { mok = pk_popc61(b) >= ave;
                if (mok)
                  tok = pk_trim_ok(trimtab, b);
              }
            const u32 m1 = hmask(wballot(rb), hb), m2 = hmask(wballot(rb && mok), hb), m3 = hmask(wballot(rb && tok), hb);
            const int l1 = m1 ? 31 - __clz((int) m1) : 0, l2 = m2 ? 31 - __clz((int) m2) : 0, l3 = m3 ? 31 - __clz((int) m3) : 0;
            const int v1 = hget(v, hb, l1), v2 = hget(v, hb, l2), v3 = hget(v, hb, l3);
            const int h3a = hget(ha, hb, l3), h3b = hget(hb_, hb, l3);
            if (m1)
              { besta = v1;  besty = (v1 - (kbase + KS * l1)) >> 1; }
            if (m2)
              lasta = v2;
            if (m3)
              { trim.a = v3;  trim.y = (v3 - (kbase + KS * l3)) >> 1;  trim.d = dif;
                trim.ha = h3a & PK_HMASK;  trim.hb = h3b & PK_HMASK;
              }
          }
      }
      if (on && ncell > cell_cap)
        { if (s == 0) atomicOr(errw, DAMAR_ERR_CELLS);
          more = 0;  ncell = 2;  bad = 1;  D.fin = 1;  on = false;
        }

      PK_CLIP()

      /* prune (align.c:977-986 / 1686-1695) */
      { const int n = REV ? besta + MAX_WAVE_LAG : besta - MAX_WAVE_LAG;
        const u32 keep = hmask(wballot(on && act && (k >= low) && (k <= hgh) && (REV ? (rV <= n) : (rV >= n))), hb);
        if (on)
          { if (keep == 0)
              hgh = low - 1;
            else
              { const int s0 = __ffs((int) keep) - 1, s1 = 31 - __clz((int) keep);
                if (REV) { low = kbase + s0;  hgh = kbase + s1; }
                else     { hgh = kbase - s0;  low = kbase - s1; }
              }
          }
      }
    }
#undef PK_CLIP
#ifdef DAMAR_PROF
  PROF_ADD(26, pf_iters);  PROF_ADD(27, pf_half);  PROF_ADD(28, __popcll(wballot(ovf != 0)) >> 5);  PROF_ADD(29, 1);
#endif
}

#undef aseq
#undef bseq
#undef va0
#undef vb0
#undef valen
#undef vblen
#undef steplimit
#undef guard

/* A half whose band outgrew its 32 lanes borrows the whole wavefront: the band goes to the one-alignment-per-wavefront
 * register path (wave_reg_cont<REV>: lane (k & 63) owns diagonal k, marks as values) and comes back as soon as it
 * fits a half again (hgh - low + 3 <= PK_NARROW) -- bands wider than 29 diagonals last a few steps -- or finishes the
 * direction there (through wave_mem<REV> if it outgrows the wavefront too).  Called for one half at a time with every
 * lane active; hsel = that half's lane base (0 or 32).  *Dp, *io are per-lane copies: only the half's lanes are changed. */
template <int REV>
__device__ __noinline__ void pk_solo(int job, const u32 *trimtab, SlotScratch sc, PkPair p, int hsel, int mida,
                                     PkDir *Dp, LaneRegs *io)
{ const ReportArgs &a = g_jobs[uni(job)];
  const int lane = lane_id();
  const int KS = REV ? 1 : -1;
  const int TS = a.tspace;
  const int edge = REV ? BIG : -1;
  const int src = hsel;
  WaveCtx c;
  WaveState ws;
#define PK_PTR_OF(T, ptr) ((T) (uintptr_t) (((u64) (u32) bcast_i((int) (u32) ((u64) (uintptr_t) (ptr) >> 32), src) << 32) | \
                                            (u32) bcast_i((int) (u32) (u64) (uintptr_t) (ptr), src)))
  c.a0 = (u32) bcast_i(p.a0, src);  c.b0 = (u32) bcast_i(p.b0, src);
  c.aseq = a.ablk.bases + c.a0;  c.bseq = a.bblk.bases + c.b0;
  c.apk = a.ablk.pk;  c.bpk = a.bblk.pk;
  c.alen = bcast_i(p.alen, src);  c.blen = bcast_i(p.blen, src);
  c.ts = TS;  c.ave = a.ave_path;  c.reach = a.reach;
  c.score = a.score;  c.table = a.table;  c.trim8 = trimtab;
  c.minp = bcast_i(p.minp, src);  c.maxp = bcast_i(p.maxp, src);
  c.aoff = 0;  c.boff = bcast_i(p.boff, src);
  c.st0 = PK_PTR_OF(DState *, sc.st0);  c.st1 = PK_PTR_OF(DState *, sc.st1);
  c.NA = PK_PTR_OF(int *, sc.NA);  c.NB = PK_PTR_OF(int *, sc.NB);
  c.koff = c.blen + 8;  c.ring = a.span;
  c.cells = PK_PTR_OF(Cell *, sc.cells);  c.cell_cap = a.cell_cap;
  c.err = &a.counters[3];
  c.atr = PK_PTR_OF(u16 *, sc.atr);  c.btr = PK_PTR_OF(u16 *, sc.btr);
#undef PK_PTR_OF
  ws.low = bcast_i(Dp->low, src);  ws.hgh = bcast_i(Dp->hgh, src);  ws.dif = bcast_i(Dp->dif, src);
  ws.besta = bcast_i(Dp->besta, src);  ws.besty = bcast_i(Dp->besty, src);  ws.lasta = bcast_i(Dp->lasta, src);
  int *const cold = pk_cold + (hsel >> 1);                  /* the half's event record (every lane reads the same words) */
  ws.more = bcast_i(Dp->more, src);  ws.reachm = uni(cold[PKC_REACHM]);
  ws.aclip = uni(cold[PKC_ACLIP]);  ws.bclip = uni(cold[PKC_BCLIP]);
  ws.ncell = (u32) bcast_i(Dp->ncell, src);
  ws.trim.a = uni(cold[PKC_TRIM]);  ws.trim.y = uni(cold[PKC_TRIM + 1]);  ws.trim.d = uni(cold[PKC_TRIM + 2]);
  ws.trim.ha = uni(cold[PKC_TRIM + 3]);  ws.trim.hb = uni(cold[PKC_TRIM + 4]);
  ws.reach.a = uni(cold[PKC_REACH]);  ws.reach.y = uni(cold[PKC_REACH + 1]);  ws.reach.d = uni(cold[PKC_REACH + 2]);
  ws.reach.ha = uni(cold[PKC_REACH + 3]);  ws.reach.hb = uni(cold[PKC_REACH + 4]);
  ws.stopped = 0;  ws.bad = 0;  ws.narrow = 0;
  /* the half's band into the 64-lane layout (lane (k & 63) owns diagonal k) */
  LaneRegs r;
  { const int kbase = bcast_i(Dp->kbase, src);
    const int k = ws.low + ((lane - ws.low) & 63);
    const bool in = k <= ws.hgh;
    const int sl = (hsel + (in ? KS * (k - kbase) : 0)) << 2;
    r.V  = __builtin_amdgcn_ds_bpermute(sl, io->V);
    r.HA = __builtin_amdgcn_ds_bpermute(sl, io->HA);
    r.HB = __builtin_amdgcn_ds_bpermute(sl, io->HB);
    r.NA = __builtin_amdgcn_ds_bpermute(sl, io->NA);
    r.NB = __builtin_amdgcn_ds_bpermute(sl, io->NB);
    { const u32 tl = (u32) __builtin_amdgcn_ds_bpermute(sl, (int) (u32) io->T);
      const u32 th = (u32) __builtin_amdgcn_ds_bpermute(sl, (int) (u32) (io->T >> 32));
      r.T = ((u64) th << 32) | tl;
    }
    if (!in)
      r.V = edge;
  }
  wave_mem_sync();
  wave_reg_cont<REV>(c, mida, ws, &r);
  if (ws.narrow)
    { /* back into the half, centred */
      const int w = ws.hgh - ws.low + 1, slo = (32 - w) >> 1;
      const int kbase = REV ? ws.low - slo : ws.hgh + slo;
      const int k = kbase + KS * (lane & 31);
      const bool in = k >= ws.low && k <= ws.hgh;
      const int sl = (k & 63) << 2;
      const int nV = __builtin_amdgcn_ds_bpermute(sl, r.V);
      const int nHA = __builtin_amdgcn_ds_bpermute(sl, r.HA), nHB = __builtin_amdgcn_ds_bpermute(sl, r.HB);
      const int nNA = __builtin_amdgcn_ds_bpermute(sl, r.NA), nNB = __builtin_amdgcn_ds_bpermute(sl, r.NB);
      const u32 tl = (u32) __builtin_amdgcn_ds_bpermute(sl, (int) (u32) r.T);
      const u32 th = (u32) __builtin_amdgcn_ds_bpermute(sl, (int) (u32) (r.T >> 32));
      if ((lane & 32) == hsel)
        { io->V = in ? nV : edge;  io->HA = nHA;  io->HB = nHB;  io->NA = nNA;  io->NB = nNB;
          io->T = ((u64) th << 32) | tl;
          Dp->kbase = kbase;
          Dp->ovf = 0;
        }
    }
  else
    { if (!ws.stopped)
        wave_mem<REV>(c, mida, ws);
      if ((lane & 32) == hsel)
        { Dp->ovf = 0;  Dp->fin = 1;  Dp->more = 0;  Dp->bad = ws.bad; }
    }
  if ((lane & 32) == hsel)
    { Dp->low = ws.low;  Dp->hgh = ws.hgh;  Dp->dif = ws.dif;  Dp->besta = ws.besta;  Dp->besty = ws.besty;
      Dp->lasta = ws.lasta;  Dp->more = (ws.narrow ? ws.more : 0);
      Dp->ncell = (int) ws.ncell;
    }
  cold[PKC_REACHM] = ws.reachm;  cold[PKC_ACLIP] = ws.aclip;  cold[PKC_BCLIP] = ws.bclip;
  cold[PKC_TRIM] = ws.trim.a;  cold[PKC_TRIM + 1] = ws.trim.y;  cold[PKC_TRIM + 2] = ws.trim.d;
  cold[PKC_TRIM + 3] = ws.trim.ha;  cold[PKC_TRIM + 4] = ws.trim.hb;
  cold[PKC_REACH] = ws.reach.a;  cold[PKC_REACH + 1] = ws.reach.y;  cold[PKC_REACH + 2] = ws.reach.d;
  cold[PKC_REACH + 3] = ws.reach.ha;  cold[PKC_REACH + 4] = ws.reach.hb;
}

/* End point and trace points of one direction (align.c:1001-1118 / 1699-1898) for the halves with `fin`: the first
 * lane of the half walks the two pebble chains exactly as wave_finish<REV> does.  In/out per half: atlen, btlen. */
template <int REV>
__device__ __noinline__ void pk_finish(Cell *cells, u16 *atrace, u16 *btrace, bool fin, int TS, int aoff, int boff,
                                       int do_reach, int guard, u32 *errw, int mida, int reachm,
                                       int ta, int ty, int td, int tha, int thb,
                                       int *ox, int *oy, int *od, int *atlen_io, int *btlen_io, int *aback, int *bback)
{ const int lane = lane_id(), hb = lane & 32, s = lane & 31;
  int rx = 0, ry = 0, rd = 0, at = 0, bt = 0;
  (void) do_reach; (void) reachm;
  if (fin && s == 0)
    { int trimx = ta - ty, trimy = ty, trimd = td, ha = tha, hb_ = thb;
      int gw = 0;
      if (!REV)
        { at = chain_to_trace<0, 0>(cells, ha, TS, aoff, mida, trimx, trimy, trimd, atrace, 0, guard, gw, errw);
          bt = chain_to_trace<0, 1>(cells, hb_, TS, boff, mida, trimx, trimy, trimd, btrace, 0, guard, gw, errw);
        }
      else
        { at = chain_to_trace<1, 0>(cells, ha, TS, aoff, mida, trimx, trimy, trimd, atrace, *atlen_io, guard, gw, errw);
          bt = chain_to_trace<1, 1>(cells, hb_, TS, boff, mida, trimx, trimy, trimd, btrace, *btlen_io, guard, gw, errw);
        }
      rx = trimx;  ry = trimy;  rd = trimd;
    }
  wave_mem_sync();
  rx = hget(rx, hb, 0);  ry = hget(ry, hb, 0);  rd = hget(rd, hb, 0);  at = hget(at, hb, 0);  bt = hget(bt, hb, 0);
  if (fin)
    { *ox = rx;  *oy = ry;  *od = rd;
      if (!REV)
        { *atlen_io = at;  *btlen_io = bt; }
      else
        { *aback = at;  *bback = bt;
          *atlen_io += at;  *btlen_io += bt;
        }
    }
}

/* One direction for both halves: wave 0, the packed loop, the whole wavefront for a half whose band outgrew its
 * lanes (and back), trace walk.  Its own function (noinline, inputs by value) so that the wave loop gets its registers
 * allocated on its own: what the state machine keeps alive sits in the caller's frame, not in the loop's way. */
struct PkOut { int x, y, d, atlen, btlen, aback, bback; };

template <int REV>
__device__ __noinline__ void pk_pass(int job, const u32 *trimtab, SlotScratch sc, int task_, PkPair p, u32 cbase,
                                     int diag, int mida, PkOut *out)
{ const ReportArgs &a = g_jobs[uni(job)];
  const bool task = task_ != 0;
  PkDir D;
  int rV, rHA, rHB, rNA, rNB;
  u64 rT;
#ifdef DAMAR_PROF
  const unsigned long long pf_t0 = wall_clock64();
  unsigned long long pf_solo = 0;
#endif
  pk_init<REV>(a, task, p, cbase, diag, mida, D, rV, rT, rHA, rHB, rNA, rNB);
  for (bool first = true; ; first = false)
    { pk_loop<REV>(a, trimtab, task, first, p, cbase, D, rV, rT, rHA, rHB, rNA, rNB);
      const u64 ov = wballot(task && D.ovf);
      if (!ov)
        break;
#ifdef DAMAR_PROF
      const unsigned long long pf_s0 = wall_clock64();
#endif
      for (int h = 0; h < 64; h += 32)
        if ((ov >> h) & 1)
          { PkDir Dc = D;                     /* copies: what a noinline callee may write must not pin the loop's state to memory */
            LaneRegs io;
            io.V = rV;  io.HA = rHA;  io.HB = rHB;  io.NA = rNA;  io.NB = rNB;  io.T = rT;
            pk_solo<REV>(a.job, trimtab, sc, p, h, bcast_i(mida, h), &Dc, &io);
            D = Dc;
            rV = io.V;  rHA = io.HA;  rHB = io.HB;  rNA = io.NA;  rNB = io.NB;  rT = io.T;
          }
#ifdef DAMAR_PROF
      pf_solo += wall_clock64() - pf_s0;
#endif
    }
  wave_mem_sync();
#ifdef DAMAR_PROF
  const unsigned long long pf_t2 = wall_clock64();
#endif
  /* the direction's end point: the trim point, or the reach candidate (align.c:1009-1016) */
  const int *const cold = pk_cold + ((lane_id() & 32) >> 1);
  const int rm = cold[PKC_REACHM];
  int ta = cold[PKC_TRIM], ty = cold[PKC_TRIM + 1], td = cold[PKC_TRIM + 2], tha = cold[PKC_TRIM + 3], thb = cold[PKC_TRIM + 4];
  if (rm >= 0 && a.reach)
    { ta = cold[PKC_REACH];  ty = cold[PKC_REACH + 1];  td = cold[PKC_REACH + 2];  tha = cold[PKC_REACH + 3];  thb = cold[PKC_REACH + 4]; }
  int ox = 0, oy = 0, od = 0, atl = out->atlen, btl = out->btlen, ab = out->aback, bb = out->bback;
  pk_finish<REV>(sc.cells, sc.atr, sc.btr, task && !D.bad, a.tspace, 0, p.boff, a.reach,
                 4 * (p.alen + p.blen) + 1024, &a.counters[3], mida, rm, ta, ty, td, tha, thb,
                 &ox, &oy, &od, &atl, &btl, &ab, &bb);
  out->x = ox;  out->y = oy;  out->d = od;  out->atlen = atl;  out->btlen = btl;  out->aback = ab;  out->bback = bb;
#ifdef DAMAR_PROF
  PROF_ADD(15, pf_t2 - pf_t0 - pf_solo);  PROF_ADD(13, pf_solo);  PROF_ADD(14, wall_clock64() - pf_t2);
#endif
}

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

/***** the per-half state machine of the report loop ******************************************************/

enum { PK_ITEM = 0, PK_PANEL, PK_FIRE, PK_DONE };

#ifndef PK_WAVES
#define PK_WAVES 5                      /* resident wavefronts per SIMD the packed kernel is compiled for (VGPR budget 512 / PK_WAVES).
                                           Report ms per config-2 step with the event state of a direction parked in LDS (pk_cold):
                                           4 -> 327, 5 -> 288 (wave loop free of spills in both directions), 6 -> 286 with a dozen
                                           reloads per step in the reverse loop but 395 after an unrelated edit (the allocation
                                           at 80 VGPRs is a coin toss), 7 spills throughout.  Limiting the launch to
                                           1 / 2 / 3 / 4 wavefronts per SIMD (DAMAR_SLOTS) gives 1073 / 556 / 407 / 337 ms: the
                                           kernel is bound by how long ONE wavefront takes per step, residency is what hides it */
#endif
#ifndef DUO_WAVES
#define DUO_WAVES 8                     /* the same for report_duo.h's kernel, whose wave loop fits 64 VGPRs */
#endif
/* which of the two packed kernels runs: report_duo.h's (halves independent) unless DAMAR_DUO=0 asks for this file's */
static int duo_enabled(void)
{ static int duo = -1;
  if (duo < 0)
    { const char *e = getenv("DAMAR_DUO");
      duo = e ? atoi(e) : 1;
    }
  return duo;
}
int damar_report2_waves_per_simd(void) { return duo_enabled() ? DUO_WAVES : PK_WAVES; }

/* one job of the launch: the two halves pull read pairs (or batch tasks) from its queue until it is empty */
__device__ __forceinline__ void report2_job(const ReportArgs &a, const u32 *trimtab, const LaTask *tasks, u32 ntasks)
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
  const bool batch = tasks != NULL;

  int  phase = PK_ITEM;
  u32  item = 0, seq = 0;
  u64  nidx = 0, cpair = 0, lidx = 0, end = 0, h2 = 0, fp = 0;
  int  ar = 0, br = 0, amark2 = 0, clo = BIG, chi = -BIG;
  PkPair p;
  p.a0 = p.b0 = 0;  p.alen = p.blen = 0;  p.minp = -BIG;  p.maxp = BIG;  p.boff = 0;

  for (;;)
    { bool task = false;
      int  sdg = 0, sap = 0, sd = 0;

      /* A: every half advances its scan until it has an alignment to compute or has run out of work */
      while (wany(phase != PK_DONE && !task))
        { if (phase == PK_ITEM && !task)
            { u32 it = 0;
              if (s == 0)
                it = atomicAdd(a.cursor, 1u);
              it = (u32) hget((int) it, hb, 0);
              if (it >= (batch ? ntasks : a.nwork))
                phase = PK_DONE;
              else if (batch)
                { const LaTask tk = tasks[it];
                  item = it;  seq = 0;
                  ar = tk.aread;  br = tk.bread;
                  p.a0 = (int) a.ablk.boff[ar];  p.b0 = (int) a.bblk.boff[br];
                  p.alen = (int) read_len(a.ablk, ar);  p.blen = (int) read_len(a.bblk, br);
                  sdg = tk.diag;  sap = 0;  sd = tk.anti;            /* (sd carries the anti-diagonal of a batch task) */
                  task = true;
                }
              else
                { item = a.order ? a.order[it] : it;
                  nidx = a.work[item];
                  cpair = keys[nidx] >> pshift;
                  ar = (int) (cpair & ((1ull << a.abits) - 1));  br = (int) (cpair >> a.abits);
                  p.a0 = (int) a.ablk.boff[ar];  p.b0 = (int) a.bblk.boff[br];
                  p.alen = (int) read_len(a.ablk, ar);  p.blen = (int) read_len(a.bblk, br);
                  seq = 0;  amark2 = 0;  clo = BIG;  chi = -BIG;
                  if (!(p.alen < a.hgap_min && p.blen < a.hgap_min))
                    phase = PK_PANEL;
                }
            }
          else if (phase == PK_PANEL && !task)
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
                          if (in)
                            { const int add = (ap - prev >= K) ? K : ap - prev;
                              atomicAdd(&sc.score[d], add);
                              if (last)
                                sc.lastp[d] = ap;
                            }
                          wave_mem_sync();
                        }
                      fp = lidx;
                      phase = PK_FIRE;
                    }
                  else
                    nidx = h2;
                }
            }
          else if (phase == PK_FIRE && !task)
            { /* pass 2 (filter.c:2283-2405): the next seed in order with enough score whose apos is beyond lasta */
              bool found = false;
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
                task = true;
              else
                { /* pass 3: reset the touched buckets (filter.c:2407-2411) */
                  for (u64 base = lidx; base < end; base += 32)
                    { const u64 f = base + s;
                      if (f < end)
                        { const int d = seed_diag(keys[f], vals, f, pmask, dbits) >> W;
                          sc.score[d] = 0;
                          sc.lastp[d] = 0;
                        }
                    }
                  wave_mem_sync();
                  nidx = h2;
                  phase = PK_PANEL;
                }
            }
        }
      if (!wany(task))
        break;

      /* B: Local_Alignment (align.c:1904-2097 for low == hgh == diag) for the halves that hold a task */
      int diag = sdg, anti = batch ? sd : sap + (sap - sdg);
      const bool selfie = (a.ablk.bases + p.a0 == a.bblk.bases + p.b0);
      p.minp = (selfie && diag >= 0) ? 1 : -BIG;
      p.maxp = (selfie && diag <= 0) ? -1 : BIG;
      p.boff = (a.comp & 1) ? (p.blen % a.tspace) : 0;
      if (task && s == 0 && !batch)
        atomicAdd(a.nfilt, 1u);
      LaResult r;
      { PkOut o;
        o.x = o.y = o.d = o.atlen = o.btlen = o.aback = o.bback = 0;
        pk_pass<0>(a.job, trimtab, sc, task ? 1 : 0, p, cbase, diag, anti, &o);
        r.aepos = o.x;  r.bepos = o.y;  r.diffs = o.d;
        pk_pass<1>(a.job, trimtab, sc, task ? 1 : 0, p, cbase, diag, anti, &o);
        r.abpos = o.x;  r.bbpos = o.y;  r.diffs += o.d;
        r.atlen = o.atlen;  r.btlen = o.btlen;  r.aback = o.aback;  r.bback = o.bback;
      }

      /* C: what the reference does with the path (filter.c:2318-2380) */
      if (batch)
        { pk_emit(a, sc, task, r, ar, br, item, 0);
          phase = PK_ITEM;
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
        }
    }
}

__global__ __launch_bounds__(64, PK_WAVES)
void report2_kernel(int njobs, const LaTask *tasks, u32 ntasks)
{ __shared__ u32 trimtab[256];
  pk_fill_trimtab(trimtab, g_jobs[0].mscore, g_jobs[0].dscore);
  __syncthreads();
#ifdef DAMAR_PROF
  struct PfExit { unsigned long long t0; __device__ ~PfExit() { unsigned long long d = wall_clock64() - t0;
    if (lane_id() == 0) { atomicAdd(&g_prof[23], d); atomicAdd(&g_prof[25], 1ull); atomicMax(&g_prof[24], d); } } } pf_exit = { (unsigned long long) wall_clock64() };
#endif
  for (int turn = 0; turn < njobs; turn++)
    report2_job(g_jobs[((int) blockIdx.x + turn) % njobs], trimtab, tasks, ntasks);
}

void damar_launch_report3(const ReportArgs *jobs, int njobs, const LaTask *tasks, u32 ntasks, int nslots, hipStream_t st);
void damar_launch_report2(const ReportArgs *jobs, int njobs, const LaTask *tasks, u32 ntasks, int nslots, hipStream_t st)
{ if (duo_enabled())
    { damar_launch_report3(jobs, njobs, tasks, ntasks, nslots, st);
      return;
    }
  jobs_upload(jobs, njobs, st);
  hipLaunchKernelGGL(report2_kernel, dim3(nslots / 2), dim3(64), 0, st, njobs, tasks, ntasks);
}

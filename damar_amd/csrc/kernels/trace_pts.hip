/* trace_pts.hip -- trace-point expansion of overlap records on gfx950.
 *
 * Replaces, for whole batches of records, the reference's
 *   Compute_Trace_PTS   dalign/align.c:5577-5692   (record -> trace-point segments)
 *   iter_np             dalign/align.c:4892-5261   (O(np) waves of one segment, every wave kept, path
 *                                                   reversal, edit script of the indels)
 * which utils/LAshow.c:245-262 calls per record on one CPU thread.
 *
 * Mapping.  The segments between trace points are independent (~100 x 100 bases, ~25 differences each,
 * tens of millions per block pair) while the cells of one segment's wave are not: a wave is computed
 * from the diagonals furthest from `del` inwards and every cell needs the cell just computed next to it
 * (the free move of the O(np) scheme), then slides down its snake.  So the parallel axis is the segment:
 * one lane per segment, 64 neighbouring segments of the same records per wavefront (neighbours have
 * similar difference counts, which keeps the lanes' trip counts close).  Every wave D of a segment is kept
 * (furthest B index as int16, predecessor code as int8) in the lane's own stripe of an HBM scratch area,
 * rows packed back to back (row D holds |del| + 3 + 2*(D/2) diagonals), because the path reversal and the
 * script emission walk them again.  A segment whose waves outgrow the stripe is deferred to a second launch
 * with stripes sized for the largest possible wave count of the batch; nothing is dropped silently.
 *
 *   trace_layout   one thread per record : segment descriptors (the host has sized the staging slots)
 *   trace_waves    one thread per segment: waves, reversal (GREEDIEST / LOWERMOST / UPPERMOST), script into
 *                                          the segment's staging slot
 *   trace_gather   one thread per record : script length and summed distance, offsets of its segments
 *   trace_pack     one thread per segment: staging -> the record's contiguous script
 */
#include "kernels.h"

#define TP_THREADS 64

template <typename PT>
__global__ __launch_bounds__(256)
void trace_layout(const TraceRecIn *__restrict__ recs, u32 nrecs, const PT *__restrict__ pts, int tspace,
                  DevBlock ablk, DevBlock bblk, TraceSeg *__restrict__ segs, u32 *__restrict__ key, u32 *__restrict__ val,
                  u32 *__restrict__ err)
{ const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrecs) return;
  const TraceRecIn in = recs[r];
  const PT *p = pts + in.poff;
  const int tlen = in.tlen;
  const u32 aoff = ablk.boff[in.aread], alen = ablk.boff[in.aread + 1] - aoff - 1;
  const u32 boff = bblk.boff[in.bread], blen = bblk.boff[in.bread + 1] - boff - 1;
  const int comp = (int) (in.flags & 1u);
  int ab = in.abpos, ae = (ab / tspace) * tspace, bb = in.bbpos;
  const int nseg = tlen >= 2 ? tlen / 2 : 1;
  u32 so = in.stage0;
  int s;
  for (s = 0; s < nseg; s++)
    { int be;
      if (s + 1 < nseg) { ae += tspace; be = bb + (int) p[2 * s + 1]; }
      else              { ae = in.aepos; be = in.bepos; }
      if (ae > (int) alen || be > (int) blen || ae < ab || be < bb || ae - ab > 32000 || be - bb > 32000)
        break;                                            /* align.c:5659, 5671 TP_Error */
      const int M = ae - ab, N = be - bb, del = M - N;
      TraceSeg g;
      g.apos = aoff + (u32) ab;
      g.bpos = comp ? boff + blen - 1 - (u32) bb : boff + (u32) bb;
      g.a0 = ab;  g.b0 = bb;
      g.mn = (u32) M | ((u32) N << 16);
      /* the segment's own difference count (0 for a record without trace points) orders the work: lanes of
         one wavefront should need about the same number of waves */
      const u32 own = tlen >= 2 ? min((u32) p[2 * s], 255u) : 0u;     /* (sorted on 255 - own: heaviest first) */
      g.flags = (in.flags & 3u) | ((u32) min(in.dmax, 65535) << 8) | (own << 24);
      g.stage = so;
      g.rec = r;
      segs[in.seg0 + s] = g;
      key[in.seg0 + s] = 255u - own;
      val[in.seg0 + s] = in.seg0 + s;
      so += (u32) (in.dmax + (del < 0 ? -del : del));
      ab = ae;
      bb = be;
    }
  if (s < nseg)
    { atomicOr(err, DAMAR_TRACE_ERR_POINTS);
      for (s = 0; s < nseg; s++)                          /* void segments: the host stops on the flag */
        { TraceSeg g = {};  g.rec = r;  g.flags = 4u;  segs[in.seg0 + s] = g;
          key[in.seg0 + s] = 255u;  val[in.seg0 + s] = in.seg0 + s;
        }
    }
}

/* the three-way choice of align.c:4981-5004: ties go to the free neighbour, then to the substitution */
__device__ __forceinline__ int choose(int am, int ac, int ap, int mcode, int pcode, int &code)
{ if (ac < am)
    { if (ap < am) { code = mcode; return am; }
      code = pcode;
      return ap;
    }
  if (ap < ac) { code = 0; return ac; }
  code = pcode;
  return ap;
}

/***** where the waves of a segment live ************************************************************
 *
 * Row D >= 0 spans the diagonals low0 - D/2 - 1 .. hgh0 + D/2 + 1 (one sentinel either side), rows -2
 * and -1 the span of row 0.  Two layouts:
 *
 *  Stripe   the lane's own stripe of `cap` cells, rows packed back to back.  Any segment fits a stripe
 *           that is large enough; the lanes of a wavefront touch 64 different cache lines per access.
 *  Slots    one area per wavefront, cell (row, slot) of the 64 lanes side by side (128 bytes of int16).
 *           A row is laid out in the order in which a wave is computed -- hgh+1, hgh, ... down to del,
 *           and low-1, low, ... up to del -- so that the 64 lanes, which the hardware steps through the
 *           loops together, sit on the same slot at the same time whatever their del: one cache line
 *           per access.
 */
struct Stripe
{ short *vf;  signed char *hf;  u32 cap;
  int w0, low0, del;
  __device__ __forceinline__ void shape(int d)
  { del = d;
    w0 = (d < 0 ? -d : d) + 3;
    low0 = d < 0 ? d : 0;
  }
  __device__ __forceinline__ int start(int D) const
  { const int m = D > 0 ? D >> 1 : 0, q = D > 0 ? (D - 1) >> 1 : 0;
    return (D + 2) * w0 + 2 * m * q;
  }
  __device__ __forceinline__ int at(int D, int k) const
  { const int m = D > 0 ? D >> 1 : 0;
    return start(D) + k - (low0 - m - 1);
  }
  __device__ __forceinline__ bool fits(int D) const        { return (u32) start(D + 1) <= cap; }
  __device__ __forceinline__ u32  rows_needed(int dmax) const { return (u32) start(dmax + 1); }
  __device__ __forceinline__ int  v(int D, int k) const    { return vf[at(D, k)]; }
  __device__ __forceinline__ int  va(int D, int k) const   { return v(D, k); }     /* k >= del */
  __device__ __forceinline__ int  vb(int D, int k) const   { return v(D, k); }     /* k <= del */
  __device__ __forceinline__ void setv(int D, int k, int x){ vf[at(D, k)] = (short) x; }
  __device__ __forceinline__ int  h(int D, int k) const    { return hf[at(D, k)]; }
  __device__ __forceinline__ void seth(int D, int k, int x){ hf[at(D, k)] = (signed char) x; }
  __device__ __forceinline__ void put_above(int D, int k, int x, int code) { setv(D, k, x);  seth(D, k, code); }
  __device__ __forceinline__ void put_below(int D, int k, int x, int code) { setv(D, k, x);  seth(D, k, code); }
  __device__ __forceinline__ void put_del(int D, int x, int code)          { setv(D, del, x);  seth(D, del, code); }
  static const int recompute = 0;
};

#define SLOT_SS    40                      /* slots either side: sentinel, diagonals, del */
#define SLOT_RS    (2 * SLOT_SS)
#define SLOT_ROWS  64                      /* rows -2 .. 61 */
#define SLOT_MAXB  240                     /* bases of A / of B staged into LDS */
#define SLOT_WORDS 17                      /* (15 + 240 + 15) / 16 + 1 */

/* A row is kept as two runs of slots, both ending at diagonal del, which is therefore stored twice:
   "above" = hgh+1, hgh, ..., del (slots 0 ..), "below" = low-1, low, ..., del (slots SS ..).  Within a
   run the slot is linear in k, so the loops of a wave step through consecutive lines.  The furthest
   points fit a byte here (-2 .. SLOT_MAXB, kept + 2).  RING: only the rows of waves D, D-1, D-2 are live
   (GREEDIEST never reads an older furthest point: the script positions are found again by sliding along
   the chosen path), so four rows are reused in turn and stay in the caches; the predecessor codes always
   keep every row. */
template <int RING>
struct Slots
{ u8 *vf;  signed char *hf;                /* the wavefront's area: uniform, so that accesses are scalar base + lane offset */
  u32 lane;
  int del, low0, hgh0, rows;
  static const int recompute = RING;
  __device__ __forceinline__ void shape(int d)
  { del = d;
    low0 = d < 0 ? d : 0;
    hgh0 = d < 0 ? 0 : d;
  }
  __device__ __forceinline__ int above(int D, int k) const          /* slot of k in [del, hgh(D) + 1] */
  { const int m = D > 0 ? D >> 1 : 0;
    return hgh0 + m + 1 - k;
  }
  __device__ __forceinline__ int below(int D, int k) const          /* slot of k in [low(D) - 1, del] */
  { const int m = D > 0 ? D >> 1 : 0;
    return SLOT_SS + (k - (low0 - m - 1));
  }
  __device__ __forceinline__ int slot(int D, int k) const  { return k >= del ? above(D, k) : below(D, k); }
  __device__ __forceinline__ int vrow(int D) const         { return RING ? ((D + 2) & 3) : D + 2; }
  __device__ __forceinline__ u32 vi(int D, int sl) const   { return (u32) ((vrow(D) * SLOT_RS + sl) * 64) + lane; }
  __device__ __forceinline__ u32 hi(int D, int sl) const   { return (u32) (((D + 2) * SLOT_RS + sl) * 64) + lane; }
  __device__ __forceinline__ bool fits(int D) const
  { const int m = D >> 1;
    return D + 2 < rows && hgh0 + m + 1 - del < SLOT_SS && del - (low0 - m - 1) < SLOT_SS;
  }
  __device__ __forceinline__ int  v(int D, int k) const    { return (int) vf[vi(D, slot(D, k))] - 2; }
  __device__ __forceinline__ int  va(int D, int k) const   { return (int) vf[vi(D, above(D, k))] - 2; }
  __device__ __forceinline__ int  vb(int D, int k) const   { return (int) vf[vi(D, below(D, k))] - 2; }
  __device__ __forceinline__ void setv(int D, int k, int x)
  { if (k >= del) vf[vi(D, above(D, k))] = (u8) (x + 2);
    if (k <= del) vf[vi(D, below(D, k))] = (u8) (x + 2);
  }
  __device__ __forceinline__ int  h(int D, int k) const    { return hf[hi(D, slot(D, k))]; }
  __device__ __forceinline__ void seth(int D, int k, int x){ hf[hi(D, slot(D, k))] = (signed char) x; }
  __device__ __forceinline__ void put_above(int D, int k, int x, int code)
  { const int sl = above(D, k);  vf[vi(D, sl)] = (u8) (x + 2);  hf[hi(D, sl)] = (signed char) code; }
  __device__ __forceinline__ void put_below(int D, int k, int x, int code)
  { const int sl = below(D, k);  vf[vi(D, sl)] = (u8) (x + 2);  hf[hi(D, sl)] = (signed char) code; }
  __device__ __forceinline__ void put_del(int D, int x, int code)
  { const int sl = above(D, del);  vf[vi(D, sl)] = (u8) (x + 2);  hf[hi(D, sl)] = (signed char) code;
    vf[vi(D, below(D, del))] = (u8) (x + 2);
  }
};

/***** how a snake is followed ************************************************************************/

/* base by base on the blocks' byte arrays (any segment; B read backwards and complemented for COMP) */
struct ByteBases
{ const u8 *A, *B;
  int sgn, comp;
  __device__ __forceinline__ int a(int i) const { return A[i]; }
  __device__ __forceinline__ int b(int j) const
  { const int x = B[sgn * j];
    return (comp && x < 4) ? 3 - x : x;
  }
  __device__ __forceinline__ int slide(int k, int j, int lim) const
  { while (j < lim && b(j) == a(k + j)) j++;
    return j;
  }
};

/* 16 bases per step on 2-bit copies of the two segments in LDS (word w of lane l at [w * 64 + l]) */
struct LdsBases
{ const u32 *la, *lb;                      /* already offset by the lane */
  int oa, ob;                              /* position of A[0] / B[0] in the packed copies */
  __device__ __forceinline__ u32 win(const u32 *x, int pos) const
  { const int w = pos >> 4;
    return __builtin_amdgcn_alignbit(x[(w + 1) * 64], x[w * 64], (u32) pos * 2);
  }
  __device__ __forceinline__ int slide(int k, int j, int lim) const
  { while (j < lim)
      { const u32 x = win(la, oa + k + j) ^ win(lb, ob + j);
        const int run = (int) ((u32) __builtin_ctzll((u64) x | (1ull << 32)) >> 1);
        j = min(j + run, lim);
        if (run < 16) break;
      }
    return j;
  }
};

__device__ __forceinline__ u32 pk_window(const u32 *pk, u32 pb)       /* 16 bases from biased position pb */
{ const u32 *q = (pk - PK_PAD) + (pb >> 4);
  return __builtin_amdgcn_alignbit(q[1], q[0], pb * 2);
}

/* One edge back from cell (D, k) with predecessor code e: the cell it came from, after the re-routing of
   UPPERMOST (align.c:5056-5120) / LOWERMOST (:5122-5186); the same text serves iter_np and middle_np.
   c is the reference's running B index.  Returns the predecessor's diagonal; D, e, c are updated. */
template <int MODE, class Cells>
__device__ __forceinline__ int back_edge(Cells &w, const ByteBases &bb, int del, int &D, int k, int &e, int &c)
{ int h = k + e, m;
  if (e > 1) h -= 3;
  else if (e == 0) D -= 1;
  else D -= 2;
  if (MODE == 1 && h < k)
    { m = k < 0 ? -k : 0;
      const int x = w.v(D, h);
      if (x <= c) c = x - 1;
      while (c >= m && bb.a(k + c) == bb.b(c)) c -= 1;
      if (e == -1)
        { if (c <= w.v(D + 2, k + 1))          { e = 4; h = k + 1; D = D + 2; }
          else if (c == w.v(D + 1, k))         { e = 0; h = k;     D = D + 1; }
          else w.setv(D, h, c + 1);
        }
      else
        { m = (k == del) ? D : D - 2;
          if (c <= w.v(m, k + 1))              { e = (k == del) ? 4 : 1; h = k + 1; D = m; }
          else if (c == w.v(D - 1, k))         { e = 0; h = k; D = D - 1; }
          else w.setv(D, h, c + 1);
        }
    }
  else if (MODE == -1 && h > k)
    { m = k < 0 ? -k : 0;
      const int x = w.v(D, h);
      if (x < c) c = x;
      while (c >= m && bb.a(k + c) == bb.b(c)) c -= 1;
      if (e == 1)
        { if (c < w.v(D + 2, k - 1))           { e = 2; h = k - 1; D = D + 2; }
          else if (c == w.v(D + 1, k))         { e = 0; h = k;     D = D + 1; }
          else { w.setv(D, h, c);  c -= 1; }
        }
      else
        { m = (k == del) ? D : D - 2;
          if (c < w.v(m, k - 1))               { e = (k == del) ? 2 : -1; h = k - 1; D = m; }
          else if (c == w.v(D - 1, k))         { e = 0; h = k; D = D - 1; }
          else { w.setv(D, h, c);  c -= 1; }
        }
    }
  return h;
}

/***** one segment: waves, then the script (iter_np, align.c:4892-5261) or the mid point (middle_np,
 * :5263-5573).  Returns 0 = done, 1 = the waves do not fit the cell storage, 2 = D > dmax (align.c:4966).
 * KIND 0: out receives the nout script values, dist = D + |del|.  KIND 1: out[0], out[1] = the A and B
 * offsets of the mid point.                                                                          */
template <int MODE, int KIND, class Cells, class Snake>
__device__ __forceinline__ int expand_segment(Cells &w, const Snake &sn, const ByteBases &bb, const TraceSeg &g,
                                              int M, int N, int dmax, int *out, int &nout, int &dist)
{ const int del = M - N;
  int low = del < 0 ? del : 0, hgh = del < 0 ? 0 : del;
  int posl = -dmax, posh = dmax;
  if (g.flags & 2u)                                       /* both reads are one buffer (align.c:4933-4951) */
    { const int off = g.b0 - g.a0;
      if (off < 0) { if (off + 1 > posl) posl = off + 1; }
      else         { if (off - 1 < posh) posh = off - 1; }
    }
  if (!w.fits(0))
    return 1;
  for (int k = low - 1; k <= hgh + 1; k++)
    { w.setv(-2, k, -2);
      w.setv(-1, k, -2);
    }
  w.setv(-1, 0, -1);
  low += 1;
  hgh -= 1;

  int D;
  for (D = 0; ; D++)
    { if (D > dmax) return 2;
      if (!w.fits(D)) return 1;
      if ((D & 1) == 0)
        { if (low > posl) low -= 1;
          if (hgh < posh) hgh += 1;
        }
      w.setv(D, hgh + 1, -2);
      w.setv(D, low - 1, -2);
      int j = -2, code;
      for (int k = hgh; k > del; k--)
        { j = choose(w.va(D - 2, k - 1), w.va(D - 1, k) + 1, j + 1, -1, 4, code);
          j = sn.slide(k, j, min(N, M - k));
          w.put_above(D, k, j, code);
        }
      j = -2;
      for (int k = low; k < del; k++)
        { j = choose(j, w.vb(D - 1, k) + 1, w.vb(D - 2, k + 1) + 1, 2, 1, code);
          j = sn.slide(k, j, min(N, M - k));
          w.put_below(D, k, j, code);
        }
      j = choose(j, w.va(D - 1, del) + 1, w.va(D, del + 1) + 1, 2, 4, code);
      j = sn.slide(del, j, N);
      w.put_del(D, j, code);
      if (j >= N) break;
    }

  int e, h, m, c = N, k = del;
  if (KIND == 1)
    { /* ceil((D + |del|) / 2) edges back from the end: the furthest point of that cell is the mid point */
      int d = D + (del < 0 ? -del : del);
      for (const int f = d / 2; d > f; d--)
        { e = w.h(D, k);
          k = back_edge<MODE>(w, bb, del, D, k, e, c);
        }
      const int x = w.v(D, k);
      out[0] = g.a0 + k + x;
      out[1] = g.b0 + x;
      nout = 0;
      dist = 0;
      return 0;
    }

  /* predecessor links -> successor links, from (D, del) back to (0, 0) (align.c:5042-5215) */
  w.seth(0, 0, 3);
  e = w.h(D, k);
  w.seth(D, k, 3);
  while (e != 3)
    { h = back_edge<MODE>(w, bb, del, D, k, e, c);
      m = w.h(D, h);
      w.seth(D, h, e);
      e = m;
      k = h;
    }

  /* forward again: one script value per indel (align.c:5217-5256) */
  int n = 0, jstart = 0;
  k = D = 0;
  e = w.h(0, 0);
  while (e != 3)
    { h = k - e;
      /* the furthest point of the path's cell: kept, or found again by sliding from where the path
         entered the diagonal (the predecessor's point, one further unless the move kept the B index) */
      c = Cells::recompute ? sn.slide(k, jstart, min(N, M - k)) : w.v(D, k);
      jstart = (e == -1 || e == 2) ? c : c + 1;
      if (e > 1) h += 3;
      else if (e == 0) D += 1;
      else D += 2;
      if (h > k)      out[n++] = g.b0 + c + 1;
      else if (h < k) out[n++] = -(g.a0 + c + k + 1);
      k = h;
      e = w.h(D, h);
    }
  nout = n;
  dist = D + (del < 0 ? -del : del);
  return 0;
}

/* any segment, one lane each, cells in per-lane stripes: the launch for what trace_waves_slots defers */
template <int MODE, int KIND>
__global__ __launch_bounds__(TP_THREADS)
void trace_waves(TraceArgs t)
{ const u32 tid = blockIdx.x * TP_THREADS + threadIdx.x;
  const u32 nthreads = gridDim.x * TP_THREADS;
  Stripe w;
  w.vf = t.vf + (size_t) tid * t.cap;
  w.hf = t.hf + (size_t) tid * t.cap;
  w.cap = t.cap;

  for (u32 it = tid; it < t.nwork; it += nthreads)
    { const u32 s = t.list ? t.list[it] : it;
      const TraceSeg g = t.segs[s];
      int n = 0, dist = 0;
      if (!(g.flags & 4u))
        { const int M = (int) (g.mn & 0xffffu), N = (int) (g.mn >> 16);
          const int dmax = (int) ((g.flags >> 8) & 0xffffu);
          ByteBases bb;
          bb.comp = (int) (g.flags & 1u);
          bb.sgn = bb.comp ? -1 : 1;
          bb.A = t.abases + g.apos;
          bb.B = t.bbases + g.bpos;
          w.shape(M - N);
          const int status = expand_segment<MODE, KIND>(w, bb, bb, g, M, N, dmax,
                                                        KIND ? t.mid + 2 * (size_t) s : t.stage + g.stage, n, dist);
          if (status == 1)
            { const u32 o = atomicAdd(t.nover, 1u);
              if (o < t.over_cap) t.over[o] = s;
              atomicMax(t.need, w.rows_needed(dmax));     /* rows -2 .. dmax of this segment */
            }
          else if (status == 2)
            atomicOr(t.err, DAMAR_TRACE_ERR_ALIGN);       /* "Bad alignment between trace points" */
          if (status) { n = 0;  dist = 0; }
        }
      t.count[s] = (u32) n;
      t.dist[s]  = dist;
    }
}

/* the common case: 64 consecutive segments per wavefront, cells in the wavefront's slot area, forward
   snakes on 2-bit copies of the segments in LDS.  Defers segments with more than SLOT_MAXB bases a side,
   waves wider than SLOT_SS diagonals either side of del or more than SLOT_ROWS - 3 waves, and the
   one-buffer case. */
template <int MODE, int KIND>
__global__ __launch_bounds__(TP_THREADS)
void trace_waves_slots(TraceArgs t)
{ __shared__ u32 lds[2 * SLOT_WORDS * 64];
  const int lane = (int) threadIdx.x;
  const size_t area = (size_t) SLOT_ROWS * SLOT_RS * 64;
  const int ring = (MODE == 0 && KIND == 0);               /* the mid point is read from an older row: keep them all */
  Slots<ring> w;
  w.vf = (u8 *) t.vf + (size_t) blockIdx.x * (ring ? (size_t) 4 * SLOT_RS * 64 : area);
  w.hf = t.hf + (size_t) blockIdx.x * area;
  w.lane = (u32) lane;
  w.rows = min((int) t.cap, SLOT_ROWS);
  u32 *const la = lds + lane, *const lb = lds + SLOT_WORDS * 64 + lane;
  const u32 nbatch = (t.nwork + 63) / 64;

  for (;;)
    { /* batches are handed out in order (heaviest segments first), one atomic per wavefront */
      u32 batch = 0;
      if (lane == 0) batch = atomicAdd(t.next, 1u);
      batch = (u32) __builtin_amdgcn_readfirstlane((int) batch);
      if (batch >= nbatch) break;
      const u32 it = batch * 64 + (u32) lane;
      if (it >= t.nwork) continue;
      const u32 s = t.list ? t.list[it] : it;
      const TraceSeg g = t.segs[s];
      int n = 0, dist = 0;
      if (!(g.flags & 4u))
        { const int M = (int) (g.mn & 0xffffu), N = (int) (g.mn >> 16);
          const int dmax = (int) ((g.flags >> 8) & 0xffffu);
          int status = 1;
          if (M <= SLOT_MAXB && N <= SLOT_MAXB && !(g.flags & 2u))
            { ByteBases bb;
              bb.comp = (int) (g.flags & 1u);
              bb.sgn = bb.comp ? -1 : 1;
              bb.A = t.abases + g.apos;
              bb.B = t.bbases + g.bpos;
              LdsBases sn;
              sn.la = la;  sn.lb = lb;
              sn.oa = (int) (g.apos & 15u);
              { const u32 *src = t.apk + (g.apos >> 4);
                const int nw = ((sn.oa + M + 15) >> 4) + 1;
                for (int i = 0; i < nw; i++) la[i * 64] = src[i];
              }
              if (!bb.comp)
                { sn.ob = (int) (g.bpos & 15u);
                  const u32 *src = t.bpk + (g.bpos >> 4);
                  const int nw = ((sn.ob + N + 15) >> 4) + 1;
                  for (int i = 0; i < nw; i++) lb[i * 64] = src[i];
                }
              else
                { /* B'[j] = 3 - B[bpos - j]: word i of the copy = the 16 bases ending at bpos - 16 i,
                     order reversed (bit reversal, then the two bits of each base swapped back), complemented */
                  sn.ob = 0;
                  const int nw = ((N + 15) >> 4) + 1;
                  for (int i = 0; i < nw; i++)
                    { u32 x = pk_window(t.bpk, g.bpos - 16u * (u32) i - 15u + 16u * PK_PAD);
                      x = __builtin_bitreverse32(x);
                      x = ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
                      lb[i * 64] = ~x;
                    }
                }
              w.shape(M - N);
              status = expand_segment<MODE, KIND>(w, sn, bb, g, M, N, dmax,
                                                  KIND ? t.mid + 2 * (size_t) s : t.stage + g.stage, n, dist);
            }
          if (status == 1)
            { const u32 o = atomicAdd(t.nover, 1u);       /* deferred to the launch with per-lane stripes */
              if (o < t.over_cap) t.over[o] = s;
              Stripe st;
              st.shape(M - N);
              atomicMax(t.need, st.rows_needed(dmax));
            }
          else if (status == 2)
            atomicOr(t.err, DAMAR_TRACE_ERR_ALIGN);
          if (status) { n = 0;  dist = 0; }
        }
      t.count[s] = (u32) n;
      t.dist[s]  = dist;
    }
}

/* mid != 0: the record has one piece more than trace-point segments, its pieces start at seg0 + r, and the
   distance of the piece that ends on the last mid point counts twice (align.c:5812-5822) */
__global__ __launch_bounds__(256)
void trace_gather(const TraceRecIn *__restrict__ recs, u32 nrecs, int mid, const u32 *__restrict__ count,
                  const int *__restrict__ dist, u32 *__restrict__ segoff, u32 *__restrict__ tlen, int *__restrict__ diffs)
{ const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrecs) return;
  const TraceRecIn in = recs[r];
  const int nseg = (in.tlen >= 2 ? in.tlen / 2 : 1) + (mid ? 1 : 0);
  const u32 s0 = in.seg0 + (mid ? r : 0u);
  u32 n = 0;
  int d = 0;
  for (int s = 0; s < nseg; s++)
    { segoff[s0 + s] = n;
      n += count[s0 + s];
      d += dist[s0 + s];
    }
  if (mid)
    d += dist[s0 + nseg - 2];
  tlen[r] = n;
  diffs[r] = d;
}

/* Compute_Trace_MID (align.c:5694-5830): the pieces between successive mid points.  Record r with n
   trace-point segments becomes n + 1 pieces at seg0 + r: start .. mid 0, mid 0 .. mid 1, ..., last mid .. end. */
__global__ __launch_bounds__(256)
void trace_mid_layout(const TraceRecIn *__restrict__ recs, u32 nrecs, const TraceSeg *__restrict__ segs,
                      const int *__restrict__ mid, DevBlock ablk, DevBlock bblk, TraceSeg *__restrict__ out,
                      u32 *__restrict__ key, u32 *__restrict__ val, u32 *__restrict__ err)
{ const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrecs) return;
  const TraceRecIn in = recs[r];
  const int nseg = in.tlen >= 2 ? in.tlen / 2 : 1;
  const u32 aoff = ablk.boff[in.aread];
  const u32 boff = bblk.boff[in.bread], blen = bblk.boff[in.bread + 1] - boff - 1;
  const int comp = (int) (in.flags & 1u);
  const bool dead = (segs[in.seg0].flags & 4u) != 0;
  int as = in.abpos, bs = in.bbpos;
  u32 so = in.stage0;
  bool bad = false;
  for (int i = 0; i <= nseg; i++)
    { int af, bf;
      if (i < nseg) { af = mid[2 * (size_t) (in.seg0 + i)];  bf = mid[2 * (size_t) (in.seg0 + i) + 1]; }
      else          { af = in.aepos;  bf = in.bepos; }
      TraceSeg g = {};
      g.rec = r;
      u32 own = 0;
      const int M = af - as, N = bf - bs;
      if (dead || M < 0 || N < 0 || M > 32000 || N > 32000)
        { g.flags = 4u;
          bad = bad || !dead;
        }
      else
        { const int del = M - N;
          const u32 cap = (u32) (in.dmax + (del < 0 ? -del : del));
          if (so + cap > in.stage0 + in.slots)
            { g.flags = 4u;
              bad = true;
            }
          else
            { g.apos = aoff + (u32) as;
              g.bpos = comp ? boff + blen - 1 - (u32) bs : boff + (u32) bs;
              g.a0 = as;  g.b0 = bs;
              g.mn = (u32) M | ((u32) N << 16);
              /* about half the differences of the two segments the piece overlaps */
              const u32 d0 = i > 0 ? segs[in.seg0 + i - 1].flags >> 24 : 0u, d1 = i < nseg ? segs[in.seg0 + i].flags >> 24 : 0u;
              own = min((d0 + d1 + 1) / 2, 255u);
              g.flags = (in.flags & 3u) | ((u32) min(in.dmax, 65535) << 8) | (own << 24);
              g.stage = so;
              so += cap;
            }
        }
      out[in.seg0 + r + i] = g;
      key[in.seg0 + r + i] = 255u - own;
      val[in.seg0 + r + i] = in.seg0 + r + i;
      as = af;
      bs = bf;
    }
  if (bad)
    atomicOr(err, DAMAR_TRACE_ERR_INTERNAL);
}

__global__ __launch_bounds__(256)
void trace_pack(const TraceSeg *__restrict__ segs, u32 nsegs, const u32 *__restrict__ count,
                const u32 *__restrict__ segoff, const u32 *__restrict__ recoff, const int *__restrict__ stage,
                int *__restrict__ script)
{ const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsegs) return;
  const TraceSeg g = segs[s];
  const int *src = stage + g.stage;
  int *dst = script + recoff[g.rec] + segoff[s];
  const u32 n = count[s];
  for (u32 i = 0; i < n; i++)
    dst[i] = src[i];
}

void damar_launch_trace_layout(const TraceRecIn *recs, u32 nrecs, const void *pts, int tbytes, int tspace,
                               const DevBlock *ablk, const DevBlock *bblk, TraceSeg *segs, u32 *key, u32 *val, u32 *err,
                               hipStream_t st)
{ if (nrecs == 0) return;
  if (tbytes == 1)
    hipLaunchKernelGGL(trace_layout<u8>, dim3((nrecs + 255) / 256), dim3(256), 0, st, recs, nrecs, (const u8 *) pts, tspace,
                       *ablk, *bblk, segs, key, val, err);
  else
    hipLaunchKernelGGL(trace_layout<u16>, dim3((nrecs + 255) / 256), dim3(256), 0, st, recs, nrecs, (const u16 *) pts, tspace,
                       *ablk, *bblk, segs, key, val, err);
}

#define LAUNCH_BY_MODE(kern)                                                                            \
  do { if (kind == 0)                                                                                    \
         { if (mode == 0)      hipLaunchKernelGGL((kern<0, 0>),  dim3(nblocks), dim3(TP_THREADS), 0, st, *t);   \
           else if (mode > 0)  hipLaunchKernelGGL((kern<1, 0>),  dim3(nblocks), dim3(TP_THREADS), 0, st, *t);   \
           else                hipLaunchKernelGGL((kern<-1, 0>), dim3(nblocks), dim3(TP_THREADS), 0, st, *t);   \
         }                                                                                               \
       else                                                                                              \
         { if (mode == 0)      hipLaunchKernelGGL((kern<0, 1>),  dim3(nblocks), dim3(TP_THREADS), 0, st, *t);   \
           else if (mode > 0)  hipLaunchKernelGGL((kern<1, 1>),  dim3(nblocks), dim3(TP_THREADS), 0, st, *t);   \
           else                hipLaunchKernelGGL((kern<-1, 1>), dim3(nblocks), dim3(TP_THREADS), 0, st, *t);   \
         }                                                                                               \
     } while (0)

/* kind 0: scripts into t->stage / count / dist; kind 1: mid points into t->mid (2 ints per segment) */
void damar_launch_trace_waves(const TraceArgs *t, int mode, int kind, u32 nblocks, hipStream_t st)
{ if (t->nwork == 0) return;
  LAUNCH_BY_MODE(trace_waves);
}

size_t damar_trace_slot_area_cells(void) { return (size_t) SLOT_ROWS * SLOT_RS * 64; }
/* bytes of furthest points per workgroup: four live rows for GREEDIEST scripts, every row otherwise */
size_t damar_trace_slot_vf_bytes(int mode, int kind) { return (mode == 0 && kind == 0) ? (size_t) 4 * SLOT_RS * 64 : (size_t) SLOT_ROWS * SLOT_RS * 64; }

/* t->vf / t->hf: nblocks areas of damar_trace_slot_area_cells() cells; t->list must be NULL */
void damar_launch_trace_waves_slots(const TraceArgs *t, int mode, int kind, u32 nblocks, hipStream_t st)
{ if (t->nwork == 0) return;
  LAUNCH_BY_MODE(trace_waves_slots);
}

void damar_launch_trace_gather(const TraceRecIn *recs, u32 nrecs, int mid, const u32 *count, const int *dist, u32 *segoff,
                               u32 *tlen, int *diffs, hipStream_t st)
{ if (nrecs == 0) return;
  hipLaunchKernelGGL(trace_gather, dim3((nrecs + 255) / 256), dim3(256), 0, st, recs, nrecs, mid, count, dist, segoff, tlen, diffs);
}

void damar_launch_trace_mid_layout(const TraceRecIn *recs, u32 nrecs, const TraceSeg *segs, const int *mid,
                                   const DevBlock *ablk, const DevBlock *bblk, TraceSeg *out, u32 *key, u32 *val, u32 *err,
                                   hipStream_t st)
{ if (nrecs == 0) return;
  hipLaunchKernelGGL(trace_mid_layout, dim3((nrecs + 255) / 256), dim3(256), 0, st, recs, nrecs, segs, mid, *ablk, *bblk, out,
                     key, val, err);
}

void damar_launch_trace_pack(const TraceSeg *segs, u32 nsegs, const u32 *count, const u32 *segoff, const u32 *recoff,
                             const int *stage, int *script, hipStream_t st)
{ if (nsegs == 0) return;
  hipLaunchKernelGGL(trace_pack, dim3((nsegs + 255) / 256), dim3(256), 0, st, segs, nsegs, count, segoff, recoff, stage, script);
}

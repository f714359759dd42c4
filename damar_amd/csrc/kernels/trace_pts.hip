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
                  DevBlock ablk, DevBlock bblk, TraceSeg *__restrict__ segs, u32 *__restrict__ err)
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
      g.flags = (in.flags & 3u) | ((u32) in.dmax << 8);
      g.stage = so;
      g.rec = r;
      segs[in.seg0 + s] = g;
      so += (u32) (in.dmax + (del < 0 ? -del : del));
      ab = ae;
      bb = be;
    }
  if (s < nseg)
    { atomicOr(err, DAMAR_TRACE_ERR_POINTS);
      for (s = 0; s < nseg; s++)                          /* void segments: the host stops on the flag */
        { TraceSeg g = {};  g.rec = r;  g.flags = 4u;  segs[in.seg0 + s] = g; }
    }
}

/* packed row layout of one segment's waves: rows -2, -1, 0, 1, ... ; row D >= 0 spans the diagonals
   low0 - D/2 - 1 .. hgh0 + D/2 + 1 (one sentinel either side), rows -2 and -1 like row 0 */
struct RowMap
{ int w0, low0;
  __device__ __forceinline__ int start(int D) const
  { const int m = D > 0 ? D >> 1 : 0, q = D > 0 ? (D - 1) >> 1 : 0;
    return (D + 2) * w0 + 2 * m * q;
  }
  __device__ __forceinline__ int at(int D, int k) const
  { const int m = D > 0 ? D >> 1 : 0;
    return start(D) + k - (low0 - m - 1);
  }
};

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

template <int MODE>
__global__ __launch_bounds__(TP_THREADS)
void trace_waves(TraceArgs t)
{ const u32 tid = blockIdx.x * TP_THREADS + threadIdx.x;
  const u32 nthreads = gridDim.x * TP_THREADS;
  short       *const vf = t.vf + (size_t) tid * t.cap;
  signed char *const hf = t.hf + (size_t) tid * t.cap;
  const u8 *const abase = t.abases;
  const u8 *const bbase = t.bbases;

  for (u32 it = tid; it < t.nwork; it += nthreads)
    { const u32 s = t.list ? t.list[it] : it;
      const TraceSeg g = t.segs[s];
      if (g.flags & 4u)
        { t.count[s] = 0;  t.dist[s] = 0;
          continue;
        }
      const int M = (int) (g.mn & 0xffffu), N = (int) (g.mn >> 16);
      const int del = M - N;
      const int dmax = (int) (g.flags >> 8);
      const int comp = (int) (g.flags & 1u);
      const int sgn = comp ? -1 : 1;
      const u8 *const A = abase + g.apos;
      const u8 *const B = bbase + g.bpos;
#define BV(j)  ({ const int v_ = B[sgn * (j)]; (comp && v_ < 4) ? 3 - v_ : v_; })
      RowMap rm;
      rm.w0 = (del < 0 ? -del : del) + 3;
      rm.low0 = del < 0 ? del : 0;
      int low = rm.low0, hgh = del < 0 ? 0 : del;
      int posl = -dmax, posh = dmax;
      if (g.flags & 2u)                                   /* both reads are one buffer (align.c:4933-4951) */
        { const int off = g.b0 - g.a0;
          if (off < 0) { if (off + 1 > posl) posl = off + 1; }
          else         { if (off - 1 < posh) posh = off - 1; }
        }
      int D = 0, status = 0;                              /* 1 = scratch exhausted, 2 = D > dmax */
      if ((u32) rm.start(1) > t.cap)
        status = 1;
      else
        { for (int k = low - 1; k <= hgh + 1; k++)
            { vf[rm.at(-2, k)] = -2;
              vf[rm.at(-1, k)] = -2;
            }
          vf[rm.at(-1, 0)] = -1;
        }
      low += 1;
      hgh -= 1;

      for (D = 0; status == 0; D++)
        { if (D > dmax) { status = 2; break; }
          if ((u32) rm.start(D + 1) > t.cap) { status = 1; break; }
          if ((D & 1) == 0)
            { if (low > posl) low -= 1;
              if (hgh < posh) hgh += 1;
            }
          const int r0 = rm.at(D, 0), r1 = rm.at(D - 1, 0), r2 = rm.at(D - 2, 0);   /* column of k = 0 */
          vf[r0 + hgh + 1] = -2;
          vf[r0 + low - 1] = -2;
          int j = -2, code;
          for (int k = hgh; k > del; k--)
            { j = choose(vf[r2 + k - 1], vf[r1 + k] + 1, j + 1, -1, 4, code);
              hf[r0 + k] = (signed char) code;
              const int lim = min(N, M - k);
              const u8 *a = A + k;
              while (j < lim && BV(j) == a[j]) j++;
              vf[r0 + k] = (short) j;
            }
          j = -2;
          for (int k = low; k < del; k++)
            { j = choose(j, vf[r1 + k] + 1, vf[r2 + k + 1] + 1, 2, 1, code);
              hf[r0 + k] = (signed char) code;
              const int lim = min(N, M - k);
              const u8 *a = A + k;
              while (j < lim && BV(j) == a[j]) j++;
              vf[r0 + k] = (short) j;
            }
          j = choose(j, vf[r1 + del] + 1, vf[r0 + del + 1] + 1, 2, 4, code);
          hf[r0 + del] = (signed char) code;
          { const u8 *a = A + del;
            while (j < N && BV(j) == a[j]) j++;
          }
          vf[r0 + del] = (short) j;
          if (j >= N) break;
        }
      if (status == 1)
        { const u32 o = atomicAdd(t.nover, 1u);           /* deferred to the launch with large stripes */
          if (o < t.over_cap) t.over[o] = s;
          atomicMax(t.need, (u32) rm.start(dmax + 1));    /* rows -2 .. dmax of this segment */
          t.count[s] = 0;  t.dist[s] = 0;
          continue;
        }
      if (status == 2)
        { atomicOr(t.err, DAMAR_TRACE_ERR_ALIGN);         /* align.c:4966: "Bad alignment between trace points" */
          t.count[s] = 0;  t.dist[s] = 0;
          continue;
        }

      /* predecessor links -> successor links, from (D, del) back to (0, 0) (align.c:5042-5215) */
      { int e, h, m, c = N, k = del;
        hf[rm.at(0, 0)] = 3;
        { const int x = rm.at(D, k);  e = hf[x];  hf[x] = 3; }
        while (e != 3)
          { h = k + e;
            if (e > 1) h -= 3;
            else if (e == 0) D -= 1;
            else D -= 2;
            if (MODE == 1 && h < k)
              { const u8 *a = A + k;
                m = k < 0 ? -k : 0;
                const int x = rm.at(D, h);
                if (vf[x] <= c) c = vf[x] - 1;
                while (c >= m && a[c] == BV(c)) c -= 1;
                if (e == -1)
                  { if (c <= vf[rm.at(D + 2, k + 1)])      { e = 4; h = k + 1; D = D + 2; }
                    else if (c == vf[rm.at(D + 1, k)])     { e = 0; h = k;     D = D + 1; }
                    else vf[x] = (short) (c + 1);
                  }
                else
                  { m = (k == del) ? D : D - 2;
                    if (c <= vf[rm.at(m, k + 1)])          { e = (k == del) ? 4 : 1; h = k + 1; D = m; }
                    else if (c == vf[rm.at(D - 1, k)])     { e = 0; h = k; D = D - 1; }
                    else vf[x] = (short) (c + 1);
                  }
              }
            else if (MODE == -1 && h > k)
              { const u8 *a = A + k;
                m = k < 0 ? -k : 0;
                const int x = rm.at(D, h);
                if (vf[x] < c) c = vf[x];
                while (c >= m && a[c] == BV(c)) c -= 1;
                if (e == 1)
                  { if (c < vf[rm.at(D + 2, k - 1)])       { e = 2; h = k - 1; D = D + 2; }
                    else if (c == vf[rm.at(D + 1, k)])     { e = 0; h = k;     D = D + 1; }
                    else { vf[x] = (short) c;  c -= 1; }
                  }
                else
                  { m = (k == del) ? D : D - 2;
                    if (c < vf[rm.at(m, k - 1)])           { e = (k == del) ? 2 : -1; h = k - 1; D = m; }
                    else if (c == vf[rm.at(D - 1, k)])     { e = 0; h = k; D = D - 1; }
                    else { vf[x] = (short) c;  c -= 1; }
                  }
              }
            { const int x = rm.at(D, h);
              m = hf[x];
              hf[x] = (signed char) e;
              e = m;
            }
            k = h;
          }

        /* forward again: one script value per indel (align.c:5217-5256) */
        int *out = t.stage + g.stage;
        int n = 0;
        k = D = 0;
        e = hf[rm.at(0, 0)];
        while (e != 3)
          { h = k - e;
            c = vf[rm.at(D, k)];
            if (e > 1) h += 3;
            else if (e == 0) D += 1;
            else D += 2;
            if (h > k)      out[n++] = g.b0 + c + 1;
            else if (h < k) out[n++] = -(g.a0 + c + k + 1);
            k = h;
            e = hf[rm.at(D, h)];
          }
        t.count[s] = (u32) n;
        t.dist[s]  = D + (del < 0 ? -del : del);
      }
#undef BV
    }
}

__global__ __launch_bounds__(256)
void trace_gather(const TraceRecIn *__restrict__ recs, u32 nrecs, const u32 *__restrict__ count,
                  const int *__restrict__ dist, u32 *__restrict__ segoff, u32 *__restrict__ tlen, int *__restrict__ diffs)
{ const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrecs) return;
  const TraceRecIn in = recs[r];
  const int nseg = in.tlen >= 2 ? in.tlen / 2 : 1;
  u32 n = 0;
  int d = 0;
  for (int s = 0; s < nseg; s++)
    { segoff[in.seg0 + s] = n;
      n += count[in.seg0 + s];
      d += dist[in.seg0 + s];
    }
  tlen[r] = n;
  diffs[r] = d;
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
                               const DevBlock *ablk, const DevBlock *bblk, TraceSeg *segs, u32 *err, hipStream_t st)
{ if (nrecs == 0) return;
  if (tbytes == 1)
    hipLaunchKernelGGL(trace_layout<u8>, dim3((nrecs + 255) / 256), dim3(256), 0, st, recs, nrecs, (const u8 *) pts, tspace,
                       *ablk, *bblk, segs, err);
  else
    hipLaunchKernelGGL(trace_layout<u16>, dim3((nrecs + 255) / 256), dim3(256), 0, st, recs, nrecs, (const u16 *) pts, tspace,
                       *ablk, *bblk, segs, err);
}

void damar_launch_trace_waves(const TraceArgs *t, int mode, u32 nblocks, hipStream_t st)
{ if (t->nwork == 0) return;
  if (mode == 0)      hipLaunchKernelGGL(trace_waves<0>,  dim3(nblocks), dim3(TP_THREADS), 0, st, *t);
  else if (mode > 0)  hipLaunchKernelGGL(trace_waves<1>,  dim3(nblocks), dim3(TP_THREADS), 0, st, *t);
  else                hipLaunchKernelGGL(trace_waves<-1>, dim3(nblocks), dim3(TP_THREADS), 0, st, *t);
}

void damar_launch_trace_gather(const TraceRecIn *recs, u32 nrecs, const u32 *count, const int *dist, u32 *segoff,
                               u32 *tlen, int *diffs, hipStream_t st)
{ if (nrecs == 0) return;
  hipLaunchKernelGGL(trace_gather, dim3((nrecs + 255) / 256), dim3(256), 0, st, recs, nrecs, count, dist, segoff, tlen, diffs);
}

void damar_launch_trace_pack(const TraceSeg *segs, u32 nsegs, const u32 *count, const u32 *segoff, const u32 *recoff,
                             const int *stage, int *script, hipStream_t st)
{ if (nsegs == 0) return;
  hipLaunchKernelGGL(trace_pack, dim3((nsegs + 255) / 256), dim3(256), 0, st, segs, nsegs, count, segoff, recoff, stage, script);
}

/* bridge.c -- ORACLE (test infrastructure): bridging of two narrowly parallel local alignments of one read pair.
 *
 * A restatement that FOLLOWS the reference function by function, so that it can be checked against it line by line:
 * filter.c:1376-1454 MapToTPAbove/Below + Check_Bridge, :1456-1571 Compute_Bridge_Path, :1747-1802 Bridge (called from
 * :1950-2059), and the exact realignment Compute_Alignment(DIFF_TRACE) = Myers' O(ND) divide and conquer with
 * per-trace-point accumulation: align.c:4327-4495 split_nd, :4497-4651 trace_nd, :4734-4869.
 * Independent of the product's own bridge code (damar_amd/csrc/host/bridge.c, written from the algorithm with a structure
 * of its own): the two meet only in the .las files the tests compare.  Never linked into libdamar_hip.so.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../damar_amd/csrc/host/damar_host.h"

typedef struct
{ int        *fwd, *rev;      /* furthest-y per diagonal, forward and reverse waves     */
  uint16     *tp;             /* tp[2*i] diffs, tp[2*i+1] b-length of A-segment i        */
  const char *abase;          /* coordinate origin of A for trace-point numbering        */
  int         ts;
} NdWork;

/* Middle snake of A[0..M) x B[0..N): returns the edit distance D and a point (x,y) on an
 * optimal path that splits it into ceil(D/2) and floor(D/2) differences (align.c:4327-4495). */
static int nd_split(const char *A, int M, const char *B, int N, NdWork *w, int *px, int *py)
{ int *VF = w->fwd, *VB = w->rev;
  int  flow, blow, bhgh, D, x, y;
  const char *a;

  y = 0;
  if (N < M)
    while (y < N && B[y] == A[y]) y += 1;
  else
    { while (y < M && B[y] == A[y]) y += 1;
      if (y >= M && N == M)
        { *px = *py = M;
          return 0;
        }
    }
  flow = 0;
  VF[0] = y;
  VF[-1] = -2;

  x = N - M;
  a = A - x;
  y = N - 1;
  if (N > M)
    while (y >= x && B[y] == a[y]) y -= 1;
  else
    while (y >= 0 && B[y] == a[y]) y -= 1;
  blow = bhgh = -x;
  VB += x;
  VB[blow] = y;
  VB[blow - 1] = N + 1;

  for (D = 1; ; D++)
    { int k, r, am, ac, ap;

      /* forward wave D */
      flow -= 1;
      am = ac = VF[flow - 1] = -2;
      a = A + D;
      x = M - D;
      for (k = D; k >= flow; k--)
        { ap = ac;
          ac = am + 1;
          am = VF[k - 1];
          if (ac < am)
            y = (ap < am) ? am : ap;
          else
            y = (ap < ac) ? ac : ap;
          if (blow <= k && k <= bhgh)
            { r = VB[k];
              if (y > r)
                { if (ap > r)      y = ap;
                  else if (ac > r) y = ac;
                  else             y = r + 1;
                  *px = k + y;
                  *py = y;
                  return 2 * D - 1;
                }
            }
          if (N < x)
            while (y < N && B[y] == a[y]) y += 1;
          else
            while (y < x && B[y] == a[y]) y += 1;
          VF[k] = y;
          a -= 1;
          x += 1;
        }

      /* reverse wave D */
      bhgh += 1;
      blow -= 1;
      am = ac = VB[blow - 1] = N + 1;
      a = A + bhgh;
      x = -bhgh;
      for (k = bhgh; k >= blow; k--)
        { ap = ac + 1;
          ac = am;
          am = VB[k - 1];
          if (ac > am)
            y = (ap > am) ? am : ap;
          else
            y = (ap > ac) ? ac : ap;
          if (flow <= k && k <= D)
            { r = VF[k];
              if (y <= r)
                { if (ap <= r)      y = ap;
                  else if (ac <= r) y = ac;
                  else              y = r;
                  *px = k + y;
                  *py = y;
                  return 2 * D;
                }
            }
          y -= 1;
          if (x > 0)
            while (y >= x && B[y] == a[y]) y -= 1;
          else
            while (y >= 0 && B[y] == a[y]) y -= 1;
          VB[k] = y;
          a -= 1;
          x += 1;
        }
    }
}

/* add `len` B-length (or diffs) spread over the A trace segments starting at A position u0 */
static void spread(NdWork *w, int slot, int u0, int len)
{ int v = u0 / w->ts, u = (v + 1) * w->ts - u0;
  for (v <<= 1; len > 0; len -= u, u = w->ts)
    { if (u > len) u = len;
      w->tp[v + slot] += (uint16) u;
      v += 2;
    }
}

/* align.c:4497-4651: exact alignment of A[0..M) vs B[0..N), accumulated per trace segment */
static int nd_trace(const char *A, int M, const char *B, int N, NdWork *w)
{ int x, y, D, s;
  const int ts = w->ts;

  if (M <= 0)
    { y = (((int) (A - w->abase)) / ts) << 1;
      w->tp[y]     += (uint16) N;
      w->tp[y + 1] += (uint16) N;
      return N;
    }
  if (N <= 0)
    { spread(w, 0, (int) (A - w->abase), M);
      return M;
    }

  D = nd_split(A, M, B, N, w, &x, &y);
  if (D > 1)
    { s = (int) (A - w->abase);
      if ((s / ts + 1) * ts - s >= x)
        { s = (s / ts) << 1;
          w->tp[s]     += (uint16) ((D + 1) / 2);
          w->tp[s + 1] += (uint16) y;
        }
      else
        nd_trace(A, x, B, y, w);

      s = (int) ((A + x) - w->abase);
      if ((s / ts + 1) * ts - s >= M - x)
        { s = (s / ts) << 1;
          w->tp[s]     += (uint16) (D / 2);
          w->tp[s + 1] += (uint16) (N - y);
        }
      else
        nd_trace(A + x, M - x, B + y, N - y, w);
    }
  else
    { s = (D == 0 || M < N) ? x : x - 1;
      if (s > 0)
        spread(w, 1, (int) (A - w->abase), s);
      if (D == 0)
        return D;
      if (M < N)
        y = (((int) ((A + x) - w->abase)) / ts) << 1;
      else
        y = (((int) ((A + (x - 1)) - w->abase)) / ts) << 1;
      w->tp[y] += 1;
      if (M <= N)
        w->tp[y + 1] += 1;
      s = M - x;
      if (s > 0)
        spread(w, 1, (int) ((A + x) - w->abase), s);
    }
  return D;
}

typedef struct
{ int abpos, bbpos, aepos, bepos, diffs, tlen;
  uint16 *trace;
} BPath;

/* Compute_Alignment(align, work, DIFF_TRACE, ts) (align.c:4734-4869) for the box in p */
static void diff_trace(const char *aseq, const char *bseq, BPath *p, int ts, int **vec, int *vmax,
                       uint16 **tr, int *tmax)
{ int asub = p->aepos - p->abpos, bsub = p->bepos - p->bbpos;
  int big = (asub > bsub) ? asub : bsub;
  int n = 2 * (((p->aepos + (ts - 1)) / ts - p->abpos / ts) + 1);
  NdWork w;
  int i;

  if (4 * big + 6 > *vmax)
    { *vmax = (int) (1.2 * (4 * big + 6)) + 10000;
      *vec = (int *) realloc(*vec, sizeof(int) * (size_t) *vmax);
    }
  if (n > *tmax)
    { *tmax = (int) (1.2 * n) + 1000;
      *tr = (uint16 *) realloc(*tr, sizeof(uint16) * (size_t) *tmax);
    }
  if (*vec == NULL || *tr == NULL)
    { fprintf(stderr, "damar: out of memory (bridge)\n");
      exit(1);
    }
  w.fwd = *vec + (big + 1);
  w.rev = w.fwd + (2 * big + 3);
  w.abase = aseq;
  w.ts = ts;
  for (i = 0; i < n; i++)
    (*tr)[i] = 0;
  w.tp = *tr - 2 * (p->abpos / ts);
  p->diffs = nd_trace(aseq + p->abpos, asub, bseq + p->bbpos, bsub, &w);
  if ((*tr)[n - 1] != 0)              /* inserts that landed exactly on the last boundary */
    { (*tr)[n - 3] += (*tr)[n - 1];
      (*tr)[n - 4] += (*tr)[n - 2];
    }
  p->tlen = n - 2;
  p->trace = *tr;
}

/* filter.c:1376-1408 / 1410-1442: first (last) trace point of `path` at or beyond (before) *x
 * in A (isA) or B coordinates; returns the other coordinate and snaps *x to the trace point */
static int tp_above(const damar_path *path, int *x, int isA, const damar_tpool *tp, int ts)
{ const uint16 *trace = tp->val + path->toff;
  int a = (path->abpos / ts) * ts, b = path->bbpos, i;
  for (i = 1; i < path->tlen; i += 2)
    { a += ts;
      b += trace[i];
      if (a > path->aepos) a = path->aepos;
      if (isA) { if (a >= *x) { *x = a; return b; } }
      else     { if (b >= *x) { *x = b; return a; } }
    }
  if (isA) { *x = a; return b; }
  *x = b;
  return a;
}

static int tp_below(const damar_path *path, int *x, int isA, const damar_tpool *tp, int ts)
{ const uint16 *trace = tp->val + path->toff;
  int a = ((path->aepos + (ts - 1)) / ts) * ts, b = path->bepos, i;
  for (i = path->tlen - 1; i >= 0; i -= 2)
    { a -= ts;
      b -= trace[i];
      if (a < path->abpos) a = path->abpos;
      if (isA) { if (a <= *x) { *x = a; return b; } }
      else     { if (b <= *x) { *x = b; return a; } }
    }
  if (isA) { *x = a; return b; }
  *x = b;
  return a;
}

/* work buffers of the realignment, per thread: the host tail runs read-pair ranges on several threads */
static __thread int *g_vec = NULL;   static __thread int g_vmax = 0;
static __thread uint16 *g_tr = NULL; static __thread int g_tmax = 0;

/* filter.c:1456-1571 without its debug branches.  The realigned box is left in *box (trace
 * in the static work buffer, valid until the next call). */
static void bridge_path(const damar_path *p1, const damar_path *p2, const char *aseq0, int alen0,
                        const char *bseq, int blen, int comp, int aovl, int bovl,
                        const damar_tpool *tp, int ts, BPath *box)
{ int ain, aout, bin, bout, boff = 0, q;
  const char *aseq = aseq0;
  int alen = alen0;

  if (bovl > aovl)
    { bin  = p2->bbpos;
      bout = p1->bepos;
      ain  = tp_below(p1, &bin, 0, tp, ts);
      aout = tp_above(p2, &bout, 0, tp, ts);
    }
  else
    { ain  = p2->abpos;
      aout = p1->aepos;
      bin  = tp_below(p1, &ain, 1, tp, ts);
      bout = tp_above(p2, &aout, 1, tp, ts);
    }
  (void) bin; (void) bout;

  box->abpos = ain - 2 * ts;
  box->aepos = aout + 2 * ts;
  box->bbpos = tp_below(p1, &box->abpos, 1, tp, ts);
  box->bepos = tp_above(p2, &box->aepos, 1, tp, ts);

  if (comp)
    { boff = ts - box->aepos % ts;
      q = alen - box->abpos;  box->abpos = alen - box->aepos;  box->aepos = q;
      q = blen - box->bbpos;  box->bbpos = blen - box->bepos;  box->bepos = q;
      boff = boff - box->abpos % ts;
      aseq -= boff;
      box->abpos += boff;
      box->aepos += boff;
      alen += boff;
    }

  diff_trace(aseq, bseq, box, ts, &g_vec, &g_vmax, &g_tr, &g_tmax);

  if (comp)
    { uint16 *trk = box->trace;
      int i = 0, j = box->tlen - 2;
      while (i < j)
        { uint16 t = trk[i];     trk[i] = trk[j];         trk[j] = t;
          t = trk[i + 1];        trk[i + 1] = trk[j + 1]; trk[j + 1] = t;
          i += 2;
          j -= 2;
        }
      box->abpos -= boff;
      box->aepos -= boff;
      alen -= boff;
      q = alen - box->abpos;  box->abpos = alen - box->aepos;  box->aepos = q;
      q = blen - box->bbpos;  box->bbpos = blen - box->bepos;  box->bepos = q;
    }
}

/* filter.c:1444-1454: a bridged segment whose values do not fit a byte is refused */
static int bridge_too_big(const BPath *box, int ts)
{ int i;
  if (ts <= TRACE_XOVR)
    for (i = 0; i < box->tlen; i++)
      if (box->trace[i] > 250)
        return 1;
  return 0;
}

/* filter.c:1747-1802: p1 := p1[.. box.abpos] ++ box ++ p3[box.aepos ..] */
static void splice(damar_path *p1, const BPath *box, const damar_path *p3, damar_tpool *tp, int ts)
{ int k1 = 2 * ((box->abpos / ts) - (p1->abpos / ts));
  int k2 = (box->aepos == p3->aepos) ? p3->tlen : 2 * ((box->aepos / ts) - (p3->abpos / ts));
  int len = k1 + box->tlen + (p3->tlen - k2);
  int64 at;
  uint16 *dst;
  int n = 0, diff = 0, k;

  if (tp->top + len >= tp->max)
    { tp->max = (int64) (1.2 * (tp->top + len)) + 1000;
      tp->val = (uint16 *) realloc(tp->val, sizeof(uint16) * (size_t) tp->max);
      if (tp->val == NULL)
        { fprintf(stderr, "damar: out of memory (trace pool)\n");
          exit(1);
        }
    }
  at = tp->top;
  tp->top += len;
  dst = tp->val + at;
  for (k = 0; k < k1; k += 2)
    { dst[n++] = tp->val[p1->toff + k];
      dst[n++] = tp->val[p1->toff + k + 1];
      diff += tp->val[p1->toff + k];
    }
  for (k = 0; k < box->tlen; k += 2)
    { dst[n++] = box->trace[k];
      dst[n++] = box->trace[k + 1];
      diff += box->trace[k];
    }
  for (k = k2; k < p3->tlen; k += 2)
    { dst[n++] = tp->val[p3->toff + k];
      dst[n++] = tp->val[p3->toff + k + 1];
      diff += tp->val[p3->toff + k];
    }
  p1->aepos = p3->aepos;
  p1->bepos = p3->bepos;
  p1->diffs = diff;
  p1->toff  = at;
  p1->tlen  = n;
}

/* One candidate of the second loop of Handle_Redundancies (filter.c:1998-2057).  Returns
 * non-zero when the candidate was skipped (`continue` in the reference). */
int damar_bridge_pair(const damar_bridge_ctx *ctx, damar_path *jp, damar_path *kp,
                      damar_path *p1, damar_path *p2, damar_path *b1, damar_path *b2,
                      int aovl, int bovl, int comp, int ts, damar_tpool *tp,
                      damar_path *bm, int j)
{ BPath box;
  damar_path jback, kback;

  bridge_path(p1, p2, ctx->aseq, ctx->alen, ctx->bseq, ctx->blen, 0, aovl, bovl, tp, ts, &box);
  if (bridge_too_big(&box, ts))
    return 1;
  jback = *jp;
  kback = *kp;
  splice(p1, &box, p2, tp, ts);
  *jp = *p1;
  kp->abpos = -1;
  __atomic_fetch_add(&damar_stat_bridges, 1, __ATOMIC_RELAXED);

  if (b1 != NULL)
    { /* the B view: roles of the sequences swapped (filter.c:1825-1829, 2025) */
      bridge_path(b1, b2, ctx->bseq, ctx->blen, ctx->aseq, ctx->alen, comp, bovl, aovl, tp, ts, &box);
      if (bridge_too_big(&box, ts))
        { *jp = jback;
          *kp = kback;
          return 1;
        }
      splice(b1, &box, b2, tp, ts);
      bm[j] = *b1;
    }
  return 0;
}

/* Frees the calling thread's realignment buffers (worker threads call this before they end). */
void damar_bridge_release(void)
{ free(g_vec);  g_vec = NULL;  g_vmax = 0;
  free(g_tr);   g_tr = NULL;   g_tmax = 0;
}

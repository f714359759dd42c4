/* bridge.c -- bridging of two narrowly parallel local alignments of one read pair (host tail).
 *
 * When two alignments of a read pair overlap in a narrow parallel strip, the reference realigns the region between
 * them exactly and splices the three pieces: filter.c:1376-1454 (trace-point look-ups, Check_Bridge), :1456-1571
 * (Compute_Bridge_Path), :1747-1802 (Bridge), called from :1950-2059.  The exact realignment is
 * Compute_Alignment(DIFF_TRACE): Myers' O(ND) algorithm in its linear-space form -- find the middle of an optimal
 * path by meeting a forward and a backward search, recurse on the two halves -- with the differences and B-lengths
 * accumulated per trace interval of A (align.c:4327-4495, :4497-4651, :4734-4869).
 *
 * Written from that algorithm with a structure of its own (the reference-shaped restatement lives in oracle/bridge.c,
 * test infrastructure): the two searches are explicit frontiers -- an array of furthest rows per diagonal with its live
 * range -- advanced one difference at a time out of the previous frontier, and the trace bookkeeping is a small
 * "ledger" object.  Only what decides the output is kept exactly as the reference decides it: an optimal path is not
 * unique, and the emitted (diffs, b-length) pairs depend on WHERE the two searches are declared to have met and on how
 * a box is cut around that point.  Those rules are marked "tie rule" below.
 * Rare path (tandem-rich reads), a few hundred bases per call: stays on the host by design.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "damar_host.h"

/***** the two searches ********************************************************************************/

/* A search frontier after d differences: row[k] for diagonals lo..hi (k = column - row).
 *   forward : row[k] = furthest row reached from the top-left corner
 *   backward: row[k] = the row just ABOVE the furthest point reached from the bottom-right corner
 * Two buffers each, swapped per difference: a new frontier is computed out of the complete old one. */
typedef struct
{ int *row[2];
  int  cur;               /* which buffer holds the frontier */
  int  lo, hi;
} Frontier;

typedef struct
{ const char *A, *B;      /* the box: A[0..M) against B[0..N) */
  int         M, N;
} Box;

/* where the searches met: a point of an optimal path and the differences on either side */
typedef struct { int col, row, diffs; } Meet;

static int slide_fwd(const Box *b, int k, int r)            /* along diagonal k while the bases agree */
{ const char *a = b->A + k;
  const int   end = (b->N < b->M - k) ? b->N : b->M - k;
  while (r < end && b->B[r] == a[r])
    r += 1;
  return r;
}

static int slide_bwd(const Box *b, int k, int r)
{ const char *a = b->A + k;
  const int   top = (-k > 0) ? -k : 0;
  while (r >= top && b->B[r] == a[r])
    r -= 1;
  return r;
}

/* rows outside a frontier's live range, chosen so that they never win and so that the tie rules below see the same
   values the reference's in-place sweep sees at the rim of its band */
static int fwd_old(const Frontier *f, int k)
{ if (k >= f->lo && k <= f->hi) return f->row[f->cur][k];
  return (k > f->hi + 1) ? -3 : -2;
}

static int bwd_old(const Frontier *f, int k, int N)
{ return (k >= f->lo && k <= f->hi) ? f->row[f->cur][k] : N + 1; }

/* One more difference for the forward search.  Returns 1 and fills *m when it runs into the backward frontier. */
static int fwd_advance(const Box *b, Frontier *f, const Frontier *g, Meet *m, int d)
{ int *nw = f->row[f->cur ^ 1];
  int  k;
  for (k = f->hi + 1; k >= f->lo - 1; k--)
    { const int right = fwd_old(f, k + 1) + 1;      /* a base of B alone   */
      const int diag  = fwd_old(f, k) + 1;          /* a substitution      */
      const int down  = fwd_old(f, k - 1);          /* a base of A alone   */
      int r = right > diag ? right : diag;
      if (down > r) r = down;
      if (k >= g->lo && k <= g->hi)
        { const int wall = g->row[g->cur][k];
          if (r > wall)
            { /* tie rule: the meeting row is the first of (right, diag) that is already past the other search, else the
                 row just below its frontier */
              m->row = (right > wall) ? right : (diag > wall) ? diag : wall + 1;
              m->col = k + m->row;
              m->diffs = 2 * d - 1;
              return 1;
            }
        }
      nw[k] = slide_fwd(b, k, r);
    }
  f->cur ^= 1;  f->lo -= 1;  f->hi += 1;
  return 0;
}

/* The same for the backward search, against the forward frontier of the SAME number of differences. */
static int bwd_advance(const Box *b, Frontier *g, const Frontier *f, Meet *m, int d)
{ int *nw = g->row[g->cur ^ 1];
  int  k;
  for (k = g->hi + 1; k >= g->lo - 1; k--)
    { const int left = bwd_old(g, k + 1, b->N) + 1;
      const int diag = bwd_old(g, k, b->N);
      const int up   = bwd_old(g, k - 1, b->N);
      int r = left < diag ? left : diag;
      if (up < r) r = up;
      if (k >= f->lo && k <= f->hi)
        { const int wall = f->row[f->cur][k];
          if (r <= wall)
            { /* tie rule, mirrored */
              m->row = (left <= wall) ? left : (diag <= wall) ? diag : wall;
              m->col = k + m->row;
              m->diffs = 2 * d;
              return 1;
            }
        }
      nw[k] = slide_bwd(b, k, r - 1);
    }
  g->cur ^= 1;  g->lo -= 1;  g->hi += 1;
  return 0;
}

/* Middle of an optimal path of the box (align.c:4327-4495): the edit distance and a point that splits it into
   ceil(D/2) and floor(D/2) differences.  ws: 4 arrays of 2 * max(M, N) + 5 ints. */
static Meet middle(const Box *b, int *ws)
{ const int span = 2 * ((b->M > b->N) ? b->M : b->N) + 5;
  Frontier f, g;
  Meet     m;
  int      d, r;
  /* four arrays of `span` rows: the forward frontiers are indexed by the diagonal around 0, the backward ones around the
     diagonal M - N of the bottom-right corner */
  f.row[0] = ws + span / 2;
  f.row[1] = ws + span + span / 2;
  g.row[0] = ws + 2 * span + span / 2 - (b->M - b->N);
  g.row[1] = ws + 3 * span + span / 2 - (b->M - b->N);

  r = slide_fwd(b, 0, 0);
  if (r >= b->M && b->N == b->M)          /* the box is a single run of matches */
    { m.col = m.row = b->M;  m.diffs = 0;
      return m;
    }
  f.cur = 0;  f.lo = f.hi = 0;
  f.row[0][0] = r;
  g.cur = 0;  g.lo = g.hi = b->M - b->N;
  g.row[0][g.lo] = slide_bwd(b, g.lo, b->N - 1);

  for (d = 1; ; d++)
    { if (fwd_advance(b, &f, &g, &m, d))
        return m;
      if (bwd_advance(b, &g, &f, &m, d))
        return m;
    }
}

/***** the trace ledger: differences and B-length per trace interval of A ********************************/

typedef struct
{ uint16     *cell;       /* cell[2 i] differences, cell[2 i + 1] B-length of interval i (i counted from A's origin) */
  const char *origin;     /* position 0 of the A coordinates the intervals are cut in */
  int         ts;
  int        *ws;
} Ledger;

static int interval_of(const Ledger *L, const char *a) { return (int) (a - L->origin) / L->ts; }
static int room_in_interval(const Ledger *L, const char *a)       /* bases from a to the end of its interval */
{ const int at = (int) (a - L->origin);
  return (at / L->ts + 1) * L->ts - at;
}

/* `len` matched columns starting at A position a: one B base each, dealt to the intervals they fall in */
static void book_matches(Ledger *L, const char *a, int len)
{ int i = interval_of(L, a), take = room_in_interval(L, a);
  for (; len > 0; len -= take, take = L->ts, i++)
    { if (take > len) take = len;
      L->cell[2 * i + 1] += (uint16) take;
    }
}

/* bases of A with nothing opposite: a difference each, dealt the same way */
static void book_deletions(Ledger *L, const char *a, int len)
{ int i = interval_of(L, a), take = room_in_interval(L, a);
  for (; len > 0; len -= take, take = L->ts, i++)
    { if (take > len) take = len;
      L->cell[2 * i] += (uint16) take;
    }
}

/* Exact alignment of a box into the ledger (align.c:4497-4651); returns its differences.  A box that lies inside one
   trace interval is booked whole (its differences and its B-length are all the ledger wants to know); a box with at most
   one difference is booked column by column; everything else is cut at the middle of an optimal path. */
static int settle(Ledger *L, const char *A, int M, const char *B, int N)
{ Box  b;
  Meet m;
  if (M <= 0)                           /* only B left: N insertions at this point of A */
    { const int i = interval_of(L, A);
      L->cell[2 * i] += (uint16) N;
      L->cell[2 * i + 1] += (uint16) N;
      return N;
    }
  if (N <= 0)
    { book_deletions(L, A, M);
      return M;
    }
  b.A = A;  b.B = B;  b.M = M;  b.N = N;
  m = middle(&b, L->ws);

  if (m.diffs >= 2)
    { /* tie rule: a half is booked whole when it does not cross a trace boundary, else it is cut again */
      if (room_in_interval(L, A) >= m.col)
        { const int i = interval_of(L, A);
          L->cell[2 * i] += (uint16) ((m.diffs + 1) / 2);
          L->cell[2 * i + 1] += (uint16) m.row;
        }
      else
        settle(L, A, m.col, B, m.row);
      if (room_in_interval(L, A + m.col) >= M - m.col)
        { const int i = interval_of(L, A + m.col);
          L->cell[2 * i] += (uint16) (m.diffs / 2);
          L->cell[2 * i + 1] += (uint16) (N - m.row);
        }
      else
        settle(L, A + m.col, M - m.col, B + m.row, N - m.row);
      return m.diffs;
    }

  /* zero or one difference: matches up to it, the difference, matches behind it.  The meeting point of a one-difference
     box sits behind the difference when A is not the shorter side (a substitution or a base of A alone), on it otherwise */
  { const int lone_a = (m.diffs == 1 && M >= N);          /* the difference consumes a base of A */
    const int head = lone_a ? m.col - 1 : m.col;
    if (head > 0)
      book_matches(L, A, head);
    if (m.diffs == 0)
      return 0;
    { const int i = interval_of(L, A + head);
      L->cell[2 * i] += 1;
      if (M <= N)                                         /* ... and one of B, unless only A's */
        L->cell[2 * i + 1] += 1;
    }
    if (M - m.col > 0)
      book_matches(L, A + m.col, M - m.col);
  }
  return m.diffs;
}

typedef struct
{ int abpos, bbpos, aepos, bepos, diffs, tlen;
  uint16 *trace;
} BPath;

/* work buffers of the realignment, per thread: the host tail runs read-pair ranges on several threads */
static __thread int    *g_ws = NULL;   static __thread size_t g_wsn = 0;
static __thread uint16 *g_tr = NULL;   static __thread size_t g_trn = 0;

/* Compute_Alignment(align, work, DIFF_TRACE, ts) (align.c:4734-4869) for the box in p: fills p->diffs, p->tlen, p->trace
   (the thread's buffer, valid until the next call) */
static void realign_box(const char *aseq, const char *bseq, BPath *p, int ts)
{ const int asub = p->aepos - p->abpos, bsub = p->bepos - p->bbpos;
  const int big = (asub > bsub) ? asub : bsub;
  const size_t need_ws = 4 * (size_t) (2 * big + 5) + 16;
  const int ncell = 2 * (((p->aepos + (ts - 1)) / ts - p->abpos / ts) + 1);        /* one interval beyond the last */
  Ledger L;
  if (need_ws > g_wsn)
    { g_wsn = need_ws + need_ws / 4 + 10000;
      g_ws = (int *) realloc(g_ws, sizeof(int) * g_wsn);
    }
  if ((size_t) ncell > g_trn)
    { g_trn = (size_t) ncell + (size_t) ncell / 4 + 1000;
      g_tr = (uint16 *) realloc(g_tr, sizeof(uint16) * g_trn);
    }
  if (g_ws == NULL || g_tr == NULL)
    { fprintf(stderr, "damar: out of memory (bridge)\n");
      exit(1);
    }
  memset(g_tr, 0, sizeof(uint16) * (size_t) ncell);
  L.cell = g_tr - 2 * (p->abpos / ts);
  L.origin = aseq;
  L.ts = ts;
  L.ws = g_ws;
  p->diffs = settle(&L, aseq + p->abpos, asub, bseq + p->bbpos, bsub);
  if (g_tr[ncell - 1] != 0)             /* insertions booked exactly on the last boundary belong to the interval before it */
    { g_tr[ncell - 3] += g_tr[ncell - 1];
      g_tr[ncell - 4] += g_tr[ncell - 2];
    }
  p->tlen = ncell - 2;
  p->trace = g_tr;
}

/***** trace points of a path ***************************************************************************/

/* Walks the trace points of `path` from its start (dir = +1) or from its end (dir = -1) and stops at the first one at or
   beyond *want in the walking direction, measured in A (on_a) or in B.  Snaps *want to that trace point and returns its
   other coordinate; past the last trace point it returns the path's far end (filter.c:1376-1442). */
static int trace_point_at(const damar_path *path, int *want, int on_a, int dir, const damar_tpool *tp, int ts)
{ const uint16 *t = tp->val + path->toff;
  const int npts = path->tlen / 2;
  int a, b, i;
  if (dir > 0)
    { a = (path->abpos / ts) * ts;  b = path->bbpos; }
  else
    { a = ((path->aepos + (ts - 1)) / ts) * ts;  b = path->bepos; }
  for (i = 0; i < npts; i++)
    { const int seg = (dir > 0) ? t[2 * i + 1] : t[path->tlen - 1 - 2 * i];
      a += dir * ts;
      b += dir * seg;
      if (dir > 0 ? a > path->aepos : a < path->abpos)
        a = (dir > 0) ? path->aepos : path->abpos;
      if (dir * ((on_a ? a : b) - *want) >= 0)
        break;
    }
  *want = on_a ? a : b;
  return on_a ? b : a;
}

/* filter.c:1456-1571 without its debug branches: the box around the strip where p1 ends and p2 begins, realigned */
static void bridge_box(const damar_path *p1, const damar_path *p2, const char *aseq0, int alen0,
                       const char *bseq, int blen, int comp, int aovl, int bovl,
                       const damar_tpool *tp, int ts, BPath *box)
{ const char *aseq = aseq0;
  int alen = alen0, shift = 0, in, out;

  /* the strip, on the side where the two paths overlap less; then two trace intervals of margin, snapped to p1's and
     p2's own trace points so that the pieces can be spliced on interval boundaries */
  if (bovl > aovl)
    { int bin = p2->bbpos, bout = p1->bepos;
      in  = trace_point_at(p1, &bin, 0, -1, tp, ts);
      out = trace_point_at(p2, &bout, 0, +1, tp, ts);
    }
  else
    { in = p2->abpos;  out = p1->aepos;
      (void) trace_point_at(p1, &in, 1, -1, tp, ts);
      (void) trace_point_at(p2, &out, 1, +1, tp, ts);
    }
  box->abpos = in - 2 * ts;
  box->aepos = out + 2 * ts;
  box->bbpos = trace_point_at(p1, &box->abpos, 1, -1, tp, ts);
  box->bepos = trace_point_at(p2, &box->aepos, 1, +1, tp, ts);

  if (comp)
    { /* the B view of a complemented pair is realigned in the coordinates of the complemented reads, shifted so that its
         trace intervals fall where the A view's do */
      int q;
      shift = ts - box->aepos % ts;
      q = alen - box->abpos;  box->abpos = alen - box->aepos;  box->aepos = q;
      q = blen - box->bbpos;  box->bbpos = blen - box->bepos;  box->bepos = q;
      shift -= box->abpos % ts;
      aseq -= shift;
      box->abpos += shift;  box->aepos += shift;
      alen += shift;
    }

  realign_box(aseq, bseq, box, ts);

  if (comp)
    { uint16 *t = box->trace;
      int i, j, q;
      for (i = 0, j = box->tlen - 2; i < j; i += 2, j -= 2)
        { const uint16 d = t[i], l = t[i + 1];
          t[i] = t[j];  t[i + 1] = t[j + 1];
          t[j] = d;     t[j + 1] = l;
        }
      box->abpos -= shift;  box->aepos -= shift;
      alen -= shift;
      q = alen - box->abpos;  box->abpos = alen - box->aepos;  box->aepos = q;
      q = blen - box->bbpos;  box->bbpos = blen - box->bepos;  box->bepos = q;
    }
}

/* filter.c:1444-1454: a bridged segment whose values do not fit a byte is refused */
static int box_fits_bytes(const BPath *box, int ts)
{ int i;
  if (ts <= TRACE_XOVR)
    for (i = 0; i < box->tlen; i++)
      if (box->trace[i] > 250)
        return 0;
  return 1;
}

/* filter.c:1747-1802: head := head[.. box.abpos] ++ box ++ tail[box.aepos ..], in a fresh stretch of the trace pool */
static void splice(damar_path *head, const BPath *box, const damar_path *tail, damar_tpool *tp, int ts)
{ const int keep_head = 2 * ((box->abpos / ts) - (head->abpos / ts));
  const int skip_tail = (box->aepos == tail->aepos) ? tail->tlen : 2 * ((box->aepos / ts) - (tail->abpos / ts));
  const int total = keep_head + box->tlen + (tail->tlen - skip_tail);
  const uint16 *src[3];
  int    cnt[3], s, k, n = 0, diffs = 0;
  int64  at;

  if (tp->top + total >= tp->max)
    { tp->max = (int64) (1.2 * (tp->top + total)) + 1000;
      tp->val = (uint16 *) realloc(tp->val, sizeof(uint16) * (size_t) tp->max);
      if (tp->val == NULL)
        { fprintf(stderr, "damar: out of memory (trace pool)\n");
          exit(1);
        }
    }
  at = tp->top;
  tp->top += total;
  src[0] = tp->val + head->toff;               cnt[0] = keep_head;
  src[1] = box->trace;                         cnt[1] = box->tlen;
  src[2] = tp->val + tail->toff + skip_tail;   cnt[2] = tail->tlen - skip_tail;
  for (s = 0; s < 3; s++)
    for (k = 0; k < cnt[s]; k += 2)
      { tp->val[at + n++] = src[s][k];
        tp->val[at + n++] = src[s][k + 1];
        diffs += src[s][k];
      }
  head->aepos = tail->aepos;
  head->bepos = tail->bepos;
  head->diffs = diffs;
  head->toff  = at;
  head->tlen  = n;
}

/* One candidate of the second loop of Handle_Redundancies (filter.c:1998-2057).  Returns
 * non-zero when the candidate was skipped (`continue` in the reference). */
int damar_bridge_pair(const damar_bridge_ctx *ctx, damar_path *jp, damar_path *kp,
                      damar_path *p1, damar_path *p2, damar_path *b1, damar_path *b2,
                      int aovl, int bovl, int comp, int ts, damar_tpool *tp,
                      damar_path *bm, int j)
{ BPath box;
  damar_path jback, kback;

  bridge_box(p1, p2, ctx->aseq, ctx->alen, ctx->bseq, ctx->blen, 0, aovl, bovl, tp, ts, &box);
  if (!box_fits_bytes(&box, ts))
    return 1;
  jback = *jp;
  kback = *kp;
  splice(p1, &box, p2, tp, ts);
  *jp = *p1;
  kp->abpos = -1;
  __atomic_fetch_add(&damar_stat_bridges, 1, __ATOMIC_RELAXED);

  if (b1 != NULL)
    { /* the B view: roles of the sequences swapped (filter.c:1825-1829, 2025) */
      bridge_box(b1, b2, ctx->bseq, ctx->blen, ctx->aseq, ctx->alen, comp, bovl, aovl, tp, ts, &box);
      if (!box_fits_bytes(&box, ts))
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
{ free(g_ws);  g_ws = NULL;  g_wsn = 0;
  free(g_tr);  g_tr = NULL;  g_trn = 0;
}

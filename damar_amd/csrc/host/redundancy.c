/* redundancy.c -- host tail of the overlap path for ONE read pair: what becomes of the local alignments found for
 * the pair before they are written (kept on the host: a handful of paths per pair, rare).
 *
 * Results are the reference's (dalign/filter.c:1573-1686 Entwine, :1691-1741 Fusion, :1804-2077 Handle_Redundancies,
 * :2442-2483 the records); the statement is this project's own.  A path is treated as a LADDER: its trace gives the B
 * coordinate at every point of the A grid (multiples of the trace spacing) it crosses.  Two paths "meet" when their
 * ladders agree at a grid point; everything below is phrased on a Walker that climbs a ladder rung by rung, and the
 * reference's two mirrored branches (j starts first / k starts first) are ONE rule on (lead, trail) = (the path that
 * starts first on A, the other).  The oracle keeps a separately worded statement (oracle/redundancy.c); the two share
 * no text and are compared through the files they produce.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>

#include "damar_host.h"

int64 damar_stat_redundancy_calls = 0, damar_stat_fusions = 0, damar_stat_bridges = 0;

/***** trace pool: paths refer to their traces by offset, so the pool may move when it grows ********************/

static int64 pool_take(damar_tpool *tp, int64 n)
{ int64 at = tp->top;
  if (at + n >= tp->max)
    { tp->max = (int64) (1.2 * (at + n)) + 1000;              /* growth rule of filter.c:1703-1709 */
      tp->val = (uint16 *) realloc(tp->val, sizeof(uint16) * (size_t) tp->max);
      if (tp->val == NULL)
        { fprintf(stderr, "damar: out of memory (trace pool)\n");
          exit(1);
        }
    }
  tp->top = at + n;
  return at;
}

int64 damar_tpool_push(damar_tpool *tp, const uint16 *src, int n)
{ int64 at = pool_take(tp, n);
  memcpy(tp->val + at, src, sizeof(uint16) * (size_t) n);
  return at;
}

/***** ladders *************************************************************************************************/

/* A Walker stands on grid index `g` of a path with B coordinate `b`; `step` points at the (diffs, B length) pair
 * of the segment that leads to g + 1.  At its creation it stands at the path's start, which counts as the grid index
 * of the segment the start lies in (abpos / ts). */
typedef struct
{ const uint16 *step;
  int           g, b;
} Walker;

static Walker walker_on(const damar_path *p, const damar_tpool *tp, int ts)
{ Walker w;
  w.step = tp->val + p->toff;
  w.g    = p->abpos / ts;
  w.b    = p->bbpos;
  return w;
}

static int climb(Walker *w, int g)                 /* B coordinate of the path at grid index g >= w->g */
{ for (; w->g < g; w->g++, w->step += 2)
    w->b += w->step[1];
  return w->b;
}

static int gap(int x, int y) { return x < y ? y - x : x - y; }

/* How close do two paths of one read pair come on the A grid?  Compared are, in this order: the common start when both
 * start at the same A position; the grid points strictly inside both (above the later first index, below the earlier
 * end); the common end when both end at the same A position.  *meet receives the last compared point with distance 0.
 * The answer is -1 when no INTERIOR grid point was compared, whatever the ends said (filter.c:1573-1686: its `cnt`). */
static void compare_at(int at, int d, int *near, int *meet)
{ if (d > *near)
    return;
  *near = d;
  if (d == 0)
    *meet = at;
}

static int nearest_approach(const damar_path *p, const damar_path *q, const damar_tpool *tp, int ts, int *meet)
{ Walker wp = walker_on(p, tp, ts), wq = walker_on(q, tp, ts);
  int    stop = p->aepos < q->aepos ? p->aepos : q->aepos;
  int    g    = wp.g > wq.g ? wp.g : wq.g;
  int    near = 10000, inside = 0;

  if (p->abpos == q->abpos)
    { near = INT_MAX;                                         /* the common start sets the distance outright */
      compare_at(q->abpos, gap(p->bbpos, q->bbpos), &near, meet);
    }
  for (g += 1; g * ts < stop; g++, inside++)
    compare_at(g * ts, gap(climb(&wp, g), climb(&wq, g)), &near, meet);
  if (p->aepos == q->aepos)
    compare_at(q->aepos, gap(p->bepos, q->bepos), &near, meet);
  return inside > 0 ? near : -1;
}

/* front := front up to the grid point `at`, then back from `at` on (filter.c:1691-1741).  The spliced trace goes to
 * fresh pool space: both sources stay valid for other paths that still refer to them. */
static void splice_at(damar_path *front, int at, const damar_path *back, damar_tpool *tp, int ts)
{ int   keep = 2 * (at / ts - front->abpos / ts);             /* trace values of front before the grid point */
  int   skip = 2 * (at / ts - back->abpos / ts);              /* trace values of back before it */
  int   rest = back->tlen - skip;
  int64 to   = pool_take(tp, keep + rest);
  uint16 *v  = tp->val + to;
  int   i, diffs = 0;

  __atomic_fetch_add(&damar_stat_fusions, 1, __ATOMIC_RELAXED);
  memcpy(v, tp->val + front->toff, sizeof(uint16) * (size_t) keep);
  memcpy(v + keep, tp->val + back->toff + skip, sizeof(uint16) * (size_t) rest);
  for (i = 0; i < keep + rest; i += 2)
    diffs += v[i];
  front->toff  = to;
  front->tlen  = keep + rest;
  front->diffs = diffs;
  front->aepos = back->aepos;
  front->bepos = back->bepos;
}

/***** pass 1: paths that share a trace point ********************************************************************/

enum { APART, ABSORBED, GROWN };

/* Slots j > k, both alive.  What the reference decides for them (filter.c:1833-1946), as one rule:
 *   - they must be able to touch (trail starts no later than lead ends, on both reads) and meet on the grid;
 *   - same start: the longer of the two survives;  trail ends inside lead: lead survives;
 *   - trail runs on beyond lead: the two become lead[..meet] ++ trail[meet..] -- provided their B views meet as well.
 *     On B the pair runs in the same order as on A, or in the opposite order when the B read is complemented.
 * The survivor always ends up in slot j and slot k is retired; GROWN tells the caller that slot j got longer, so every
 * earlier slot has to be looked at again. */
static int settle(damar_path *am, damar_path *bm, int j, int k, int comp, int ts, damar_tpool *tp)
{ int         j_leads = am[j].abpos < am[k].abpos;
  int         li = j_leads ? j : k, ti = j_leads ? k : j;
  damar_path *lead = am + li, *trail = am + ti;
  int         ma = 0, mb = 0;

  if (trail->abpos > lead->aepos || trail->bbpos > lead->bepos)
    return APART;
  if (nearest_approach(lead, trail, tp, ts, &ma) != 0)
    return APART;

  if (trail->aepos > lead->aepos && trail->abpos != lead->abpos)
    { if (bm != NULL)
        { damar_path *bfront = bm + (comp ? ti : li), *bback = bm + (comp ? li : ti);
          if (nearest_approach(bfront, bback, tp, ts, &mb) != 0)
            return APART;
          splice_at(lead, ma, trail, tp, ts);
          splice_at(bfront, mb, bback, tp, ts);
          bm[j] = *bfront;
        }
      else
        splice_at(lead, ma, trail, tp, ts);
      am[j] = *lead;
      am[k].abpos = -1;
      return GROWN;
    }

  /* containment: with a common start (then k is the lead) the longer one, otherwise the lead */
  if (!j_leads && (lead->abpos != trail->abpos || lead->aepos > trail->aepos))
    { am[j] = am[k];
      if (bm != NULL)
        bm[j] = bm[k];
    }
  am[k].abpos = -1;
  return ABSORBED;
}

/***** pass 2: narrow parallel overlaps (filter.c:1950-2059) *****************************************************/

/* `first` starts before `second` on A.  A candidate for bridging is a pair that overlaps on both reads in the same
 * sense, neither containing the other, whose overlap is square to within 20 % (the reference compares in double). */
static int staggered(const damar_path *first, const damar_path *second, int *aovl, int *bovl)
{ int a_ok = second->abpos < first->aepos && first->aepos < second->aepos;
  int b_ok = first->bbpos < second->bbpos && second->bbpos < first->bepos && first->bepos < second->bepos;
  if (!a_ok || !b_ok)
    return 0;
  *aovl = first->aepos - second->abpos;
  *bovl = first->bepos - second->bbpos;
  return gap(*aovl, *bovl) <= .2 * (*aovl + *bovl);
}

static void bridge_all(damar_path *am, int n, damar_path *bm, int comp, int ts, damar_tpool *tp,
                       const damar_bridge_ctx *bridge)
{ int j, k;
  for (j = 1; j < n; j++)
    for (k = j - 1; k >= 0 && am[j].abpos >= 0; k--)
      { int         j_first = am[j].abpos < am[k].abpos, aovl, bovl;
        damar_path *first  = am + (j_first ? j : k), *second = am + (j_first ? k : j);
        damar_path *bfirst = NULL, *bsecond = NULL;

        if (am[k].abpos < 0 || !staggered(first, second, &aovl, &bovl))
          continue;
        if (bm != NULL)
          { int bj_first = comp ? !j_first : j_first;      /* on a complemented B read the two come in the opposite order */
            bfirst  = bm + (bj_first ? j : k);
            bsecond = bm + (bj_first ? k : j);
            if (bfirst->abpos > bsecond->abpos)
              { printf("  SYMFAIL %d %d\n", j, k);         /* the reference's diagnostic, on stdout as there */
                continue;
              }
          }
        damar_bridge_pair(bridge, am + j, am + k, first, second, bfirst, bsecond, aovl, bovl, comp, ts, tp, bm, j);
      }
}

/* am[0..n): the A-view paths of one read pair in discovery order; bm: the matching B views, or NULL.  Returns how
 * many survive, packed to the front in their slot order.  datander's variant (scrub/tandem.c:767-850) has no
 * bridging pass: its callers hand in bridge == NULL. */
static int close_ranks(damar_path *am, damar_path *bm, int n)      /* survivors to the front, slot order kept */
{ int hole = 0, s;
  while (hole < n && am[hole].abpos >= 0)      /* nothing moves before the first retired slot */
    hole += 1;
  for (s = hole + 1; s < n; s++)
    { if (am[s].abpos < 0)
        continue;
      if (bm != NULL)
        bm[hole] = bm[s];
      am[hole++] = am[s];
    }
  return hole;
}

int damar_handle_redundancies(damar_path *am, int n, damar_path *bm, int comp, int ts,
                              damar_tpool *tp, const damar_bridge_ctx *bridge)
{ int j, k;

  __atomic_fetch_add(&damar_stat_redundancy_calls, 1, __ATOMIC_RELAXED);

  for (j = 1; j < n; j++)                      /* slot j against every live earlier slot, latest first; again from */
    { k = j;                                   /* the top whenever slot j has grown */
      while (--k >= 0)
        if (am[k].abpos >= 0 && settle(am, bm, j, k, comp, ts, tp) == GROWN)
          k = j;
    }
  if (bridge != NULL)
    bridge_all(am, n, bm, comp, ts, tp, bridge);
  return close_ranks(am, bm, n);
}

/***** the records of a pair (filter.c:2442-2483): A views first, then B views with the reads exchanged ***********/

static void write_views(const damar_path *v, int n, int aread, int bread, int comp, int ts, const damar_tpool *tp,
                        Overlap_IO_Buffer *obuf)
{ int     narrow = (ts <= TRACE_XOVR);
  Overlap rec;

  memset(&rec, 0, sizeof(rec));
  rec.flags = (uint32) comp;
  rec.aread = aread;
  rec.bread = bread;
  for (; n > 0; n--, v++)
    { rec.path.abpos = v->abpos;
      rec.path.bbpos = v->bbpos;
      rec.path.aepos = v->aepos;
      rec.path.bepos = v->bepos;
      rec.path.diffs = v->diffs;
      rec.path.tlen  = v->tlen;
      rec.path.trace = tp->val + v->toff;
      if (narrow)
        Compress_TraceTo8(&rec, 1);
      AddOverlapToBuffer(obuf, &rec, narrow ? 1 : 2);
    }
}

void damar_emit_pair(damar_path *am, int na, damar_path *bm, int nb, damar_tpool *tp,
                     int comp, int ts, int aread, int bread,
                     const damar_bridge_ctx *bridge, Overlap_IO_Buffer *obuf,
                     int64 *nrec)
{ if (na > 1 && nb > 1)
    na = nb = damar_handle_redundancies(am, na, bm, comp, ts, tp, bridge);
  else if (na > 1)
    na = damar_handle_redundancies(am, na, NULL, comp, ts, tp, bridge);
  else if (nb > 1)
    nb = damar_handle_redundancies(bm, nb, NULL, comp, ts, tp, bridge);
  write_views(am, na, aread, bread, comp, ts, tp, obuf);
  write_views(bm, nb, bread, aread, comp, ts, tp, obuf);
  if (nrec != NULL)
    *nrec += na + nb;
}

/* oracle/redundancy.c -- TEST INFRASTRUCTURE: the oracle's own statement of what the reference does with the local
 * alignments of one read pair (dalign/filter.c:1573-1686 Entwine, :1691-1741 Fusion, :1804-2077 Handle_Redundancies,
 * :2442-2483 the records).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under
 * oracle/; the product's host tail is damar_amd/csrc/host/redundancy.c and is NOT linked into the oracle any more.
 *
 * It is a second, independently worded restatement (so that an oracle-vs-product comparison of this stage is not a
 * self-comparison): where the reference (and the product) walk two trace arrays in step, this file first lays out, for
 * each path, the B coordinate at every trace point of the A grid ("rungs"), and states the rules on those tables; the
 * two mirrored branches of the fusion loop are stated once, in terms of the path that starts first (`lead`) and the one
 * that starts later (`trail`).  Pinned by the reference-written goldens of tests/golden (fusion*, tandem*, tan_O ...).
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../damar_amd/csrc/host/damar_host.h"

int64 damar_stat_redundancy_calls = 0, damar_stat_fusions = 0, damar_stat_bridges = 0;

/***** trace pool *******************************************************************************************/

static uint16 *pool_room(damar_tpool *tp, int64 n)          /* filter.c:1703-1709: grow by a fifth plus a thousand */
{ if (tp->top + n >= tp->max)
    { tp->max = (int64) (1.2 * (tp->top + n)) + 1000;
      tp->val = (uint16 *) realloc(tp->val, sizeof(uint16) * (size_t) tp->max);
      if (tp->val == NULL)
        { fprintf(stderr, "oracle: out of memory (trace pool)\n");
          exit(1);
        }
    }
  return tp->val + tp->top;
}

int64 damar_tpool_push(damar_tpool *tp, const uint16 *src, int n)
{ uint16 *to = pool_room(tp, n);
  int64   at = tp->top;
  int     i;
  for (i = 0; i < n; i++)
    to[i] = src[i];
  tp->top += n;
  return at;
}

/***** a path as a ladder: the B coordinate at each rung of the A grid *************************************/

/* Rung g of a path is the A position g * ts for grid indexes first < g <= last where first = abpos / ts and the
 * path has tlen / 2 segments; the B coordinate there is bbpos plus the B lengths of the segments up to it (the final
 * segment ends at aepos, not on the grid).  rung_b(p, g) for first <= g <= first + tlen / 2, with g == first
 * standing for the start of the path. */
typedef struct
{ const damar_path *p;
  const uint16     *t;
  int               first;           /* grid index of the segment the path starts in */
} Ladder;

static Ladder ladder_of(const damar_path *p, const damar_tpool *tp, int ts)
{ Ladder l;
  l.p = p;
  l.t = tp->val + p->toff;
  l.first = p->abpos / ts;
  return l;
}

static int rung_b(const Ladder *l, int g)
{ int b = l->p->bbpos, s;
  for (s = 0; s < g - l->first; s++)
    b += l->t[2 * s + 1];
  return b;
}

static int absdiff(int a, int b) { return a > b ? a - b : b - a; }

/* Entwine (filter.c:1573-1686).  The points at which the two paths are compared, in this order: the common start if
 * both start at the same A position; every grid point strictly between the later of the two first grid indexes and the
 * earlier of the two ends; the common end if both end at the same A position.  The result is the smallest B distance
 * seen (the start sets it outright, the others lower it from an initial 10000), *where the LAST compared point at which
 * the distance is zero -- and -1, whatever was seen at the ends, if there was no grid point to compare. */
static int oracle_entwine(const damar_path *jp, const damar_path *kp, const damar_tpool *tp, int ts, int *where)
{ Ladder lj = ladder_of(jp, tp, ts), lk = ladder_of(kp, tp, ts);
  int    g0 = (lj.first > lk.first) ? lj.first : lk.first;
  int    ae = (jp->aepos < kp->aepos) ? jp->aepos : kp->aepos;
  int    closest = 10000, compared = 0, g, d;

  if (jp->abpos == kp->abpos)
    { closest = absdiff(jp->bbpos, kp->bbpos);
      if (closest == 0)
        *where = kp->abpos;
    }
  for (g = g0 + 1; g * ts < ae; g++)
    { d = absdiff(rung_b(&lj, g), rung_b(&lk, g));
      if (d <= closest)
        { closest = d;
          if (d == 0)
            *where = g * ts;
        }
      compared += 1;
    }
  if (jp->aepos == kp->aepos)
    { d = absdiff(jp->bepos, kp->bepos);
      if (d <= closest)
        { closest = d;
          if (d == 0)
            *where = kp->aepos;
        }
    }
  return compared ? closest : -1;
}

/* Fusion (filter.c:1691-1741): head := the segments of head before the grid point `at`, then the segments of tail from
 * it on, written to fresh pool space; head keeps its start and takes tail's end, diffs is recounted from the trace. */
static void oracle_fuse(damar_path *head, int at, const damar_path *tail, damar_tpool *tp, int ts)
{ int    nh = at / ts - head->abpos / ts;                 /* whole segments taken from head */
  int    st = at / ts - tail->abpos / ts;                 /* segments of tail that are dropped */
  int    nt = tail->tlen / 2 - st;
  int64  spot;
  uint16 *out;
  int    s, n = 0, diffs = 0;

  __atomic_fetch_add(&damar_stat_fusions, 1, __ATOMIC_RELAXED);
  out  = pool_room(tp, 2 * nh + 2 * nt);
  spot = tp->top;
  tp->top += 2 * nh + 2 * nt;
  for (s = 0; s < nh; s++)
    { const uint16 *seg = tp->val + head->toff + 2 * s;
      out[n++] = seg[0];  out[n++] = seg[1];  diffs += seg[0];
    }
  for (s = st; s < tail->tlen / 2; s++)
    { const uint16 *seg = tp->val + tail->toff + 2 * s;
      out[n++] = seg[0];  out[n++] = seg[1];  diffs += seg[0];
    }
  head->aepos = tail->aepos;
  head->bepos = tail->bepos;
  head->diffs = diffs;
  head->toff  = spot;
  head->tlen  = n;
}

/***** Handle_Redundancies (filter.c:1804-2077) ***********************************************************/

/* One (j, k) encounter of the fusion loop, k < j, both alive.  `lead` is the path that starts first on A (k when they
 * start together), `trail` the other.  Returns 1 if the loop over k must start again from j - 1 (a fusion made j
 * longer), 0 otherwise; in both cases k may have been retired (abpos = -1). */
static int meet(damar_path *am, damar_path *bm, int j, int k, int comp, int ts, damar_tpool *tp)
{ damar_path *jp = am + j, *kp = am + k;
  int   lead_is_j = (jp->abpos < kp->abpos);
  damar_path *lead = lead_is_j ? jp : kp, *trail = lead_is_j ? kp : jp;
  int   wa = 0, wb = 0;

  if (!(trail->abpos <= lead->aepos && trail->bbpos <= lead->bepos))      /* filter.c:1841 / 1887: cannot touch */
    return 0;
  if (oracle_entwine(lead, trail, tp, ts, &wa) != 0)
    return 0;

  if (!lead_is_j && kp->abpos == jp->abpos)                               /* same start: the longer one stays, in slot j */
    { if (kp->aepos > jp->aepos)
        { *jp = *kp;
          if (bm) bm[j] = bm[k];
        }
      kp->abpos = -1;
      return 0;
    }

  if (trail->aepos > lead->aepos)                                         /* trail runs on beyond lead: fuse them into slot j */
    { if (bm)
        { /* the B views of the two run in the same order on B as on A for a direct pair and in the opposite order for a
             complemented one (filter.c:1850-1866, 1902-1921) */
          damar_path *blead = bm + (lead_is_j ? j : k), *btrail = bm + (lead_is_j ? k : j);
          damar_path *bhead = comp ? btrail : blead, *btail = comp ? blead : btrail;
          if (oracle_entwine(bhead, btail, tp, ts, &wb) != 0)
            return 0;                                                      /* (k stays alive) */
          oracle_fuse(lead, wa, trail, tp, ts);
          oracle_fuse(bhead, wb, btail, tp, ts);
          bm[j] = *bhead;
        }
      else
        oracle_fuse(lead, wa, trail, tp, ts);
      *jp = *lead;
      kp->abpos = -1;
      return 1;
    }

  /* trail ends inside lead: lead stays, in slot j */
  if (!lead_is_j)
    { *jp = *kp;
      if (bm) bm[j] = bm[k];
    }
  kp->abpos = -1;
  return 0;
}

/* the geometric test of the bridging loop (filter.c:1972-1982): p1 starts first; they overlap on both reads without one
 * containing the other, and the overlap is within 20 % of square */
static int parallel_overlap(const damar_path *p1, const damar_path *p2, int *aovl, int *bovl)
{ if (p2->abpos >= p1->aepos || p1->aepos >= p2->aepos)
    return 0;
  if (p1->bbpos >= p2->bbpos || p2->bbpos >= p1->bepos || p1->bepos >= p2->bepos)
    return 0;
  *aovl = p1->aepos - p2->abpos;
  *bovl = p1->bepos - p2->bbpos;
  return !(abs(*aovl - *bovl) > .2 * (*aovl + *bovl));
}

int damar_handle_redundancies(damar_path *am, int n, damar_path *bm, int comp, int ts,
                              damar_tpool *tp, const damar_bridge_ctx *bridge)
{ int j, k, left;

  __atomic_fetch_add(&damar_stat_redundancy_calls, 1, __ATOMIC_RELAXED);

  for (j = 1; j < n; j++)                                  /* filter.c:1833-1946 */
    { k = j - 1;
      while (k >= 0)
        { if (am[k].abpos >= 0 && meet(am, bm, j, k, comp, ts, tp))
            k = j - 1;
          else
            k -= 1;
        }
    }

  if (bridge != NULL)                                      /* filter.c:1950-2059 (datander has no such loop, scrub/tandem.c:767-850) */
    for (j = 1; j < n; j++)
      for (k = j - 1; k >= 0 && am[j].abpos >= 0; k--)
        { damar_path *jp = am + j, *kp = am + k, *p1, *p2, *b1 = NULL, *b2 = NULL;
          int aovl, bovl;
          if (kp->abpos < 0)
            continue;
          p1 = (jp->abpos < kp->abpos) ? jp : kp;
          p2 = (jp->abpos < kp->abpos) ? kp : jp;
          if (!parallel_overlap(p1, p2, &aovl, &bovl))
            continue;
          if (bm != NULL)
            { int jfirst = (jp->abpos < kp->abpos);
              b1 = bm + ((comp == jfirst) ? k : j);
              b2 = bm + ((comp == jfirst) ? j : k);
              if (b1->abpos > b2->abpos)
                { printf("  SYMFAIL %d %d\n", j, k);
                  continue;
                }
            }
          (void) damar_bridge_pair(bridge, jp, kp, p1, p2, b1, b2, aovl, bovl, comp, ts, tp, bm, j);
        }

  left = 0;
  for (j = 0; j < n; j++)
    if (am[j].abpos >= 0)
      { if (bm != NULL)
          bm[left] = bm[j];
        am[left++] = am[j];
      }
  return left;
}

/***** the records of a pair (filter.c:2442-2483) **********************************************************/

static void put_records(const damar_path *v, int n, int aread, int bread, int comp, int ts, damar_tpool *tp,
                        Overlap_IO_Buffer *obuf)
{ Overlap o;
  int     i;
  for (i = 0; i < n; i++)
    { memset(&o, 0, sizeof(o));
      o.flags = (uint32) comp;
      o.aread = aread;
      o.bread = bread;
      o.path.abpos = v[i].abpos;  o.path.aepos = v[i].aepos;
      o.path.bbpos = v[i].bbpos;  o.path.bepos = v[i].bepos;
      o.path.diffs = v[i].diffs;  o.path.tlen  = v[i].tlen;
      o.path.trace = tp->val + v[i].toff;
      if (ts <= TRACE_XOVR)
        Compress_TraceTo8(&o, 1);
      AddOverlapToBuffer(obuf, &o, (ts <= TRACE_XOVR) ? 1 : 2);
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
  put_records(am, na, aread, bread, comp, ts, tp, obuf);
  put_records(bm, nb, bread, aread, comp, ts, tp, obuf);
  if (nrec)
    *nrec += na + nb;
}

/* oracle/redundancy.c -- TEST INFRASTRUCTURE: the oracle's statement of what the reference does with the local
 * alignments of one read pair, following the reference function by function: dalign/filter.c:1573-1686 Entwine (the
 * lock-step walk over two traces), :1691-1741 Fusion, :1804-2077 Handle_Redundancies (its two mirrored branches kept
 * apart as there), :2442-2483 the records.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.  The product's host tail (damar_amd/csrc/host/redundancy.c) is a differently worded
 * statement of the same rules (ladders / walkers, one lead-trail rule) and is NOT linked into the oracle: the two are
 * compared through the .las files they produce.  Pinned by the reference-written goldens of tests/golden (fusion*,
 * tandem*, tan_O ...).
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../damar_amd/csrc/host/damar_host.h"

int64 damar_stat_redundancy_calls = 0, damar_stat_fusions = 0, damar_stat_bridges = 0;

static int iabs(int x) { return x < 0 ? -x : x; }

static void tpool_reserve(damar_tpool *tp, int64 extra)
{ if (tp->top + extra >= tp->max)
    { tp->max = (int64) (1.2 * (tp->top + extra)) + 1000;
      tp->val = (uint16 *) realloc(tp->val, sizeof(uint16) * (size_t) tp->max);
      if (tp->val == NULL)
        { fprintf(stderr, "oracle: out of memory (trace pool)\n");
          exit(1);
        }
    }
}

int64 damar_tpool_push(damar_tpool *tp, const uint16 *src, int n)
{ int64 at;
  tpool_reserve(tp, n);
  at = tp->top;
  memcpy(tp->val + at, src, sizeof(uint16) * (size_t) n);
  tp->top += n;
  return at;
}

/* filter.c:1573-1686.  Walk two A-view paths over the trace points they share and
 * return the smallest B-distance between them (0 => they meet, *where = A coordinate
 * of the meeting trace point), or -1 if they share no interior trace point. */
static int entwine(const damar_path *jp, const damar_path *kp, const damar_tpool *tp, int ts, int *where)
{ const uint16 *jt = tp->val + jp->toff, *kt = tp->val + kp->toff;
  int y2 = jp->bbpos, b2 = kp->bbpos;
  int j = jp->abpos / ts, k = kp->abpos / ts;
  int best = 10000, seen = 0;
  int ac, ae, i, d;

  if (jp->abpos == kp->abpos)
    { best = iabs(y2 - b2);
      if (best == 0)
        *where = kp->abpos;
    }
  if (j < k)
    { ac = k * ts;
      j = 1 + 2 * (k - j);
      k = 1;
      for (i = 1; i < j; i += 2)
        y2 += jt[i];
    }
  else
    { ac = j * ts;
      k = 1 + 2 * (j - k);
      j = 1;
      for (i = 1; i < k; i += 2)
        b2 += kt[i];
    }
  ae = (jp->aepos < kp->aepos) ? jp->aepos : kp->aepos;
  for (;;)
    { ac += ts;
      if (ac >= ae)
        break;
      y2 += jt[j];
      b2 += kt[k];
      j += 2;
      k += 2;
      d = iabs(y2 - b2);
      if (d <= best)
        { best = d;
          if (d == 0)
            *where = ac;
        }
      seen += 1;
    }
  if (jp->aepos == kp->aepos)
    { d = iabs(jp->bepos - kp->bepos);
      if (d <= best)
        { best = d;
          if (d == 0)
            *where = kp->aepos;
        }
    }
  return (seen == 0) ? -1 : best;
}

/* filter.c:1691-1741: p1 := p1[..ap] ++ p2[ap..], written to fresh pool space. */
static void fuse(damar_path *p1, int ap, const damar_path *p2, damar_tpool *tp, int ts)
{ int    k1 = 2 * ((ap / ts) - (p1->abpos / ts));
  int    k2 = 2 * ((ap / ts) - (p2->abpos / ts));
  int    len = k1 + (p2->tlen - k2);
  int64  at;
  int    n = 0, diff = 0, k;
  uint16 *dst;

  __atomic_fetch_add(&damar_stat_fusions, 1, __ATOMIC_RELAXED);
  tpool_reserve(tp, len);
  at  = tp->top;
  tp->top += len;
  dst = tp->val + at;
  for (k = 0; k < k1; k += 2)
    { dst[n++] = tp->val[p1->toff + k];
      dst[n++] = tp->val[p1->toff + k + 1];
      diff += tp->val[p1->toff + k];
    }
  for (k = k2; k < p2->tlen; k += 2)
    { dst[n++] = tp->val[p2->toff + k];
      dst[n++] = tp->val[p2->toff + k + 1];
      diff += tp->val[p2->toff + k];
    }
  p1->aepos = p2->aepos;
  p1->bepos = p2->bepos;
  p1->diffs = diff;
  p1->toff  = at;
  p1->tlen  = n;
}

/* filter.c:1804-2077.  am[0..n) are the A-view paths of one read pair in discovery
 * order, bm (may be NULL) the matching B-view paths.  Returns the surviving count. */
int damar_handle_redundancies(damar_path *am, int n, damar_path *bm, int comp, int ts,
                              damar_tpool *tp, const damar_bridge_ctx *bridge)
{ int hasB = (bm != NULL);
  int j, k, dist, awhen = 0, bwhen = 0, out;

  __atomic_fetch_add(&damar_stat_redundancy_calls, 1, __ATOMIC_RELAXED);

  /* pass 1: alignments that share a trace point are fused (filter.c:1833-1946) */
  for (j = 1; j < n; j++)
    { damar_path *jp = am + j;
      for (k = j - 1; k >= 0; k--)
        { damar_path *kp = am + k;
          if (kp->abpos < 0)
            continue;
          if (jp->abpos < kp->abpos)
            { if (!(kp->abpos <= jp->aepos && kp->bbpos <= jp->bepos))
                continue;
              dist = entwine(jp, kp, tp, ts, &awhen);
              if (dist != 0)
                continue;
              if (kp->aepos > jp->aepos)
                { if (hasB)
                    { if (comp)
                        { if (entwine(bm + k, bm + j, tp, ts, &bwhen) != 0)
                            continue;
                          fuse(jp, awhen, kp, tp, ts);
                          fuse(bm + k, bwhen, bm + j, tp, ts);
                          bm[j] = bm[k];
                        }
                      else
                        { if (entwine(bm + j, bm + k, tp, ts, &bwhen) != 0)
                            continue;
                          fuse(jp, awhen, kp, tp, ts);
                          fuse(bm + j, bwhen, bm + k, tp, ts);
                        }
                    }
                  else
                    fuse(jp, awhen, kp, tp, ts);
                  kp->abpos = -1;
                  k = j;                 /* rescan everything before j against the fusion */
                  continue;
                }
              kp->abpos = -1;
            }
          else
            { if (!(jp->abpos <= kp->aepos && jp->bbpos <= kp->bepos))
                continue;
              dist = entwine(kp, jp, tp, ts, &awhen);
              if (dist != 0)
                continue;
              if (kp->abpos == jp->abpos)
                { if (kp->aepos > jp->aepos)
                    { *jp = *kp;
                      if (hasB)
                        bm[j] = bm[k];
                    }
                }
              else if (jp->aepos > kp->aepos)
                { if (hasB)
                    { if (comp)
                        { if (entwine(bm + j, bm + k, tp, ts, &bwhen) != 0)
                            continue;
                          fuse(kp, awhen, jp, tp, ts);
                          *jp = *kp;
                          fuse(bm + j, bwhen, bm + k, tp, ts);
                        }
                      else
                        { if (entwine(bm + k, bm + j, tp, ts, &bwhen) != 0)
                            continue;
                          fuse(kp, awhen, jp, tp, ts);
                          *jp = *kp;
                          fuse(bm + k, bwhen, bm + j, tp, ts);
                          bm[j] = bm[k];
                        }
                    }
                  else
                    { fuse(kp, awhen, jp, tp, ts);
                      *jp = *kp;
                    }
                  kp->abpos = -1;
                  k = j;
                  continue;
                }
              else
                { *jp = *kp;
                  if (hasB)
                    bm[j] = bm[k];
                }
              kp->abpos = -1;
            }
        }
    }

  /* pass 2: narrow parallel overlaps are bridged by an exact realignment
   * (filter.c:1950-2059).  datander's variant (scrub/tandem.c:767-850) has no such pass:
   * its callers hand in bridge == NULL. */
  for (j = 1; bridge != NULL && j < n; j++)
    { damar_path *jp = am + j;
      if (jp->abpos < 0)
        continue;
      for (k = j - 1; k >= 0; k--)
        { damar_path *kp = am + k, *p1, *p2, *b1 = NULL, *b2 = NULL;
          int aovl, bovl;

          if (kp->abpos < 0)
            continue;
          if (jp->abpos < kp->abpos)
            { p1 = jp; p2 = kp; }
          else
            { p1 = kp; p2 = jp; }
          if (p2->abpos >= p1->aepos || p1->aepos >= p2->aepos ||
              p1->bbpos >= p2->bbpos || p2->bbpos >= p1->bepos || p1->bepos >= p2->bepos)
            continue;
          aovl = p1->aepos - p2->abpos;
          bovl = p1->bepos - p2->bbpos;
          if (iabs(aovl - bovl) > .2 * (aovl + bovl))
            continue;
          if (hasB)
            { if (comp == (jp->abpos < kp->abpos))
                { b1 = bm + k; b2 = bm + j; }
              else
                { b1 = bm + j; b2 = bm + k; }
              if (b1->abpos > b2->abpos)
                { printf("  SYMFAIL %d %d\n", j, k);
                  continue;
                }
            }
          if (damar_bridge_pair(bridge, jp, kp, p1, p2, b1, b2, aovl, bovl, comp, ts, tp, bm, j))
            continue;
        }
    }

  out = 0;
  for (j = 0; j < n; j++)
    if (am[j].abpos >= 0)
      { if (hasB)
          bm[out] = bm[j];
        am[out++] = am[j];
      }
  return out;
}

/* filter.c:2442-2483: redundancy handling, then A records, then B records. */
void damar_emit_pair(damar_path *am, int na, damar_path *bm, int nb, damar_tpool *tp,
                     int comp, int ts, int aread, int bread,
                     const damar_bridge_ctx *bridge, Overlap_IO_Buffer *obuf,
                     int64 *nrec)
{ int     small  = (ts <= TRACE_XOVR);
  int     tbytes = small ? 1 : 2;
  Overlap ovl;
  int     i;

  if (na > 1)
    { if (nb > 1)
        na = nb = damar_handle_redundancies(am, na, bm, comp, ts, tp, bridge);
      else
        na = damar_handle_redundancies(am, na, NULL, comp, ts, tp, bridge);
    }
  else if (nb > 1)
    nb = damar_handle_redundancies(bm, nb, NULL, comp, ts, tp, bridge);

  memset(&ovl, 0, sizeof(ovl));
  ovl.flags = (uint32) comp;
  for (i = 0; i < na + nb; i++)
    { const damar_path *p = (i < na) ? am + i : bm + (i - na);
      ovl.aread      = (i < na) ? aread : bread;
      ovl.bread      = (i < na) ? bread : aread;
      ovl.path.tlen  = p->tlen;
      ovl.path.diffs = p->diffs;
      ovl.path.abpos = p->abpos;
      ovl.path.bbpos = p->bbpos;
      ovl.path.aepos = p->aepos;
      ovl.path.bepos = p->bepos;
      ovl.path.trace = tp->val + p->toff;
      if (small)
        Compress_TraceTo8(&ovl, 1);
      AddOverlapToBuffer(obuf, &ovl, tbytes);
    }
  if (nrec)
    *nrec += na + nb;
}

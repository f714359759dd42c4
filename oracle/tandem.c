/* tandem.c -- ORACLE (test infrastructure): datander's self-tandem path on the CPU.
 *
 * Restates reference scrub/tandem.c: tuple_thread :395-423 (rpos = index of the k-mer's last
 * base PLUS ONE), count_thread :556-589 (.code := distance to the previous equal k-mer of the
 * same read, 0 for the first of a run -- except the very first record of the sorted array,
 * which keeps its k-mer code: nothing ever overwrites it), the re-sort on (read, rpos)
 * :1298, report_thread :895-1175 and Match_Self :1182-1428.  Because every position of every
 * read owns exactly one k-mer, "re-sort by (read, rpos)" is a scatter back to position order.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "oracle.h"
#include "../damar_amd/csrc/host/damar_host.h"

#define PANEL_SIZE     50000
#define PANEL_OVERLAP  10000

typedef struct { uint64 code; int pos; int read; } TK;

static int by_code(const void *x, const void *y)      /* stable through the pos tie-break */
{ const TK *a = (const TK *) x, *b = (const TK *) y;
  if (a->code != b->code) return (a->code < b->code) ? -1 : 1;
  if (a->read != b->read) return a->read - b->read;
  return a->pos - b->pos;
}

static void span_of(const Path *p, int ts, int bshift, int *mind, int *maxd)   /* tandem.c:852-881 */
{ const uint16 *pt = (const uint16 *) p->trace;
  int low, hgh, dd, i, tlen = p->tlen - 2;
  low = hgh = p->abpos - p->bbpos;
  dd = p->aepos - p->bepos;
  if (dd < low) low = dd; else if (dd > hgh) hgh = dd;
  dd = (p->abpos / ts) * ts - p->bbpos;
  for (i = 1; i < tlen; i += 2)
    { dd += ts - pt[i];
      if (dd < low) low = dd; else if (dd > hgh) hgh = dd;
    }
  *mind = (low >> bshift) - 1;
  *maxd = (hgh >> bshift) + 1;
}

void oracle_match_self(const HITS_DB *block, const OParams *prm, Align_Spec *spec,
                       int64 *counts, OWaveStats *stats)
{ const int K = prm->kmer, W = prm->binshift, H = prm->hitmin, ts = Trace_Spacing(spec);
  const int nreads = block->nreads;
  const uint64 kmask = (K == 32) ? ~0ull : ((1ull << (2 * K)) - 1);
  const char *bases = (const char *) block->bases;
  int64 kmers = block->reads[nreads].boff - (int64) K * nreads, n = 0, i;
  TK   *t;
  int  *dist;                       /* dist[koff[r] + (apos - K)] for apos in [K, rlen] */
  int64 *koff;
  int   maxdiag = block->maxlen >> W, mindiag = (-block->maxlen) >> W, w = maxdiag - mindiag + 1;
  int  *store, *score, *lastp, *lasta;
  int   maxtp = 2 * (block->maxlen / ts + 2) + 4, r;
  uint16 *atr, *btr;
  damar_path *am = NULL;
  int   amax = 0;
  damar_tpool tp = { NULL, 0, 0 };
  Overlap_IO_Buffer *obuf = OVL_IO_Buffer(spec);
  int64 nfilt = 0, ncheck = 0;

  if (counts) counts[0] = counts[1] = counts[2] = 0;
  if (kmers <= 0)
    return;
  t = (TK *) malloc(sizeof(TK) * (size_t) kmers);
  koff = (int64 *) malloc(sizeof(int64) * (size_t) (nreads + 1));
  for (r = 0; r < nreads; r++)
    { const char *s = bases + block->reads[r].boff;
      uint64 c = 0;
      int    p = 0, x;
      koff[r] = n;
      for (x = 1; x < K; x++)
        c = (c << 2) | (uint64) s[p++];
      while ((x = s[p]) != 4)
        { c = ((c << 2) | (uint64) x) & kmask;
          t[n].code = c;
          t[n].pos  = ++p;
          t[n].read = r;
          n += 1;
        }
    }
  koff[nreads] = n;
  qsort(t, (size_t) kmers, sizeof(TK), by_code);

  dist = (int *) calloc((size_t) kmers, sizeof(int));
  for (i = 0; i < kmers; i++)
    { int d = 0;
      if (i == 0)
        d = (int) t[0].code;                                   /* never overwritten in the reference */
      else if (t[i].code == t[i - 1].code && t[i].read == t[i - 1].read)
        d = t[i].pos - t[i - 1].pos;
      dist[koff[t[i].read] + (t[i].pos - K)] = d;
    }
  free(t);

  store = (int *) calloc((size_t) (3 * w + 8), sizeof(int));
  score = store + 4 - mindiag;  lastp = score + w;  lasta = lastp + w;
  atr = (uint16 *) malloc(sizeof(uint16) * (size_t) maxtp);
  btr = (uint16 *) malloc(sizeof(uint16) * (size_t) maxtp);

  for (r = 0; r < nreads; r++)
    { const int alen = block->reads[r].rlen;
      const int *code = dist + koff[r] - K;       /* code[apos] */
      const char *aseq = bases + block->reads[r].boff;
      int amarkb = K, amarke = PANEL_SIZE, apos, na = 0;
      if (amarke >= alen)
        amarke = alen + 1;
      tp.top = 0;
      for (;;)
        { for (apos = amarkb; apos < amarke; apos++)
            { int d = code[apos];
              if (d == 0) continue;
              d >>= W;
              if (apos - lastp[d] >= K) score[d] += K; else score[d] += apos - lastp[d];
              lastp[d] = apos;
            }
          for (apos = amarkb; apos < amarke; apos++)
            { int dg = code[apos], d, bpos, lo, hi, ae;
              Path apath, bpath;
              if (dg == 0) continue;
              d = dg >> W;
              if (!(apos > lasta[d] && (score[d] + score[d + 1] >= H || score[d] + score[d - 1] >= H)))
                continue;
              bpos = apos - dg;
              nfilt += 1;
              oracle_local_alignment(aseq, alen, aseq, alen, 0, dg, apos + bpos, spec, &apath, &bpath, atr, btr, stats);
              span_of(&apath, ts, W, &lo, &hi);
              if (d < lo) lo = d; else if (d > hi) hi = d;
              ae = apath.aepos;
              for (d = lo; d <= hi; d++)
                if (ae > lasta[d])
                  lasta[d] = ae;
              if ((apath.aepos - apath.abpos) + (apath.bepos - apath.bbpos) >= prm->minover)
                { if (na >= amax)
                    { amax = (int) (1.2 * na) + 100;
                      am = (damar_path *) realloc(am, sizeof(damar_path) * (size_t) amax);
                    }
                  am[na].tlen = apath.tlen;   am[na].diffs = apath.diffs;
                  am[na].abpos = apath.abpos; am[na].bbpos = apath.bbpos;
                  am[na].aepos = apath.aepos; am[na].bepos = apath.bepos;
                  am[na].toff = damar_tpool_push(&tp, atr, apath.tlen);
                  na += 1;
                }
            }
          for (apos = amarkb; apos < amarke; apos++)
            { int d = code[apos];
              if (d == 0) continue;
              d >>= W;
              score[d] = lastp[d] = 0;
            }
          if (amarke > alen)
            break;
          amarkb = amarke - PANEL_OVERLAP;
          amarke = amarkb + PANEL_SIZE;
          if (amarke > alen)
            amarke = alen + 1;
        }
      for (apos = K; apos <= alen; apos++)
        { int d0 = code[apos], d;
          if (d0 == 0) continue;
          d0 >>= W;
          for (d = d0; d <= maxdiag; d++)
            { if (lasta[d] == 0) break;
              lasta[d] = 0;
            }
          for (d = d0 - 1; d >= mindiag; d--)
            { if (lasta[d] == 0) break;
              lasta[d] = 0;
            }
        }
      damar_emit_pair(am, na, NULL, 0, &tp, 0, ts, r + block->ufirst, r + block->ufirst, NULL, obuf, &ncheck);
    }
  free(dist); free(koff); free(store); free(atr); free(btr); free(am); free(tp.val);
  if (counts)
    { counts[0] = kmers; counts[1] = nfilt; counts[2] = ncheck; }
}

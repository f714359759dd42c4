/* wave.c -- ORACLE (test infrastructure): the Local_Alignment O(n*d) wave on the CPU.
 *
 * Restates align.c:409-1122 (forward_wave), :1126-1898 (reverse_wave) and
 * :1904-2097 (Local_Alignment) for the only call shape the overlap path uses
 * (filter.c:2316: low == hgh == seed diagonal, lbord = hbord = -1).
 *
 * Shape: every wave step reads the PREVIOUS wave's per-diagonal state (cur) and
 * writes the next one (nxt); each diagonal is computed independently of the others,
 * then one ordered scan picks the new best / trim point.  That is the form the HIP
 * kernel uses (one lane per diagonal), so the two can be compared line by line.
 * Per-diagonal state: V furthest anti-diagonal, T 64-bit match
 * history, M its gated popcount, HA/HB heads of the A-/B-pebble chains, NA/NB the
 * next A/B trace mark the diagonal will cross.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>

#include "oracle.h"

#define HIST_TOP   0x1000000000000000ull      /* bit 60, align.c:192 */
#define HIST_FULL  0x0fffffffffffffffull      /* align.c:193 */
#define HIST_LEN   60
#define TRIM_MASK  0x7fff
#define TRIM_BITS  15
#define MAX_TRIM_LAG 200                      /* align.c:195 TRIM_MLAG */
#define MAX_WAVE_LAG 30                       /* align.c:196 WAVE_LAG  */

typedef struct { int ptr, diag, diff, mark; } Pebble;

typedef struct
{ int    *V[2];
  int    *M[2];
  uint64 *T[2];
  int    *HA[2];
  int    *HB[2];
  int    *NA;
  int    *NB;
  int     kmin;          /* diagonal of array slot 0 */
  int     span;
  Pebble *cells;
  int     ncell, cmax;
} Band;

static void band_init(Band *w, int alen, int blen)
{ int i;
  w->kmin = -(blen + 4);
  w->span = alen + blen + 9;
  for (i = 0; i < 2; i++)
    { w->V[i]  = (int *) malloc(sizeof(int) * (size_t) w->span);
      w->M[i]  = (int *) malloc(sizeof(int) * (size_t) w->span);
      w->T[i]  = (uint64 *) malloc(sizeof(uint64) * (size_t) w->span);
      w->HA[i] = (int *) malloc(sizeof(int) * (size_t) w->span);
      w->HB[i] = (int *) malloc(sizeof(int) * (size_t) w->span);
    }
  w->NA = (int *) malloc(sizeof(int) * (size_t) w->span);
  w->NB = (int *) malloc(sizeof(int) * (size_t) w->span);
  w->cmax  = 1 << 14;
  w->cells = (Pebble *) malloc(sizeof(Pebble) * (size_t) w->cmax);
  w->ncell = 0;
}

static void band_free(Band *w)
{ int i;
  for (i = 0; i < 2; i++)
    { free(w->V[i]); free(w->M[i]); free(w->T[i]); free(w->HA[i]); free(w->HB[i]); }
  free(w->NA); free(w->NB); free(w->cells);
}

static int drop(Band *w, int ptr, int diag, int diff, int mark)
{ if (w->ncell >= w->cmax)
    { w->cmax = 2 * w->cmax;
      w->cells = (Pebble *) realloc(w->cells, sizeof(Pebble) * (size_t) w->cmax);
    }
  w->cells[w->ncell].ptr  = ptr;
  w->cells[w->ncell].diag = diag;
  w->cells[w->ncell].diff = diff;
  w->cells[w->ncell].mark = mark;
  return w->ncell++;
}

/* chain head -> array of cell indices in root-to-head order; returns count */
static int unwind(const Band *w, int head, int **buf, int *cap)
{ int n = 0, h, i;
  for (h = head; h >= 0; h = w->cells[h].ptr)
    n += 1;
  if (n > *cap)
    { *cap = n + 64;
      *buf = (int *) realloc(*buf, sizeof(int) * (size_t) *cap);
    }
  for (h = head, i = n - 1; h >= 0; h = w->cells[h].ptr, i--)
    (*buf)[i] = h;
  return n;
}

typedef struct
{ const char *aseq, *bseq;
  int   ts, ave, reach;
  const int16 *score, *table;
  int   minp, maxp, aoff, boff;
  OWaveStats *st;
} Ctx;

/* band statistics (DAMAR_ORACLE_BANDHIST, profiles/r05_bandhist.txt): per step and per pass */
static void bandstat_pass_begin(OWaveStats *st)
{ if (st->pass_cells > 0)
    { const int m = st->pass_max;
      const int c = m <= 13 ? 0 : m <= 14 ? 1 : m <= 16 ? 2 : m <= 29 ? 3 : 4;
      st->pass_n[c] += 1;  st->pass_cellsum[c] += st->pass_cells;
    }
  st->pass_max = 0;  st->pass_cells = 0;  st->pass_wide = 0;
  st->cur_over = 0;  st->dirs += 1;
}

static void bandstat_step(OWaveStats *st, int width)
{ st->bandhist[width > 129 ? 129 : width] += 1;
  if (width > 31 && !st->cur_over) { st->cur_over = 1; st->dirs_over31 += 1; }
  if (st->cur_over) st->steps_after_over31 += 1;
  st->pass_cells += width;
  if (width > st->pass_max) st->pass_max = width;
  if (!st->pass_wide && width > 14) { st->pass_wide = 1; st->promotions += 1; }
  else if (st->pass_wide && width <= 12) st->pass_wide = 0;
  if (st->pass_wide) st->steps_wide += 1; else st->steps_narrow += 1;
}


/* What the end point of one direction needs (align.c:436-442). */
typedef struct
{ int a, y, d, ha, hb; } Tip;

/***** forward: align.c:409-1122 ***********************************************************/

static void forward(const Ctx *c, Band *w, int diag, int mida,
                    Path *apath, uint16 *atrace, int *atlen_out, uint16 *btrace, int *btlen_out)
{ if (c->st) bandstat_pass_begin(c->st);
  const char *aseq = c->aseq, *bseq = c->bseq;
  const int   TS = c->ts;
  int   cur = 0, nxt = 1;
  int   low = diag, hgh = diag, dif = 0;
  int   o = -w->kmin;                 /* slot of diagonal k is k+o */
  int   besta, besty, lasta, more;
  Tip   trim, reach;
  int   reachm = -1;
  int   aclip = INT_MAX, bclip = -INT_MAX;

  w->ncell = 0;
  besta = lasta = mida;
  besty = (mida - hgh) >> 1;
  trim.a = reach.a = mida;
  trim.y = reach.y = besty;
  trim.d = reach.d = 0;
  trim.ha = reach.ha = 0;
  trim.hb = reach.hb = 1;
  more = 1;

  /* wave 0 on the seed diagonal (align.c:491-626) */
  { int k = diag, y = (mida - k) >> 1, na, nb, ha, hb, v;
    const char *a = aseq + k;

    na = (((y + k) + (TS - c->aoff)) / TS - 1) * TS + c->aoff;
    ha = drop(w, -1, k, 0, na);
    na += TS;
    nb = ((y + (TS - c->boff)) / TS - 1) * TS + c->boff;
    hb = drop(w, -1, k, 0, nb);
    nb += TS;
    for (;;)
      { int cb = bseq[y], ca;
        if (cb == 4)
          { more = 0; if (bclip < k) bclip = k; break; }
        ca = a[y];
        if (cb != ca)
          { if (ca == 4) { more = 0; aclip = k; }
            break;
          }
        y += 1;
      }
    v = (y << 1) + k;
    while (y + k >= na) { ha = drop(w, ha, k, 0, na); na += TS; }
    while (y >= nb)     { hb = drop(w, hb, k, 0, nb); nb += TS; }
    if (v > besta)
      { besta = lasta = trim.a = v;
        besty = trim.y = y;
        trim.ha = ha;
        trim.hb = hb;
      }
    w->V[cur][k + o] = v;  w->T[cur][k + o] = HIST_FULL;  w->M[cur][k + o] = HIST_LEN;
    w->HA[cur][k + o] = ha; w->HB[cur][k + o] = hb;
    w->NA[k + o] = na;      w->NB[k + o] = nb;
  }

#define CLIP_FWD()                                                                         \
  if (more == 0)                                                                           \
    { if (bseq[besty] != 4 && aseq[besta - besty] != 4)                                    \
        more = 1;                                                                          \
      if (hgh >= aclip)                                                                    \
        { hgh = aclip - 1;                                                                 \
          if (reachm <= w->M[cur][aclip + o])                                              \
            { reachm = w->M[cur][aclip + o]; reach.a = w->V[cur][aclip + o];               \
              reach.y = (reach.a - aclip) / 2; reach.d = dif;                              \
              reach.ha = w->HA[cur][aclip + o]; reach.hb = w->HB[cur][aclip + o]; }        \
        }                                                                                  \
      if (low <= bclip)                                                                    \
        { low = bclip + 1;                                                                 \
          if (reachm <= w->M[cur][bclip + o])                                              \
            { reachm = w->M[cur][bclip + o]; reach.a = w->V[cur][bclip + o];               \
              reach.y = (reach.a - bclip) / 2; reach.d = dif;                              \
              reach.ha = w->HA[cur][bclip + o]; reach.hb = w->HB[cur][bclip + o]; }        \
        }                                                                                  \
      aclip = INT_MAX;                                                                     \
      bclip = -INT_MAX;                                                                    \
    }

  CLIP_FWD()

  while (more && lasta >= besta - MAX_TRIM_LAG)
    { int k, nlow, nhgh;

      if (hgh < low)
        { if (c->st) c->st->empty_band = 1;
          break;
        }
      /* widen by one on each side unless a border forbids it (align.c:675-776) */
      nlow = low - 1;
      nhgh = hgh + 1;
      if (nlow >= c->minp)
        { w->NA[nlow + o] = w->NA[nlow + 1 + o];
          w->NB[nlow + o] = w->NB[nlow + 1 + o];
          w->V[cur][nlow + o] = -1;
        }
      else
        nlow += 1;
      if (nhgh <= c->maxp)
        { w->NA[nhgh + o] = w->NA[nhgh - 1 + o];
          w->NB[nhgh + o] = w->NB[nhgh - 1 + o];
          w->V[cur][nhgh + o] = -1;
        }
      else
        nhgh -= 1;
      low = nlow;
      hgh = nhgh;
      w->V[cur][hgh + 1 + o] = w->V[cur][low - 1 + o] = -1;
      dif += 1;

      if (c->st)
        bandstat_step(c->st, hgh - low + 1);
      /* every diagonal of the new wave from the old wave (align.c:781-909) */
      for (k = low; k <= hgh; k++)
        { int ac = w->V[cur][k + o], am = w->V[cur][k - 1 + o], ap = w->V[cur][k + 1 + o];
          int from, v, y, m, ha, hb;
          uint64 b;
          const char *a = aseq + k;

          if (ac < am)
            from = (am < ap) ? k + 1 : k - 1;
          else
            from = (ac < ap) ? k + 1 : k;
          v  = (from == k) ? ac + 2 : (from == k + 1 ? ap + 1 : am + 1);
          m  = w->M[cur][from + o];
          b  = w->T[cur][from + o];
          ha = w->HA[cur][from + o];
          hb = w->HB[cur][from + o];

          if (b & HIST_TOP)
            m -= 1;
          b <<= 1;
          y = (v - k) >> 1;
          for (;;)
            { int cb = bseq[y], ca;
              if (cb == 4)
                { more = 0; if (bclip < k) bclip = k; break; }
              ca = a[y];
              if (cb != ca)
                { if (ca == 4) { more = 0; if (k < aclip) aclip = k; }
                  break;
                }
              y += 1;
              if ((b & HIST_TOP) == 0)
                m += 1;
              b = (b << 1) | 1;
            }
          v = (y << 1) + k;

          while (y + k >= w->NA[k + o])
            { if (w->cells[ha].mark < w->NA[k + o])
                ha = drop(w, ha, k, dif, w->NA[k + o]);
              w->NA[k + o] += TS;
            }
          while (y >= w->NB[k + o])
            { if (w->cells[hb].mark < w->NB[k + o])
                hb = drop(w, hb, k, dif, w->NB[k + o]);
              w->NB[k + o] += TS;
            }
          w->V[nxt][k + o] = v;  w->T[nxt][k + o] = b;  w->M[nxt][k + o] = m;
          w->HA[nxt][k + o] = ha; w->HB[nxt][k + o] = hb;
        }

      /* ordered scan, highest diagonal first (align.c:911-928) */
      for (k = hgh; k >= low; k--)
        { int v = w->V[nxt][k + o];
          if (v > besta)
            { uint64 b = w->T[nxt][k + o];
              besta = v;
              besty = (v - k) >> 1;
              if (w->M[nxt][k + o] >= c->ave)
                { lasta = v;
                  if (c->table[b & TRIM_MASK] >= 0 &&
                      c->table[(b >> TRIM_BITS) & TRIM_MASK] + c->score[b & TRIM_MASK] >= 0)
                    { trim.a = v; trim.y = besty; trim.d = dif;
                      trim.ha = w->HA[nxt][k + o]; trim.hb = w->HB[nxt][k + o];
                    }
                }
            }
        }
      cur ^= 1;
      nxt ^= 1;

      CLIP_FWD()

      /* drop diagonals lagging more than 30 behind the best (align.c:977-986) */
      { int n = besta - MAX_WAVE_LAG;
        while (hgh >= low)
          if (w->V[cur][hgh + o] < n)
            hgh -= 1;
          else
            { while (w->V[cur][low + o] < n)
                low += 1;
              break;
            }
      }
      if (c->st)
        { int wd = hgh - low + 1;
          c->st->waves += 1;
          c->st->cells += wd;
          if (wd > c->st->maxband)
            c->st->maxband = wd;
        }
    }
#undef CLIP_FWD

  /* end point and traces (align.c:1001-1118) */
  { int  trimx, trimy, trimd, ha, hb;
    int *chain = NULL, cap = 0, n, i;
    int  atlen = 0, btlen = 0;
    int  b, e, k;

    if (reachm >= 0 && c->reach)
      { trimx = reach.a - reach.y; trimy = reach.y; trimd = reach.d; ha = reach.ha; hb = reach.hb; }
    else
      { trimx = trim.a - trim.y; trimy = trim.y; trimd = trim.d; ha = trim.ha; hb = trim.hb; }

    n = unwind(w, ha, &chain, &cap);
    k = w->cells[chain[0]].diag;
    b = (mida - k) / 2;
    e = 0;
    for (i = 1; i < n; i++)
      { const Pebble *p = w->cells + chain[i];
        int a = p->mark - p->diag;
        k = p->diag;
        atrace[atlen++] = (uint16) (p->diff - e);
        atrace[atlen++] = (uint16) (a - b);
        b = a;
        e = p->diff;
      }
    if (b + k != trimx)
      { atrace[atlen++] = (uint16) (trimd - e);
        atrace[atlen++] = (uint16) (trimy - b);
      }
    else if (b != trimy && atlen > 0)
      { atrace[atlen - 1] = (uint16) (atrace[atlen - 1] + (trimy - b));
        atrace[atlen - 2] = (uint16) (atrace[atlen - 2] + (trimd - e));
      }

    n = unwind(w, hb, &chain, &cap);
    k = w->cells[chain[0]].diag;
    b = (mida + k) / 2;
    e = 0;
    for (i = 1; i < n; i++)
      { const Pebble *p = w->cells + chain[i];
        int a = p->mark + p->diag;
        k = p->diag;
        btrace[btlen++] = (uint16) (p->diff - e);
        btrace[btlen++] = (uint16) (a - b);
        b = a;
        e = p->diff;
      }
    if (b - k != trimy)
      { btrace[btlen++] = (uint16) (trimd - e);
        btrace[btlen++] = (uint16) (trimx - b);
      }
    else if (b != trimx && btlen > 0)
      { btrace[btlen - 1] = (uint16) (btrace[btlen - 1] + (trimx - b));
        btrace[btlen - 2] = (uint16) (btrace[btlen - 2] + (trimd - e));
      }
    free(chain);

    apath->aepos = trimx;
    apath->bepos = trimy;
    apath->diffs = trimd;
    *atlen_out = atlen;
    *btlen_out = btlen;
    if (c->st)
      c->st->pebbles += w->ncell;
  }
}

/***** reverse: align.c:1126-1898 **********************************************************/

/* atrace/btrace point at the forward pass's first value; this pass writes at negative
 * indices (prepends).  *atlen_io / *btlen_io hold the forward lengths on entry and the
 * total lengths on return; *aback / *bback return how many values were prepended. */
static void reverse(const Ctx *c, Band *w, int diag, int mida,
                    Path *apath, uint16 *atrace, int *atlen_io, int *aback,
                    uint16 *btrace, int *btlen_io, int *bback)
{ if (c->st) bandstat_pass_begin(c->st);
  const char *aseq = c->aseq - 1, *bseq = c->bseq - 1;
  const int   TS = c->ts;
  int   cur = 0, nxt = 1;
  int   low = diag, hgh = diag, dif = 0;
  int   o = -w->kmin;
  int   besta, besty, lasta, more;
  Tip   trim, reach;
  int   reachm = -1;
  int   aclip = -INT_MAX, bclip = INT_MAX;

  w->ncell = 0;
  besta = lasta = mida;
  besty = (mida - hgh) >> 1;
  trim.a = reach.a = mida;
  trim.y = reach.y = besty;
  trim.d = reach.d = 0;
  trim.ha = reach.ha = 0;
  trim.hb = reach.hb = 1;
  more = 1;

  /* wave 0 (align.c:1206-1339) */
  { int k = diag, y = (mida - k) >> 1, na, nb, ha, hb, v;
    const char *a = aseq + k;

    na = (((y + k) + (TS - c->aoff) - 1) / TS - 1) * TS + c->aoff;
    ha = drop(w, -1, k, 0, y + k);
    nb = ((y + (TS - c->boff) - 1) / TS - 1) * TS + c->boff;
    hb = drop(w, -1, k, 0, y);
    for (;;)
      { int cb = bseq[y], ca;
        if (cb == 4)
          { more = 0; if (bclip > k) bclip = k; break; }
        ca = a[y];
        if (cb != ca)
          { if (ca == 4) { more = 0; aclip = k; }
            break;
          }
        y -= 1;
      }
    v = (y << 1) + k;
    while (y + k <= na) { ha = drop(w, ha, k, 0, na); na -= TS; }
    while (y <= nb)     { hb = drop(w, hb, k, 0, nb); nb -= TS; }
    if (v < besta)
      { besta = lasta = trim.a = v;
        besty = trim.y = y;
        trim.ha = ha;
        trim.hb = hb;
      }
    w->V[cur][k + o] = v;  w->T[cur][k + o] = HIST_FULL;  w->M[cur][k + o] = HIST_LEN;
    w->HA[cur][k + o] = ha; w->HB[cur][k + o] = hb;
    w->NA[k + o] = na;      w->NB[k + o] = nb;
  }

#define CLIP_REV()                                                                         \
  if (more == 0)                                                                           \
    { if (bseq[besty] != 4 && aseq[besta - besty] != 4)                                    \
        more = 1;                                                                          \
      if (low <= aclip)                                                                    \
        { low = aclip + 1;                                                                 \
          if (reachm <= w->M[cur][aclip + o])                                              \
            { reachm = w->M[cur][aclip + o]; reach.a = w->V[cur][aclip + o];               \
              reach.y = (reach.a - aclip) / 2; reach.d = dif;                              \
              reach.ha = w->HA[cur][aclip + o]; reach.hb = w->HB[cur][aclip + o]; }        \
        }                                                                                  \
      if (hgh >= bclip)                                                                    \
        { hgh = bclip - 1;                                                                 \
          if (reachm <= w->M[cur][bclip + o])                                              \
            { reachm = w->M[cur][bclip + o]; reach.a = w->V[cur][bclip + o];               \
              reach.y = (reach.a - bclip) / 2; reach.d = dif;                              \
              reach.ha = w->HA[cur][bclip + o]; reach.hb = w->HB[cur][bclip + o]; }        \
        }                                                                                  \
      aclip = -INT_MAX;                                                                    \
      bclip = INT_MAX;                                                                     \
    }

  CLIP_REV()

  while (more && lasta <= besta + MAX_TRIM_LAG)
    { int k, nlow, nhgh;

      if (hgh < low)
        { if (c->st) c->st->empty_band = 1;
          break;
        }
      nlow = low - 1;
      nhgh = hgh + 1;
      if (nlow >= c->minp)
        { w->NA[nlow + o] = w->NA[nlow + 1 + o];
          w->NB[nlow + o] = w->NB[nlow + 1 + o];
          w->V[cur][nlow + o] = INT_MAX;
        }
      else
        nlow += 1;
      if (nhgh <= c->maxp)
        { w->NA[nhgh + o] = w->NA[nhgh - 1 + o];
          w->NB[nhgh + o] = w->NB[nhgh - 1 + o];
          w->V[cur][nhgh + o] = INT_MAX;
        }
      else
        nhgh -= 1;
      low = nlow;
      hgh = nhgh;
      w->V[cur][hgh + 1 + o] = w->V[cur][low - 1 + o] = INT_MAX;
      dif += 1;

      if (c->st)
        bandstat_step(c->st, hgh - low + 1);
      for (k = low; k <= hgh; k++)
        { int ac = w->V[cur][k + o], am = w->V[cur][k - 1 + o], ap = w->V[cur][k + 1 + o];
          int from, v, y, m, ha, hb;
          uint64 b;
          const char *a = aseq + k;

          if (ac > ap)
            from = (ap > am) ? k - 1 : k + 1;
          else
            from = (ac > am) ? k - 1 : k;
          v  = (from == k) ? ac - 2 : (from == k - 1 ? am - 1 : ap - 1);
          m  = w->M[cur][from + o];
          b  = w->T[cur][from + o];
          ha = w->HA[cur][from + o];
          hb = w->HB[cur][from + o];

          if (b & HIST_TOP)
            m -= 1;
          b <<= 1;
          y = (v - k) >> 1;
          for (;;)
            { int cb = bseq[y], ca;
              if (cb == 4)
                { more = 0; if (bclip > k) bclip = k; break; }
              ca = a[y];
              if (cb != ca)
                { if (ca == 4) { more = 0; if (k > aclip) aclip = k; }
                  break;
                }
              y -= 1;
              if ((b & HIST_TOP) == 0)
                m += 1;
              b = (b << 1) | 1;
            }
          v = (y << 1) + k;

          while (y + k <= w->NA[k + o])
            { if (w->cells[ha].mark > w->NA[k + o])
                ha = drop(w, ha, k, dif, w->NA[k + o]);
              w->NA[k + o] -= TS;
            }
          while (y <= w->NB[k + o])
            { if (w->cells[hb].mark > w->NB[k + o])
                hb = drop(w, hb, k, dif, w->NB[k + o]);
              w->NB[k + o] -= TS;
            }
          w->V[nxt][k + o] = v;  w->T[nxt][k + o] = b;  w->M[nxt][k + o] = m;
          w->HA[nxt][k + o] = ha; w->HB[nxt][k + o] = hb;
        }

      /* ordered scan, lowest diagonal first (align.c:1620-1637) */
      for (k = low; k <= hgh; k++)
        { int v = w->V[nxt][k + o];
          if (v < besta)
            { uint64 b = w->T[nxt][k + o];
              besta = v;
              besty = (v - k) >> 1;
              if (w->M[nxt][k + o] >= c->ave)
                { lasta = v;
                  if (c->table[b & TRIM_MASK] >= 0 &&
                      c->table[(b >> TRIM_BITS) & TRIM_MASK] + c->score[b & TRIM_MASK] >= 0)
                    { trim.a = v; trim.y = besty; trim.d = dif;
                      trim.ha = w->HA[nxt][k + o]; trim.hb = w->HB[nxt][k + o];
                    }
                }
            }
        }
      cur ^= 1;
      nxt ^= 1;

      CLIP_REV()

      { int n = besta + MAX_WAVE_LAG;
        while (hgh >= low)
          if (w->V[cur][hgh + o] > n)
            hgh -= 1;
          else
            { while (w->V[cur][low + o] > n)
                low += 1;
              break;
            }
      }
      if (c->st)
        { int wd = hgh - low + 1;
          c->st->waves += 1;
          c->st->cells += wd;
          if (wd > c->st->maxband)
            c->st->maxband = wd;
        }
    }
#undef CLIP_REV

  /* start point and prepended traces (align.c:1710-1895) */
  { int  trimx, trimy, trimd, ha, hb;
    int *chain = NULL, cap = 0, n, i;
    int  at = 0, bt = 0;          /* negative write cursors */
    int  b, e, k, a, d;

    if (reachm >= 0 && c->reach)
      { trimx = reach.a - reach.y; trimy = reach.y; trimd = reach.d; ha = reach.ha; hb = reach.hb; }
    else
      { trimx = trim.a - trim.y; trimy = trim.y; trimd = trim.d; ha = trim.ha; hb = trim.hb; }

    /* A view */
    n = unwind(w, ha, &chain, &cap);
    k = w->cells[chain[0]].diag;
    b = w->cells[chain[0]].mark - k;
    e = 0;
    i = 0;                                   /* index of the last cell consumed */
    if ((b + k) % TS != c->aoff)
      { i = 1;
        if (i >= n)
          { a = trimy; d = trimd; i = -1; }
        else
          { k = w->cells[chain[i]].diag;
            a = w->cells[chain[i]].mark - k;
            d = w->cells[chain[i]].diff;
          }
        if (*atlen_io == 0)
          { atrace[--at] = (uint16) (b - a);
            atrace[--at] = (uint16) (d - e);
          }
        else
          { atrace[1] = (uint16) (atrace[1] + (b - a));
            atrace[0] = (uint16) (atrace[0] + (d - e));
          }
        b = a;
        e = d;
      }
    if (i >= 0)
      { for (i = i + 1; i < n; i++)
          { k = w->cells[chain[i]].diag;
            a = w->cells[chain[i]].mark - k;
            d = w->cells[chain[i]].diff;
            atrace[--at] = (uint16) (b - a);
            atrace[--at] = (uint16) (d - e);
            b = a;
            e = d;
          }
        if (b + k != trimx)
          { atrace[--at] = (uint16) (b - trimy);
            atrace[--at] = (uint16) (trimd - e);
          }
        else if (b != trimy && (*atlen_io - at) > 0)
          { atrace[at + 1] = (uint16) (atrace[at + 1] + (b - trimy));
            atrace[at]     = (uint16) (atrace[at] + (trimd - e));
          }
      }

    /* B view */
    n = unwind(w, hb, &chain, &cap);
    k = w->cells[chain[0]].diag;
    b = w->cells[chain[0]].mark + k;
    e = 0;
    i = 0;
    if ((b - k) % TS != c->boff)
      { i = 1;
        if (i >= n)
          { a = trimx; d = trimd; i = -1; }
        else
          { k = w->cells[chain[i]].diag;
            a = w->cells[chain[i]].mark + k;
            d = w->cells[chain[i]].diff;
          }
        if (*btlen_io == 0)
          { btrace[--bt] = (uint16) (b - a);
            btrace[--bt] = (uint16) (b - a);      /* sic: align.c:1843-1844 */
          }
        else
          { btrace[1] = (uint16) (btrace[1] + (b - a));
            btrace[0] = (uint16) (btrace[0] + (d - e));
          }
        b = a;
        e = d;
      }
    if (i >= 0)
      { for (i = i + 1; i < n; i++)
          { k = w->cells[chain[i]].diag;
            a = w->cells[chain[i]].mark + k;
            d = w->cells[chain[i]].diff;
            btrace[--bt] = (uint16) (b - a);
            btrace[--bt] = (uint16) (d - e);
            b = a;
            e = d;
          }
        if (b - k != trimy)
          { btrace[--bt] = (uint16) (b - trimx);
            btrace[--bt] = (uint16) (trimd - e);
          }
        else if (b != trimx && (*btlen_io - bt) > 0)
          { btrace[bt + 1] = (uint16) (btrace[bt + 1] + (b - trimx));
            btrace[bt]     = (uint16) (btrace[bt] + (trimd - e));
          }
      }
    free(chain);

    apath->abpos = trimx;
    apath->bbpos = trimy;
    apath->diffs += trimd;
    *atlen_io -= at;
    *btlen_io -= bt;
    *aback = -at;
    *bback = -bt;
    if (c->st)
      c->st->pebbles += w->ncell;
  }
}

/***** Local_Alignment: align.c:1904-2097 **************************************************/

void oracle_local_alignment(const char *aseq, int alen, const char *bseq, int blen,
                            uint32 flags, int diag, int anti, Align_Spec *spec,
                            Path *apath, Path *bpath, uint16 *atrace, uint16 *btrace,
                            OWaveStats *stats)
{ Ctx   c;
  Band  w;
  int   maxtp = 2 * (((alen < blen) ? blen : alen) / Trace_Spacing(spec) + 2);
  uint16 *abuf = (uint16 *) malloc(sizeof(uint16) * (size_t) (2 * maxtp + 4));
  uint16 *bbuf = (uint16 *) malloc(sizeof(uint16) * (size_t) (2 * maxtp + 4));
  uint16 *amid = abuf + maxtp + 2, *bmid = bbuf + maxtp + 2;
  int   atlen, btlen, aback, bback, selfie;

  c.aseq = aseq;  c.bseq = bseq;
  c.ts   = Trace_Spacing(spec);
  c.ave  = damar_spec_ave_path(spec);
  c.reach = Overlap_If_Possible(spec);
  c.score = damar_spec_score_table(spec);
  c.table = damar_spec_trim_table(spec);
  c.st   = stats;

  selfie = (aseq == bseq);
  c.minp = (selfie && diag >= 0) ? 1 : -INT_MAX;     /* align.c:1949-1968 with no borders */
  c.maxp = (selfie && diag <= 0) ? -1 : INT_MAX;
  if (COMP(flags))
    { c.aoff = 0; c.boff = blen % c.ts; }            /* align.c:1975-1979 */
  else
    { c.aoff = 0; c.boff = 0; }

  band_init(&w, alen, blen);
  forward(&c, &w, diag, anti, apath, amid, &atlen, bmid, &btlen);
  reverse(&c, &w, diag, anti, apath, amid, &atlen, &aback, bmid, &btlen, &bback);
  band_free(&w);

  apath->tlen = atlen;
  bpath->tlen = btlen;
  memcpy(atrace, amid - aback, sizeof(uint16) * (size_t) atlen);
  memcpy(btrace, bmid - bback, sizeof(uint16) * (size_t) btlen);
  apath->trace = atrace;
  bpath->trace = btrace;
  free(abuf);
  free(bbuf);

  bpath->diffs = apath->diffs;
  if (COMP(flags))
    { int i, j;                                      /* align.c:2033-2056 */
      bpath->abpos = blen - apath->bepos;
      bpath->bbpos = alen - apath->aepos;
      bpath->aepos = blen - apath->bbpos;
      bpath->bepos = alen - apath->abpos;
      for (i = btlen - 2, j = 0; j < i; i -= 2, j += 2)
        { uint16 p = btrace[i];     btrace[i] = btrace[j];         btrace[j] = p;
          p = btrace[i + 1];        btrace[i + 1] = btrace[j + 1]; btrace[j + 1] = p;
        }
    }
  else
    { bpath->aepos = apath->bepos;                   /* align.c:2057-2063 */
      bpath->bepos = apath->aepos;
      bpath->abpos = apath->bbpos;
      bpath->bbpos = apath->abpos;
    }
}

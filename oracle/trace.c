/* trace.c -- TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of the trace-point expansion that is the
 * next consumer of daligner's records:
 *
 *   Compute_Trace_PTS   dalign/align.c:5577-5692   walk the trace-point segments of one record
 *   iter_np             dalign/align.c:4892-5261   O(np) furthest-reaching waves of one segment with every
 *                                                   wave kept, path reversal, edit-script emission
 *
 * Pinned: oracle/_ref/ref_lastrace (our driver around the REAL Compute_Trace_PTS) and oracle_lastrace (this
 * file) write byte-identical dumps for every golden .las in all three modes (tests/test_trace_oracle.py).
 *
 * Geometry of one segment: A[0..M) against B[0..N), del = M - N, diagonal k = i - j.  A wave D holds, for
 * the diagonals low..hgh, the furthest B index j reached with "D half-costs": a substitution costs one
 * wave, an indel that moves away from diagonal del costs two, an indel towards del is free, so the edit
 * distance is D + |del| when diagonal del reaches N.  Each cell remembers which neighbour it came from:
 *     0  same diagonal, wave D-1 (substitution)      -1 / 1   diagonal k-1 / k+1, wave D-2
 *     2  diagonal k-1, same wave                      4        diagonal k+1, same wave
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct
{ int *vf, *hf;      /* (dmax + 3) rows of `width` cells, row r = wave r - 2, column = k - kmin */
  int  width, kmin;
} Waves;

#define VF(w, D, k) ((w)->vf[((D) + 2) * (w)->width + ((k) - (w)->kmin)])
#define HF(w, D, k) ((w)->hf[((D) + 2) * (w)->width + ((k) - (w)->kmin)])

/* the three-way choice of align.c:4981-5004 (FS_MOVE): ties go to the `p` neighbour, then to the
   substitution; returns the B index to continue from and the code of the neighbour chosen */
static inline int choose(int am, int ac, int ap, int mcode, int pcode, int *code)
{ if (ac < am)
    { if (ap < am) { *code = mcode; return am; }
      *code = pcode;
      return ap;
    }
  if (ap < ac) { *code = 0; return ac; }
  *code = pcode;
  return ap;
}

static inline int slide(const char *a, const char *B, int j, int i, int N)
{ const int lim = N < i ? N : i;
  while (j < lim && B[j] == a[j])
    j += 1;
  return j;
}

/* The waves of one segment (align.c:4892-5040, identical in middle_np :5263-5411).  same = the two reads are
   one buffer (align.c:4933-4951), a0 / b0 = offsets of the segment in them.  Returns the wave D on which
   diagonal del reached N, or -1 where the reference exits. */
static int forward_np(const char *A, int M, const char *B, int N, int a0, int b0, int same, Waves *w, int dmax)
{ const int del = M - N;
  int low = del < 0 ? del : 0;
  int hgh = del < 0 ? 0 : del;
  int posl = -dmax, posh = dmax;
  int D, k;

  if (same)
    { const int off = b0 - a0;                       /* B - A as pointers */
      if (off == 0)
        { fprintf(stderr, "oracle: self comparison starts on diagonal 0 (Compute_Trace)\n");
          return -1;
        }
      if (off < 0) { if (off + 1 > posl) posl = off + 1; }
      else         { if (off - 1 < posh) posh = off - 1; }
    }

  for (k = low - 1; k <= hgh + 1; k++)
    VF(w, -2, k) = VF(w, -1, k) = -2;
  VF(w, -1, 0) = -1;
  low += 1;
  hgh -= 1;

  for (D = 0; ; D++)
    { int j, code;

      if (D > dmax)
        return -1;                                   /* align.c:4966-4970: the reference exits here */
      if ((D & 1) == 0)
        { if (low > posl) low -= 1;
          if (hgh < posh) hgh += 1;
        }
      VF(w, D, hgh + 1) = VF(w, D, low - 1) = -2;

      j = -2;                                        /* above del: the free move comes from k+1 */
      for (k = hgh; k > del; k--)
        { j = choose(VF(w, D - 2, k - 1), VF(w, D - 1, k) + 1, j + 1, -1, 4, &code);
          HF(w, D, k) = code;
          VF(w, D, k) = j = slide(A + k, B, j, M - k, N);
        }
      j = -2;                                        /* below del: the free move comes from k-1 */
      for (k = low; k < del; k++)
        { j = choose(j, VF(w, D - 1, k) + 1, VF(w, D - 2, k + 1) + 1, 2, 1, &code);
          HF(w, D, k) = code;
          VF(w, D, k) = j = slide(A + k, B, j, M - k, N);
        }
      j = choose(j, VF(w, D - 1, del) + 1, VF(w, D, del + 1) + 1, 2, 4, &code);
      HF(w, D, del) = code;
      VF(w, D, del) = j = slide(A + del, B, j, N, N);
      if (j >= N)
        return D;
    }
}

/* One backward edge from cell (*pD, k) whose predecessor code is e: the cell it came from and, for
   UPPERMOST / LOWERMOST, the re-routing of align.c:5056-5120 / 5122-5186 (the same text in middle_np).
   *pc is the running B index of the reference's `c`.  Returns the predecessor's diagonal, *pe the code
   of the edge actually taken. */
static int back_edge(const char *A, const char *B, Waves *w, int mode, int del, int *pD, int k, int *pe, int *pc)
{ int e = *pe, D = *pD, c = *pc, h, m;

  h = k + e;
  if (e > 1) h -= 3;
  else if (e == 0) D -= 1;
  else D -= 2;

  if (mode == 1 && h < k)
    { const char *a = A + k;
      m = k < 0 ? -k : 0;
      if (VF(w, D, h) <= c) c = VF(w, D, h) - 1;
      while (c >= m && a[c] == B[c]) c -= 1;
      if (e == -1)
        { if (c <= VF(w, D + 2, k + 1))    { e = 4; h = k + 1; D = D + 2; }
          else if (c == VF(w, D + 1, k))   { e = 0; h = k;     D = D + 1; }
          else VF(w, D, h) = c + 1;
        }
      else
        { m = (k == del) ? D : D - 2;
          if (c <= VF(w, m, k + 1))        { e = (k == del) ? 4 : 1; h = k + 1; D = m; }
          else if (c == VF(w, D - 1, k))   { e = 0; h = k; D = D - 1; }
          else VF(w, D, h) = c + 1;
        }
    }
  else if (mode == -1 && h > k)
    { const char *a = A + k;
      m = k < 0 ? -k : 0;
      if (VF(w, D, h) < c) c = VF(w, D, h);
      while (c >= m && a[c] == B[c]) c -= 1;
      if (e == 1)
        { if (c < VF(w, D + 2, k - 1))     { e = 2; h = k - 1; D = D + 2; }
          else if (c == VF(w, D + 1, k))   { e = 0; h = k;     D = D + 1; }
          else VF(w, D, h) = c--;
        }
      else
        { m = (k == del) ? D : D - 2;
          if (c < VF(w, m, k - 1))         { e = (k == del) ? 2 : -1; h = k - 1; D = m; }
          else if (c == VF(w, D - 1, k))   { e = 0; h = k; D = D - 1; }
          else VF(w, D, h) = c--;
        }
    }
  *pe = e;  *pD = D;  *pc = c;
  return h;
}

/* iter_np (align.c:4892-5261): waves, link reversal, edit script appended to script[*ns...]; returns
   D + |del| or -1. */
static int segment_np(const char *A, int M, const char *B, int N, int a0, int b0, int same,
                      Waves *w, int mode, int dmax, int *script, int *ns)
{ const int del = M - N;
  int D = forward_np(A, M, B, N, a0, b0, same, w, dmax);
  int k, e, h, m, c;

  if (D < 0)
    return -1;

  /* reverse the predecessor links into successor links, from (D, del) back to (0, 0) */
  HF(w, 0, 0) = 3;
  c = N;
  k = del;
  e = HF(w, D, k);
  HF(w, D, k) = 3;
  while (e != 3)
    { h = back_edge(A, B, w, mode, del, &D, k, &e, &c);
      m = HF(w, D, h);
      HF(w, D, h) = e;
      e = m;
      k = h;
    }

  /* forward along the successor links: one script value per indel (align.c:5217-5256) */
  k = D = 0;
  e = HF(w, 0, 0);
  while (e != 3)
    { h = k - e;
      c = VF(w, D, k);
      if (e > 1) h += 3;
      else if (e == 0) D += 1;
      else D += 2;
      if (h > k)      script[(*ns)++] = b0 + c + 1;
      else if (h < k) script[(*ns)++] = -(a0 + c + k + 1);
      k = h;
      e = HF(w, D, h);
    }
  return D + abs(del);
}

/* middle_np (align.c:5263-5573): the same waves, then ceil((D + |del|) / 2) edges back from the end; the
   furthest point of the cell reached is the mid point (offsets into the two reads).  Returns 0 or -1. */
static int middle_np(const char *A, int M, const char *B, int N, int a0, int b0, int same,
                     Waves *w, int mode, int dmax, int *mida, int *midb)
{ const int del = M - N;
  int D = forward_np(A, M, B, N, a0, b0, same, w, dmax);
  int k = del, c = N, d, f, e;

  if (D < 0)
    return -1;
  d = D + abs(del);
  for (f = d / 2; d > f; d--)
    { e = HF(w, D, k);
      k = back_edge(A, B, w, mode, del, &D, k, &e, &c);
    }
  *midb = b0 + VF(w, D, k);
  *mida = a0 + k + VF(w, D, k);
  return 0;
}

int oracle_compute_trace_pts(const char *aseq, int alen, const char *bseq, int blen, const Path *path,
                             int tspace, int mode, int *script, int *diffs)
{ const uint16 *pts = (const uint16 *) path->trace;
  int tlen = path->tlen;
  int dmax = 0, nmax = 0, d, i, ns = 0, total = 0;
  int ab, ae, bb, be;
  Waves w;

  for (d = 1; d < tlen; d += 2)
    { if (pts[d - 1] > dmax) dmax = pts[d - 1];
      if (pts[d] > nmax) nmax = pts[d];
    }
  if (tlen <= 1)
    nmax = path->bepos - path->bbpos;
  /* diagonals used: min(0,del) - dmax/2 - 2 .. max(0,del) + dmax/2 + 2 with |del| <= tspace + nmax */
  w.kmin  = -(tspace + nmax) - dmax / 2 - 3;
  w.width = 2 * (tspace + nmax + dmax / 2 + 3) + 1;
  w.vf = (int *) malloc(sizeof(int) * (size_t) w.width * (dmax + 3));
  w.hf = (int *) malloc(sizeof(int) * (size_t) w.width * (dmax + 3));
  if (w.vf == NULL || w.hf == NULL)
    { fprintf(stderr, "oracle: out of memory (trace waves)\n");
      exit(1);
    }

  ab = path->abpos;
  ae = (ab / tspace) * tspace;
  bb = path->bbpos;
  for (i = 1; i < tlen - 2; i += 2)
    { ae += tspace;
      be = bb + pts[i];
      if (ae > alen || be > blen)
        goto bad;
      d = segment_np(aseq + ab, ae - ab, bseq + bb, be - bb, ab, bb, aseq == bseq, &w, mode, dmax, script, &ns);
      if (d < 0)
        goto bad;
      total += d;
      ab = ae;
      bb = be;
    }
  ae = path->aepos;
  be = path->bepos;
  if (ae > alen || be > blen)
    goto bad;
  d = segment_np(aseq + ab, ae - ab, bseq + bb, be - bb, ab, bb, aseq == bseq, &w, mode, dmax, script, &ns);
  if (d < 0)
    goto bad;
  total += d;
  free(w.vf);
  free(w.hf);
  *diffs = total;
  return ns;

bad:
  free(w.vf);
  free(w.hf);
  return -1;
}

/* align.c:5694-5830: the edit script between the MID points of the trace-point segments.  Every segment
   gives a mid point (middle_np), the script is computed between successive mid points (iter_np), so no
   piece ends on a trace point.  *diffs reproduces the reference's sum, in which the distance of the piece
   that ends on the last mid point is added twice (align.c:5812-5822: `d += iter_np(...); diffs += d`). */
int oracle_compute_trace_mid(const char *aseq, int alen, const char *bseq, int blen, const Path *path,
                             int tspace, int mode, int *script, int *diffs)
{ const uint16 *pts = (const uint16 *) path->trace;
  int tlen = path->tlen;
  int dmax = 0, nmax = 0, d, i, ns = 0, total = 0;
  int ab, ae, bb, be, as, bs, af, bf;
  const int same = (aseq == bseq);
  Waves w;

  for (d = 1; d < tlen; d += 2)
    { if (pts[d - 1] > dmax) dmax = pts[d - 1];
      if (pts[d] > nmax) nmax = pts[d];
    }
  if (tlen <= 1)
    nmax = path->bepos - path->bbpos;
  /* a piece between mid points can be as long as two segments */
  w.kmin  = -2 * (tspace + nmax) - dmax / 2 - 3;
  w.width = 2 * (2 * (tspace + nmax) + dmax / 2 + 3) + 1;
  w.vf = (int *) malloc(sizeof(int) * (size_t) w.width * (dmax + 3));
  w.hf = (int *) malloc(sizeof(int) * (size_t) w.width * (dmax + 3));
  if (w.vf == NULL || w.hf == NULL)
    { fprintf(stderr, "oracle: out of memory (trace waves)\n");
      exit(1);
    }

  ab = as = af = path->abpos;
  ae = (ab / tspace) * tspace;
  bb = bs = bf = path->bbpos;
  d = 0;
  for (i = 1; i < tlen - 2; i += 2)
    { ae += tspace;
      be = bb + pts[i];
      if (ae > alen || be > blen)
        goto bad;
      if (middle_np(aseq + ab, ae - ab, bseq + bb, be - bb, ab, bb, same, &w, mode, dmax, &af, &bf))
        goto bad;
      d = segment_np(aseq + as, af - as, bseq + bs, bf - bs, as, bs, same, &w, mode, dmax, script, &ns);
      if (d < 0)
        goto bad;
      total += d;
      ab = ae;  bb = be;
      as = af;  bs = bf;
    }
  ae = path->aepos;
  be = path->bepos;
  if (ae > alen || be > blen)
    goto bad;
  if (middle_np(aseq + ab, ae - ab, bseq + bb, be - bb, ab, bb, same, &w, mode, dmax, &af, &bf))
    goto bad;
  d = segment_np(aseq + as, af - as, bseq + bs, bf - bs, as, bs, same, &w, mode, dmax, script, &ns);
  if (d < 0)
    goto bad;
  total += d;
  as = af;  bs = bf;
  { const int last = segment_np(aseq + af, ae - as, bseq + bf, be - bs, af, bf, same, &w, mode, dmax, script, &ns);
    d += last;                                       /* the reference tests this sum, not `last`, for < 0 */
    if (d < 0)
      goto bad;
    total += d;
  }
  free(w.vf);
  free(w.hf);
  *diffs = total;
  return ns;

bad:
  free(w.vf);
  free(w.hf);
  return -1;
}

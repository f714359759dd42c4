/* oracle_lastrace.c -- TEST INFRASTRUCTURE.  Same command line and dump format as ref_lastrace.c, with the
 * oracle's restatement (trace.c) in place of the reference's Compute_Trace_PTS:
 *
 *     oracle_lastrace <db root> <file.las> <out.bin> [mode [mid]]        (mid: Compute_Trace_MID)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define OVL_IO ((int) (sizeof(Overlap) - sizeof(void *)))

int main(int argc, char *argv[])
{ HITS_DB db;
  FILE   *in, *out;
  int64   novl, i;
  int     tspace, tbytes, mode = 0, mid = 0, j;
  Overlap ovl;
  uint16 *pts = NULL;
  int    *script = NULL;
  int     pmax = 0, smax = 0;
  char   *bbuf;

  if (argc < 4)
    { fprintf(stderr, "usage: oracle_lastrace <db> <las> <out> [mode]\n");
      return 1;
    }
  if (argc > 4) mode = atoi(argv[4]);
  if (argc > 5) mid = (strcmp(argv[5], "mid") == 0);
  if (damar_read_block(argv[1], &db)) return 1;
  if ((in = fopen(argv[2], "rb")) == NULL || (out = fopen(argv[3], "wb")) == NULL)
    { fprintf(stderr, "oracle_lastrace: cannot open files\n");
      return 1;
    }
  if (fread(&novl, sizeof(int64), 1, in) != 1 || fread(&tspace, sizeof(int), 1, in) != 1) return 1;
  tbytes = (tspace <= TRACE_XOVR) ? 1 : 2;
  { int32_t h[2] = { tspace, mode };
    fwrite(h, sizeof(int32_t), 2, out);
    fwrite(&novl, sizeof(int64), 1, out);
  }
  bbuf = (char *) malloc((size_t) db.maxlen + 4);
  for (i = 0; i < novl; i++)
    { int32_t rec[5];
      int     alen, blen, ns, diffs, need;
      const char *aseq, *bseq;

      if (fread(((char *) &ovl) + sizeof(void *), OVL_IO, 1, in) != 1) return 1;
      if (ovl.path.tlen > pmax)
        { pmax = 2 * ovl.path.tlen + 1000;
          pts = (uint16 *) realloc(pts, sizeof(uint16) * (size_t) pmax);
        }
      if (tbytes == 1)
        { uint8 *t8 = (uint8 *) pts;
          if (ovl.path.tlen > 0 && fread(t8, 1, (size_t) ovl.path.tlen, in) != (size_t) ovl.path.tlen) return 1;
          for (j = ovl.path.tlen - 1; j >= 0; j--)          /* Decompress_TraceTo16, align.c:3398 */
            pts[j] = t8[j];
        }
      else if (ovl.path.tlen > 0 && fread(pts, 2, (size_t) ovl.path.tlen, in) != (size_t) ovl.path.tlen)
        return 1;
      ovl.path.trace = pts;
      alen = db.reads[ovl.aread].rlen;
      blen = db.reads[ovl.bread].rlen;
      aseq = (char *) db.bases + db.reads[ovl.aread].boff;
      bseq = (char *) db.bases + db.reads[ovl.bread].boff;
      if (ovl.flags & COMP_FLAG)
        { bbuf[0] = 4;
          for (j = 0; j < blen; j++)                          /* Complement_Seq, align.c:3587 */
            bbuf[1 + j] = (char) (3 - bseq[blen - 1 - j]);
          bbuf[1 + blen] = 4;
          bseq = bbuf + 1;
        }
      else if (ovl.aread == ovl.bread)
        { bbuf[0] = 4;                                        /* LAshow loads B into its own buffer */
          memcpy(bbuf + 1, bseq, (size_t) blen + 1);
          bseq = bbuf + 1;
        }
      need = (ovl.path.aepos - ovl.path.abpos) + (ovl.path.bepos - ovl.path.bbpos) + 16;
      if (need > smax)
        { smax = 2 * need;
          script = (int *) realloc(script, sizeof(int) * (size_t) smax);
        }
      ns = mid ? oracle_compute_trace_mid(aseq, alen, bseq, blen, &ovl.path, tspace, mode, script, &diffs)
               : oracle_compute_trace_pts(aseq, alen, bseq, blen, &ovl.path, tspace, mode, script, &diffs);
      if (ns < 0)
        { fprintf(stderr, "oracle_lastrace: bad alignment between trace points, record %lld\n", (long long) i);
          return 1;
        }
      rec[0] = ovl.aread;  rec[1] = ovl.bread;  rec[2] = (int32_t) ovl.flags;  rec[3] = diffs;  rec[4] = ns;
      fwrite(rec, sizeof(int32_t), 5, out);
      fwrite(script, sizeof(int), (size_t) ns, out);
    }
  fclose(out);
  fclose(in);
  return 0;
}

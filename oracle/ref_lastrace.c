/* ref_lastrace.c -- TEST INFRASTRUCTURE ONLY.  A small driver of OUR OWN that is linked against the REAL
 * reference (dalign/align.c, db/DB.c compiled where they lie under /root/reference by oracle/Makefile.ref,
 * output oracle/_ref/ref_lastrace) and calls the reference's Compute_Trace_PTS (align.c:5577) on every
 * record of a .las file exactly the way utils/LAshow.c:245-262 does (load A, load B, complement B when
 * OVL_COMP is set, trace points widened to 16 bits).  It dumps what the call leaves in the Path:
 *
 *     int32 tspace, int32 mode, int64 novl, then per record
 *     int32 aread, bread, flags, diffs, tlen, followed by tlen int32 edit-script values
 *
 * The oracle's restatement (oracle/trace.c, oracle_lastrace) writes the same format; tests compare the two
 * byte for byte, and the md5 of these dumps for the golden cases is committed under tests/golden/.
 *
 *     ref_lastrace <db root> <file.las> <out.bin> [mode: -1 LOWERMOST | 0 GREEDIEST | 1 UPPERMOST [mid]]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "db/DB.h"
#include "dalign/align.h"

int main(int argc, char *argv[])
{ HITS_DB db;
  FILE   *in, *out;
  int64   novl, i;
  int     tspace, tbytes, mode = 0, mid = 0, tmax = 0;
  Overlap ovl;
  Alignment aln;
  Work_Data *work;
  char *abuf, *bbuf;
  int   lasta = -1;

  if (argc < 4)
    { fprintf(stderr, "usage: ref_lastrace <db> <las> <out> [mode]\n");
      return 1;
    }
  if (argc > 4) mode = atoi(argv[4]);
  if (argc > 5) mid = (strcmp(argv[5], "mid") == 0);          /* Compute_Trace_MID, as corrector/LAcorrect.c:545 */
  if (Open_DB(argv[1], &db) < 0) return 1;
  if ((in = fopen(argv[2], "rb")) == NULL || (out = fopen(argv[3], "wb")) == NULL)
    { fprintf(stderr, "ref_lastrace: cannot open files\n");
      return 1;
    }
  if (fread(&novl, sizeof(int64), 1, in) != 1 || fread(&tspace, sizeof(int), 1, in) != 1) return 1;
  tbytes = (tspace <= TRACE_XOVR) ? 1 : 2;
  { int32_t h[2] = { tspace, mode };
    fwrite(h, sizeof(int32_t), 2, out);
    fwrite(&novl, sizeof(int64), 1, out);
  }
  work = New_Work_Data();
  abuf = New_Read_Buffer(&db);
  bbuf = New_Read_Buffer(&db);
  aln.aseq = abuf;
  aln.bseq = bbuf;
  ovl.path.trace = NULL;
  for (i = 0; i < novl; i++)
    { int32_t rec[5];
      if (Read_Overlap(in, &ovl)) return 1;
      if (ovl.path.tlen > tmax)
        { tmax = ovl.path.tlen * 2 + 1000;
          ovl.path.trace = realloc(ovl.path.trace, sizeof(uint16) * tmax);
        }
      if (Read_Trace(in, &ovl, tbytes)) return 1;
      if (tbytes == 1) Decompress_TraceTo16(&ovl);
      if (ovl.aread != lasta)
        { Load_Read(&db, ovl.aread, abuf, 0);
          lasta = ovl.aread;
        }
      Load_Read(&db, ovl.bread, bbuf, 0);
      aln.alen = db.reads[ovl.aread].rlen;
      aln.blen = db.reads[ovl.bread].rlen;
      aln.flags = ovl.flags;
      aln.path = &ovl.path;
      if (ovl.flags & COMP_FLAG) Complement_Seq(bbuf, aln.blen);
      { void *keep = ovl.path.trace;                 /* the call redirects path.trace into the work data */
        if (mid ? Compute_Trace_MID(&aln, work, tspace, mode) : Compute_Trace_PTS(&aln, work, tspace, mode)) return 1;
        rec[0] = ovl.aread;  rec[1] = ovl.bread;  rec[2] = ovl.flags;
        rec[3] = ovl.path.diffs;  rec[4] = ovl.path.tlen;
        fwrite(rec, sizeof(int32_t), 5, out);
        fwrite(ovl.path.trace, sizeof(int), ovl.path.tlen, out);
        ovl.path.trace = keep;
      }
    }
  fclose(out);
  fclose(in);
  return 0;
}

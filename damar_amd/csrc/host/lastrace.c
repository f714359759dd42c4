/* lastrace.c -- host driver of the MI355X trace-point expansion: for every record of a
 * .las file computes what the reference's Compute_Trace_PTS (align.c:5577) leaves in the record's Path when
 * utils/LAshow.c:245-262 calls it -- the edit script of the alignment and its difference count -- with
 * damar_trace_pts of libdamar_hip.so, and writes them as a flat binary file:
 *
 *     int32 tspace, int32 mode, int64 novl, then per record
 *     int32 aread, bread, flags, diffs, tlen, followed by tlen int32 script values
 *
 *     lastrace [-g<gpu>] [-m<-1|0|1>] [-M] <A block or DB> <B block or DB> <file.las> <out.bin>
 *
 * -M computes Compute_Trace_MID (align.c:5694, the corrector's variant) with damar_trace_mid instead.
 * -L<jobs> runs one such job per line of a file in one process (the GPU is initialised once).
 *
 * A and B name the blocks (or whole DBs) that hold the A and the B reads of the file.  Host code stays C.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "damar_filter.h"
#include "damar_hip.h"

#define OVL_IO ((int) (sizeof(Overlap) - sizeof(void *)))

static int verbose = 0, reps = 1;         /* -R<n>: repeat the expansion (timing: the first call allocates) */

/* one file: A / B block names, .las, output, mode, mid (0 = Compute_Trace_PTS, 1 = Compute_Trace_MID) */
static void run_job(const char *aname, const char *bname, const char *lasname, const char *outname, int mode, int mid)
{ int     tspace, tbytes, same_block;
  int64   novl, i, ptop = 0, pmax;
  HITS_DB adb, bdb;
  damar_dev_block *ablk, *bblk;
  FILE   *in, *out;
  Overlap *ovls;
  uint8  *pts;
  int64  *soff;
  int    *diffs, *script = NULL;

  same_block = (strcmp(aname, bname) == 0);
  if (damar_read_block(aname, &adb)) exit(1);
  if (!same_block && damar_read_block(bname, &bdb)) exit(1);
  if ((in = fopen(lasname, "rb")) == NULL)
    { fprintf(stderr, "lastrace: cannot open %s\n", lasname);
      exit(1);
    }
  if (fread(&novl, sizeof(int64), 1, in) != 1 || fread(&tspace, sizeof(int), 1, in) != 1 || novl < 0 || tspace <= 0)
    { fprintf(stderr, "lastrace: %s is not a .las file\n", lasname);
      exit(1);
    }
  tbytes = (tspace <= TRACE_XOVR) ? 1 : 2;
  ovls = (Overlap *) malloc(sizeof(Overlap) * (size_t) (novl + 1));
  pmax = 1 << 20;
  pts  = (uint8 *) malloc((size_t) pmax);
  for (i = 0; i < novl; i++)
    { int64 n;
      if (fread(((char *) (ovls + i)) + sizeof(void *), OVL_IO, 1, in) != 1)
        { fprintf(stderr, "lastrace: %s is truncated\n", lasname);
          exit(1);
        }
      n = (int64) ovls[i].path.tlen * tbytes;
      if (ptop + n > pmax)
        { pmax = 2 * (ptop + n);
          pts = (uint8 *) realloc(pts, (size_t) pmax);
        }
      if (n > 0 && fread(pts + ptop, (size_t) n, 1, in) != 1)
        { fprintf(stderr, "lastrace: %s is truncated\n", lasname);
          exit(1);
        }
      ovls[i].path.trace = (void *) (uintptr_t) ptop;       /* offset now, pointer once pts stops moving */
      ptop += n;
    }
  fclose(in);
  for (i = 0; i < novl; i++)
    ovls[i].path.trace = pts + (uintptr_t) ovls[i].path.trace;

  ablk = damar_block_upload(&adb);
  bblk = same_block ? ablk : damar_block_upload(&bdb);
  soff  = (int64 *) malloc(sizeof(int64) * (size_t) (novl + 1));
  diffs = (int *) malloc(sizeof(int) * (size_t) (novl + 1));
  { int rep;
    for (rep = 0; rep < reps; rep++)
      { free(script);
        script = NULL;
        if ((mid ? damar_trace_mid : damar_trace_pts)(ablk, adb.ufirst, bblk, same_block ? adb.ufirst : bdb.ufirst, ovls, novl,
                                                     tbytes, tspace, mode, 0, soff, diffs, &script))
          exit(1);
        if (verbose)
          { double ms[4];
            int64  cnt[4];
            damar_trace_last(ms, cnt);
            printf("lastrace: %lld records, %lld segments (%lld deferred), %lld script values; waves %.2f ms, device %.2f ms, call %.2f ms\n",
                   (long long) cnt[0], (long long) cnt[1], (long long) cnt[2], (long long) cnt[3], ms[0], ms[1], ms[2]);
          }
      }
  }

  if ((out = fopen(outname, "wb")) == NULL)
    { fprintf(stderr, "lastrace: cannot create %s\n", outname);
      exit(1);
    }
  { int32_t h[2] = { tspace, mode };
    fwrite(h, sizeof(int32_t), 2, out);
    fwrite(&novl, sizeof(int64), 1, out);
  }
  for (i = 0; i < novl; i++)
    { int32_t rec[5];
      rec[0] = ovls[i].aread;  rec[1] = ovls[i].bread;  rec[2] = (int32_t) ovls[i].flags;
      rec[3] = diffs[i];  rec[4] = (int32_t) (soff[i + 1] - soff[i]);
      fwrite(rec, sizeof(int32_t), 5, out);
      fwrite(script + soff[i], sizeof(int), (size_t) rec[4], out);
    }
  fclose(out);
  free(script);  free(soff);  free(diffs);  free(ovls);  free(pts);
  damar_block_free(ablk);
  if (!same_block)
    { damar_block_free(bblk);
      damar_close_block(&bdb);
    }
  damar_close_block(&adb);
}

int main(int argc, char *argv[])
{ int   c, gpu = -1, mode = GREEDIEST, mid = 0;
  char *list = NULL;

  opterr = 0;
  while ((c = getopt(argc, argv, "vMg:m:L:R:")) != -1)
    switch (c)
    { case 'g': gpu = atoi(optarg); break;
      case 'm': mode = atoi(optarg); break;
      case 'v': verbose = 1; break;
      case 'M': mid = 1; break;
      case 'L': list = optarg; break;
      case 'R': reps = atoi(optarg) > 0 ? atoi(optarg) : 1; break;
      default:
        fprintf(stderr, "Unsupported option: %s\n", argv[optind - 1]);
        exit(1);
    }
  if ((list == NULL && argc - optind != 4) || (list != NULL && argc != optind) || mode < -1 || mode > 1)
    { fprintf(stderr, "usage: lastrace [-v] [-M] [-g<gpu>] [-m<-1|0|1>] <A block> <B block> <file.las> <out.bin>\n"
                      "       lastrace [-v] [-g<gpu>] -L<jobs>     (lines: <mode> <mid 0|1> <A block> <B block> <file.las> <out.bin>)\n");
      exit(1);
    }
  if (gpu >= 0)
    damar_hip_init(gpu);
  if (list == NULL)
    run_job(argv[optind], argv[optind + 1], argv[optind + 2], argv[optind + 3], mode, mid);
  else
    { FILE *f = fopen(list, "r");
      char  a[2048], b[2048], l[2048], o[2048];
      if (f == NULL)
        { fprintf(stderr, "lastrace: cannot open %s\n", list);
          exit(1);
        }
      while (fscanf(f, " %d %d %2047s %2047s %2047s %2047s", &mode, &mid, a, b, l, o) == 6)
        { if (mode < -1 || mode > 1)
            { fprintf(stderr, "lastrace: bad mode in %s\n", list);
              exit(1);
            }
          run_job(a, b, l, o, mode, mid != 0);
        }
      fclose(f);
    }
  return 0;
}

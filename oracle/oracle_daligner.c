/* oracle_daligner.c -- ORACLE (test infrastructure): command-line driver of the CPU
 * restatement, same options, block-pair loop and output files as the reference's
 * dalign/daligner.c:662-1077 (-b, -D not restated).  Used by tests/ and by
 * bench.py's cpu_baseline leg only. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>

#include "oracle.h"
#include "../damar_amd/csrc/host/damar_host.h"

static void make_dir(const char *d)
{ struct stat s;
  if (stat(d, &s) != 0)
    mkdir(d, 0755);
}

int main(int argc, char *argv[])
{ OParams prm;
  double  ecorr = .70;
  int     spacing = 100, runid = 1, nthreads = 4, notrace = 0, only_id = 0, verbose = 0;
  int     c, i;
  HITS_DB ablock, bblock, *cblock;
  char   *afile, *aroot;
  OKmer  *aidx, *bidx;
  int     alen, blen;
  Align_Spec *spec;
  OWaveStats st;
  int64   cnt[3];
  char   *mask[64];
  int     mtop = 0;

  memset(&prm, 0, sizeof(prm));
  memset(&st, 0, sizeof(st));
  prm.kmer = 14; prm.binshift = 6; prm.hitmin = 35; prm.minover = 1000; prm.symmetric = 1;
  prm.mem_limit = (int64) sysconf(_SC_PHYS_PAGES) * (int64) sysconf(_SC_PAGESIZE);

  opterr = 0;
  while ((c = getopt(argc, argv, "vbOTAIk:w:h:t:M:e:l:s:H:D:m:r:j:")) != -1)
    switch (c)
    { case 'v': verbose = 1; break;
      case 'T': notrace = 1; break;
      case 'I': prm.identity = 1; break;
      case 'O': prm.identity = 1; only_id = 1; break;
      case 'A': prm.symmetric = 0; break;
      case 'k': prm.kmer = atoi(optarg); break;
      case 'w': prm.binshift = atoi(optarg); break;
      case 'h': prm.hitmin = atoi(optarg); break;
      case 't': prm.suppress = atoi(optarg); break;
      case 'H': break;                                   /* no effect, SURVEY App. A.1 */
      case 'e': ecorr = atof(optarg); break;
      case 'l': prm.minover = atoi(optarg); break;
      case 's': spacing = atoi(optarg); break;
      case 'j': nthreads = atoi(optarg); break;
      case 'M': prm.mem_limit = atoi(optarg) * 0x40000000ll; break;
      case 'r': runid = atoi(optarg); break;
      case 'b': prm.biased = 1; break;
      case 'm': if (mtop < 64) mask[mtop++] = optarg; break;       /* daligner.c:788-795 */
      default:
        fprintf(stderr, "oracle_daligner: unsupported option -%c\n", optopt ? optopt : c);
        return 1;
    }
  if (optind + 2 > argc)
    { fprintf(stderr, "usage: oracle_daligner [options] <subject> <target> ...\n");
      return 1;
    }
  prm.minover *= 2;
  prm.nthreads = nthreads;

  afile = argv[optind++];
  if (damar_read_block(afile, &ablock) || damar_load_masks(&ablock, mask, mtop))
    return 1;
  aroot = damar_root(afile, ".db");
  if (prm.symmetric)                                     /* daligner.c:911-946 */
    for (i = optind; i < argc; i++)
      if (strcmp(afile, argv[i]) != 0)
        { char *broot = damar_root(argv[i], ".db");
          char *ad = strrchr(aroot, '.'), *bd = strrchr(broot, '.');
          size_t la = ad ? (size_t) (ad - aroot + 1) : strlen(aroot);
          size_t lb = bd ? (size_t) (bd - broot + 1) : strlen(broot);
          if (strncmp(aroot, broot, la > lb ? la : lb) != 0)
            prm.symmetric = 0;
          free(broot);
        }
  { char *d = damar_get_dir(runid, ablock.part);
    make_dir(d);
    free(d);
  }
  spec = New_Align_Spec(ecorr, spacing, ablock.freq, 1, prm.symmetric, only_id, notrace, 1);
  aidx = oracle_sort_kmers(&ablock, &prm, &alen);

  for (i = optind; i < argc; i++)
    { char *bfile = argv[i];
      if (strcmp(afile, bfile) != 0)
        { char *broot = damar_root(bfile, ".db");
          char *d1 = NULL, *d2 = NULL;
          int   last;
          if (damar_read_block(bfile, &bblock) || damar_load_masks(&bblock, mask, mtop))
            return 1;
          if (prm.symmetric)
            { char *d = damar_get_dir(runid, bblock.part);
              make_dir(d);
              free(d);
            }
          bidx = oracle_sort_kmers(&bblock, &prm, &blen);
          oracle_match_filter(&ablock, &bblock, aidx, alen, bidx, blen, 0, 0, &prm, spec, cnt, &st);
          if (verbose) printf("N %s x %s: %lld hits %lld seeds %lld confirmed\n", aroot, broot, (long long) cnt[0], (long long) cnt[1], (long long) cnt[2]);
          free(bidx);
          damar_complement_block(&bblock, 1);
          bidx = oracle_sort_kmers(&bblock, &prm, &blen);
          oracle_match_filter(&ablock, &bblock, aidx, alen, bidx, blen, 0, 1, &prm, spec, cnt, &st);
          if (verbose) printf("C %s x %s: %lld hits %lld seeds %lld confirmed\n", aroot, broot, (long long) cnt[0], (long long) cnt[1], (long long) cnt[2]);
          free(bidx);
          last = (bblock.part < ablock.part) ? bblock.ufirst + bblock.nreads - 1
                                             : ablock.ufirst + ablock.nreads - 1;
          if (ablock.part > 0) d1 = damar_get_dir(runid, ablock.part);
          if (bblock.part > 0) d2 = damar_get_dir(runid, bblock.part);
          Write_Overlap_Buffer(spec, d1, d2, aroot, broot, last);
          Reset_Overlap_Buffer(spec);
          free(d1); free(d2); free(broot);
          damar_close_block(&bblock);
        }
      else
        { char *d1 = NULL;
          oracle_match_filter(&ablock, &ablock, aidx, alen, aidx, alen, 1, 0, &prm, spec, cnt, &st);
          if (verbose) printf("N %s x %s: %lld hits %lld seeds %lld confirmed\n", aroot, aroot, (long long) cnt[0], (long long) cnt[1], (long long) cnt[2]);
          cblock = damar_complement_block(&ablock, 0);
          bidx = oracle_sort_kmers(cblock, &prm, &blen);
          oracle_match_filter(&ablock, cblock, aidx, alen, bidx, blen, 1, 1, &prm, spec, cnt, &st);
          if (verbose) printf("C %s x %s: %lld hits %lld seeds %lld confirmed\n", aroot, aroot, (long long) cnt[0], (long long) cnt[1], (long long) cnt[2]);
          free(bidx);
          free(((char *) cblock->bases) - 1);
          if (ablock.part > 0) d1 = damar_get_dir(runid, ablock.part);
          Write_Overlap_Buffer(spec, d1, NULL, aroot, aroot, ablock.ufirst + ablock.nreads - 1);
          Reset_Overlap_Buffer(spec);
          free(d1);
        }
    }
  if (verbose)
    printf("redundancy calls %lld fusions %lld bridges %lld\n", (long long) damar_stat_redundancy_calls,
           (long long) damar_stat_fusions, (long long) damar_stat_bridges);
  if (verbose)
    printf("waves %lld cells %lld maxband %d pebbles %lld emptyband %d\n", (long long) st.waves,
           (long long) st.cells, st.maxband, (long long) st.pebbles, st.empty_band);
  if (verbose && getenv("DAMAR_ORACLE_BANDHIST"))
    { int i;
      printf("directions %lld over31 %lld steps_after %lld\n", (long long) st.dirs, (long long) st.dirs_over31, (long long) st.steps_after_over31);
      { static const char *cls[5] = { "<=13", "<=14", "<=16", "<=29", ">29" };
        int q;
        printf("passes by their widest step (computed diagonals): ");
        for (q = 0; q < 5; q++)
          printf(" %s: %lld passes %lld cells;", cls[q], (long long) st.pass_n[q], (long long) st.pass_cellsum[q]);
        printf(" (the last pass of the run is not counted)\n");
        printf("quarter mode with parking (leave at > 14 computed diagonals, return at <= 12): steps narrow %lld wide %lld promotions %lld\n",
               (long long) st.steps_narrow, (long long) st.steps_wide, (long long) st.promotions);
      }
      printf("bandhist");
      for (i = 0; i < 130; i++)
        printf(" %lld", (long long) st.bandhist[i]);
      printf("\n");
    }
  return 0;
}

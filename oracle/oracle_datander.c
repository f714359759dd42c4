/* oracle_datander.c -- ORACLE (test infrastructure): command-line driver of the CPU
 * restatement of datander (reference scrub/datander.c:121-263): same options, same
 * tan/<block>.<block>.las output. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>
#include "oracle.h"

int main(int argc, char *argv[])
{ OParams prm;
  double  ecorr = .70;
  int     spacing = 100, nthreads = 4, verbose = 0, c, i;
  char   *outdir = "tan";

  memset(&prm, 0, sizeof(prm));
  prm.kmer = 12; prm.binshift = 4; prm.hitmin = 35; prm.minover = 500; prm.symmetric = 1;
  opterr = 0;
  while ((c = getopt(argc, argv, "vk:w:h:e:l:s:o:j:")) != -1)
    switch (c)
    { case 'v': verbose = 1; break;
      case 'k': prm.kmer = atoi(optarg); break;
      case 'w': prm.binshift = atoi(optarg); break;
      case 'h': prm.hitmin = atoi(optarg); break;
      case 'e': ecorr = atof(optarg); break;
      case 'l': prm.minover = atoi(optarg); break;
      case 's': spacing = atoi(optarg); break;
      case 'j': nthreads = atoi(optarg); break;
      case 'o': outdir = optarg; break;
      default: fprintf(stderr, "oracle_datander: unsupported option\n"); return 1;
    }
  if (optind + 1 > argc)
    { fprintf(stderr, "usage: oracle_datander [options] <block> ...\n"); return 1; }
  prm.minover *= 2;
  prm.nthreads = nthreads;
  mkdir(outdir, 0755);
  for (i = optind; i < argc; i++)
    { HITS_DB blk;
      char   *root;
      Align_Spec *spec;
      int64   cnt[3];
      if (damar_read_block(argv[i], &blk))
        return 1;
      root = damar_root(argv[i], ".db");
      spec = New_Align_Spec(ecorr, spacing, blk.freq, 1, 1, 0, 0, 0);
      oracle_match_self(&blk, &prm, spec, cnt, NULL);
      if (verbose)
        printf("%s: %lld k-mers, %lld seed hits, %lld confirmed\n", root, (long long) cnt[0], (long long) cnt[1], (long long) cnt[2]);
      Write_Overlap_Buffer(spec, outdir, outdir, root, root, blk.ufirst + blk.nreads - 1);
      Reset_Overlap_Buffer(spec);
      Free_Align_Spec(spec);
      free(root);
      damar_close_block(&blk);
    }
  return 0;
}

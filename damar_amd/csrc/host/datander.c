/* datander.c -- host driver of the MI355X self-tandem finder: same command line and
 * tan/<block>.<block>.las output as the reference's scrub/datander.c:121-263, calling
 * Match_Self of libdamar_hip.so.  Host code stays C. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <sys/stat.h>

#include "damar_filter.h"
#include "damar_hip.h"

int main(int argc, char *argv[])
{ int    kmer = 12, hitmin = 35, binshift = 4, spacing = 100, nthreads = 4, c, i, gpu = -1;
  double ecorr = .70;
  char  *outdir = "tan";
  struct stat st;

  MINOVER = 500;
  opterr = 0;
  while ((c = getopt(argc, argv, "vk:w:h:e:l:s:o:j:g:")) != -1)
    switch (c)
    { case 'v': VERBOSE = 1; break;
      case 'k': kmer = atoi(optarg); break;
      case 'w': binshift = atoi(optarg); break;
      case 'h': hitmin = atoi(optarg); break;
      case 'e': ecorr = atof(optarg); break;
      case 'l': MINOVER = atoi(optarg); break;
      case 's': spacing = atoi(optarg); break;
      case 'j': nthreads = atoi(optarg); break;
      case 'o': outdir = optarg; break;
      case 'g': gpu = atoi(optarg); break;
      default:
        fprintf(stderr, "Unsupported option: %s\n", argv[optind - 1]);
        exit(1);
    }
  if (kmer < 0 || hitmin < 0 || MINOVER < 0 || spacing < 0)
    { fprintf(stderr, "datander: negative option value\n");
      exit(1);
    }
  if (ecorr < .5 || ecorr >= 1.)
    { fprintf(stderr, "Average correlation must be in [.5,1.) (%g)\n", ecorr);
      exit(1);
    }
  if (optind + 1 > argc)
    { fprintf(stderr, "[ERROR] - at least one subject block is required\n\n");
      exit(1);
    }
  MINOVER *= 2;
  if (damar_tandem_set_params(kmer, binshift, hitmin, nthreads))
    { fprintf(stderr, "Illegal combination of filter parameters\n");
      exit(1);
    }
  if (gpu >= 0)
    damar_hip_init(gpu);
  if (stat(outdir, &st) != 0)
    { if (errno == ENOENT)
        mkdir(outdir, S_IRWXU | S_IRGRP | S_IXGRP | S_IROTH | S_IXOTH);
      else
        { fprintf(stderr, "Cannot create output directory: %s\n", outdir);
          exit(1);
        }
    }
  else if (!S_ISDIR(st.st_mode))
    { fprintf(stderr, "Output directory name: \"%s\" exist - but its not a directory\n", outdir);
      exit(1);
    }
  for (i = optind; i < argc; i++)
    { HITS_DB blk;
      char   *root;
      Align_Spec *spec;
      int     r;
      if (damar_read_block(argv[i], &blk))
        exit(1);
      for (r = 0; r < blk.nreads; r++)
        if (blk.reads[r].rlen < kmer)
          { fprintf(stderr, "[ERROR] - datander: Block %s contains reads < %dbp long !  Run DBsplit.\n", argv[i], kmer);
            exit(1);
          }
      root = damar_root(argv[i], ".db");
      spec = New_Align_Spec(ecorr, spacing, blk.freq, nthreads, 1, 0, 0, 0);
      Match_Self(root, &blk, spec);
      Write_Overlap_Buffer(spec, outdir, outdir, root, root, blk.ufirst + blk.nreads - 1);
      Reset_Overlap_Buffer(spec);
      Free_Align_Spec(spec);
      free(root);
      damar_close_block(&blk);
    }
  return 0;
}

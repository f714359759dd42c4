/* simdb -- write a synthetic PacBio-style read database (simulator | FA2db | DBsplit
 * equivalent, see damar_db.h) so that the GPU box can make BASELINE.json's inputs
 * from seeds alone.  usage: simdb <dir> <root> <genome_Mbp> [-cCOV] [-rSEED] [-eERR]
 *                                 [-mMEAN] [-sSDEV] [-xSHORT] [-bBIAS] [-SBLOCK_MBP]
 *                                 [-TFRAC: implant a tandem array into this fraction of the reads]
 *                                 [-NBLOCKS: only the first BLOCKS blocks of that database] */
#include <stdio.h>
#include <stdlib.h>
#include "damar_db.h"

int main(int argc, char *argv[])
{ damar_sim_params p;
  int i, nb;

  if (argc < 4)
    { fprintf(stderr, "usage: simdb <dir> <root> <genome_Mbp> [-c -r -e -m -s -x -b -S -T -N]\n");
      return 1;
    }
  damar_sim_defaults(&p);
  p.genome_mbp = atof(argv[3]);
  for (i = 4; i < argc; i++)
    if (argv[i][0] == '-')
      switch (argv[i][1])
      { case 'c': p.coverage  = atof(argv[i] + 2); break;
        case 'r': p.seed      = atoi(argv[i] + 2); break;
        case 'e': p.erate     = atof(argv[i] + 2); break;
        case 'm': p.rmean     = atoi(argv[i] + 2); break;
        case 's': p.rsdev     = atoi(argv[i] + 2); break;
        case 'x': p.rshort    = atoi(argv[i] + 2); break;
        case 'b': p.bias      = atof(argv[i] + 2); break;
        case 'S': p.block_mbp = atoi(argv[i] + 2); break;
        case 'T': p.tandem_frac = atof(argv[i] + 2); break;
        case 'N': p.max_blocks = atoi(argv[i] + 2); break;
        default:
          fprintf(stderr, "simdb: unknown option %s\n", argv[i]);
          return 1;
      }
  nb = damar_sim_write_db(&p, argv[1], argv[2]);
  if (nb < 0)
    return 1;
  printf("%d\n", nb);
  return 0;
}

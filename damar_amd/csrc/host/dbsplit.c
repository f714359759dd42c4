/* dbsplit.c -- block partition of a database, the reference's db/DBsplit.c:82-245: appends (or replaces)
 * the "blocks = / size = / first read of every block" section of <path>.db; a block is closed as soon as
 * its reads total -s * 10^6 bases (DBsplit.c:201-226).  The .idx header is written back unchanged, as the
 * reference does.
 *
 *     DBsplit [-s<int(200)>] <path:db>
 *
 * Host code, C, no GPU.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "damar_db.h"

int main(int argc, char *argv[])
{ int    size = 200, c, nfiles, nblocks, i, nblock = 0, ireads = 0;
  char  *root, *dir, path[8192], line[32768];
  FILE  *stub, *idx;
  HITS_DB    db;
  HITS_READ  rec;
  long   pos;
  int64  tot = 0, lim;

  opterr = 0;
  while ((c = getopt(argc, argv, "s:")) != -1)
    switch (c)
    { case 's':
        size = atoi(optarg);
        if (size <= 0)
          { fprintf(stderr, "invalid block size of %d\n", size);
            exit(1);
          }
        break;
      default:
        fprintf(stderr, "usage: [-s<int(200)>] <path:db|dam>\n");
        exit(1);
    }
  if (argc - optind != 1)
    { fprintf(stderr, "usage: [-s<int(200)>] <path:db|dam>\n");
      exit(1);
    }
  root = damar_root(argv[optind], ".db");
  { const char *s = strrchr(argv[optind], '/');
    dir = s ? strndup(argv[optind], (size_t) (s - argv[optind])) : strdup(".");
  }
  { char *dot = strrchr(root, '.');
    if (dot != NULL && dot[1] >= '1' && dot[1] <= '9' && strspn(dot + 1, "0123456789") == strlen(dot + 1))
      { fprintf(stderr, "[ERROR] Cannot be called on a block: %s\n", argv[optind]);
        exit(1);
      }
  }
  snprintf(path, sizeof(path), "%s/%s.db", dir, root);
  stub = fopen(path, "r+");
  snprintf(path, sizeof(path), "%s/.%s.idx", dir, root);
  idx = fopen(path, "r+");
  if (stub == NULL || idx == NULL)
    { fprintf(stderr, "[ERROR] Cannot open database %s\n", argv[optind]);
      exit(1);
    }
  if (fscanf(stub, "files = %9d\n", &nfiles) != 1)
    { fprintf(stderr, "DBsplit: stub file of %s is junk\n", root);
      exit(1);
    }
  for (i = 0; i < nfiles; i++)
    if (fgets(line, sizeof(line), stub) == NULL)
      { fprintf(stderr, "DBsplit: stub file of %s is junk\n", root);
        exit(1);
      }
  if (fread(&db, sizeof(HITS_DB), 1, idx) != 1)
    { fprintf(stderr, "DBsplit: index of %s is junk\n", root);
      exit(1);
    }
  pos = ftell(stub);
  if (fscanf(stub, "blocks = %9d\n", &nblocks) == 1)                  /* DBsplit.c:168-185 */
    { printf("You are about to overwrite the current partition settings.  This\n");
      printf("will invalidate any tracks, overlaps, and other derivative files.\n");
      printf("Are you sure you want to proceed? [Y/N] ");
      fflush(stdout);
      if (fgets(line, 100, stdin) == NULL || strchr(line, 'n') != NULL || strchr(line, 'N') != NULL)
        { printf("Aborted\n");
          fflush(stdout);
          fclose(stub);
          exit(1);
        }
    }
  fseek(stub, pos, SEEK_SET);
  fprintf(stub, "blocks = %9d\n", 0);
  fprintf(stub, "size = %9lld\n", (long long) size);
  lim = size * 1000000ll;
  fprintf(stub, " %9d\n", 0);
  for (i = 0; i < db.ureads; i++)
    { if (fread(&rec, sizeof(HITS_READ), 1, idx) != 1)
        { fprintf(stderr, "DBsplit: index of %s is truncated\n", root);
          exit(1);
        }
      ireads += 1;
      tot += rec.rlen;
      if (tot >= lim)
        { fprintf(stub, " %9d\n", i + 1);
          tot = 0;
          ireads = 0;
          nblock += 1;
        }
    }
  if (ireads > 0)
    { fprintf(stub, " %9d\n", db.ureads);
      nblock += 1;
    }
  fflush(stub);
  if (ftruncate(fileno(stub), ftell(stub)) < 0)
    { fprintf(stderr, "DBsplit: cannot truncate the stub of %s\n", root);
      exit(1);
    }
  fseek(stub, pos, SEEK_SET);
  fprintf(stub, "blocks = %9d\n", nblock);
  rewind(idx);
  fwrite(&db, sizeof(HITS_DB), 1, idx);
  fclose(idx);
  fclose(stub);
  return 0;
}

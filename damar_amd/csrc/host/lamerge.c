/* lamerge.c -- LAmerge: merge the sorted per-pair .las files of one block into one sorted .las.
 *
 * The step that follows daligner in every plan (dalign/HPCdaligner.c:790-808 emits
 * `LAmerge [-v] [-k] [-s] -n <N> <db> <name>.<i>.las <dir>`); restates the observable behaviour of
 * utils/LAmerge.c + utils/LAmergeUtils.c for that use: the inputs are the .las files of <dir>
 * (or the files named on the command line), each sorted as Write_Overlap_Buffer leaves them
 * (align.c:6104-6164), and the output is their merge in the order of LAmergeUtils.c:63-99
 * (aread, bread, COMP, abpos ascending; aepos descending; bbpos ascending; bepos descending).
 * All inputs are merged in ONE pass through a tournament over the file heads: the reference
 * merges at most -n files at a time in rounds, which yields the same sequence whenever no two
 * files hold records that compare equal -- always true for daligner output, where every file of
 * a block directory has its own B block.  -s sorts every input first (LAmergeUtils.c:31-61).
 * Host-only: this is file plumbing around the hot path, not part of it.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <dirent.h>

#include "damar_align.h"

typedef struct
{ FILE   *f;
  char   *name;
  int64   left;          /* records not yet read */
  Overlap ovl;           /* current head (path.trace unused) */
  char   *trace;         /* its trace bytes */
  int     tcap;
  int     tbytes;
} Src;

static int sort_flag = 0, verbose = 0;

static void die(const char *msg, const char *arg)
{ fprintf(stderr, "[ERROR] - LAmerge: %s %s\n", msg, arg ? arg : "");
  exit(1);
}

/* LAmergeUtils.c:63-99 COMPARE: is l strictly after r? */
static int after(const Overlap *l, const Overlap *r)
{ if (l->aread != r->aread) return l->aread > r->aread;
  if (l->bread != r->bread) return l->bread > r->bread;
  if (COMP(l->flags) != COMP(r->flags)) return COMP(l->flags) > COMP(r->flags);
  if (l->path.abpos != r->path.abpos) return l->path.abpos > r->path.abpos;
  if (l->path.aepos != r->path.aepos) return l->path.aepos < r->path.aepos;
  if (l->path.bbpos != r->path.bbpos) return l->path.bbpos > r->path.bbpos;
  if (l->path.bepos != r->path.bepos) return l->path.bepos < r->path.bepos;
  return 0;
}

/* one record: 40 bytes from &ovl.path.tlen on (align.c:3335-3363 Read_Overlap), then the trace */
#define OVL_IO ((int) (sizeof(Overlap) - sizeof(void *)))

static int next_record(Src *s)
{ int n;
  if (s->left <= 0)
    return 0;
  if (fread(((char *) &s->ovl) + sizeof(void *), OVL_IO, 1, s->f) != 1)
    die("truncated record in", s->name);
  n = s->ovl.path.tlen * s->tbytes;
  if (n > s->tcap)
    { s->tcap = n + 256;
      s->trace = (char *) realloc(s->trace, (size_t) s->tcap);
    }
  if (n > 0 && fread(s->trace, (size_t) n, 1, s->f) != 1)
    die("truncated trace in", s->name);
  s->left -= 1;
  return 1;
}

/* LAmergeUtils.c:31-61 SORT_OVL for -s: every key ascending */
typedef struct { Overlap ovl; char *trace; } Rec;

static int by_sort_ovl(const void *x, const void *y)
{ const Overlap *l = &((const Rec *) x)->ovl, *r = &((const Rec *) y)->ovl;
  if (l->aread != r->aread) return l->aread - r->aread;
  if (l->bread != r->bread) return l->bread - r->bread;
  if (COMP(l->flags) != COMP(r->flags)) return COMP(l->flags) > COMP(r->flags) ? 1 : -1;
  if (l->path.abpos != r->path.abpos) return l->path.abpos - r->path.abpos;
  if (l->path.aepos != r->path.aepos) return l->path.aepos - r->path.aepos;
  if (l->path.bbpos != r->path.bbpos) return l->path.bbpos - r->path.bbpos;
  return l->path.bepos - r->path.bepos;
}

static void sort_file_in_place(const char *name)
{ FILE *f = fopen(name, "r");
  int64 novl, i;
  int   tspace, tbytes;
  Rec  *all;
  if (f == NULL) die("Cannot open file for reading:", name);
  if (fread(&novl, sizeof(int64), 1, f) != 1 || fread(&tspace, sizeof(int), 1, f) != 1)
    die("failed to read header of", name);
  tbytes = (tspace <= TRACE_XOVR) ? 1 : 2;
  all = (Rec *) malloc(sizeof(Rec) * (size_t) (novl > 0 ? novl : 1));
  for (i = 0; i < novl; i++)
    { int n;
      if (fread(((char *) &all[i].ovl) + sizeof(void *), OVL_IO, 1, f) != 1) die("truncated record in", name);
      n = all[i].ovl.path.tlen * tbytes;
      all[i].trace = (char *) malloc((size_t) (n > 0 ? n : 1));
      if (n > 0 && fread(all[i].trace, (size_t) n, 1, f) != 1) die("truncated trace in", name);
    }
  fclose(f);
  qsort(all, (size_t) novl, sizeof(Rec), by_sort_ovl);
  f = fopen(name, "w");
  if (f == NULL) die("Cannot open file for writing:", name);
  fwrite(&novl, sizeof(int64), 1, f);
  fwrite(&tspace, sizeof(int), 1, f);
  for (i = 0; i < novl; i++)
    { fwrite(((char *) &all[i].ovl) + sizeof(void *), OVL_IO, 1, f);
      fwrite(all[i].trace, (size_t) (all[i].ovl.path.tlen * tbytes), 1, f);
      free(all[i].trace);
    }
  fclose(f);
  free(all);
}

static int by_name(const void *x, const void *y) { return strcmp(*(char *const *) x, *(char *const *) y); }

static void usage(const char *prog)
{ fprintf(stderr, "Usage:\t%s\t[-hksv] [-n numFiles(8)] <db> <out.las> [<directory>| <in.1.las in.2.las ...>]\n", prog);
}

#ifdef LAMERGE_AS_LIB       /* linked into the daligner driver: its node scheduler merges in a forked child, without an exec */
int lamerge_main(int argc, char *argv[])
#else
int main(int argc, char *argv[])
#endif
{ char **files = NULL;
  int    nfiles = 0, cap = 0, c, i, tspace = -1;
  const char *out;
  Src   *src;
  int   *heap, hsize;
  int64  total = 0, written = 0;
  FILE  *of;

  while ((c = getopt(argc, argv, "hksvn:C:S:f:")) != -1)
    switch (c)
    { case 'v': verbose += 1; break;
      case 's': sort_flag = 1; break;
      case 'k': case 'n': case 'C': break;            /* rounds / intermediates / checks: single pass here */
      case 'h': usage(argv[0]); return 0;
      default:
        fprintf(stderr, "LAmerge: option -%c is not supported by this build\n", optopt ? optopt : c);
        return 1;
    }
  if (argc - optind < 3)
    { fprintf(stderr, "At least a database, an output file and a directory or input files are required!\n\n");
      usage(argv[0]);
      return 1;
    }
  out = argv[optind + 1];
  { DIR *d = opendir(argv[optind + 2]);
    if (d != NULL && argc - optind == 3)
      { struct dirent *e;
        while ((e = readdir(d)) != NULL)
          { size_t n = strlen(e->d_name);
            char  *p;
            if (n < 5 || strcmp(e->d_name + n - 4, ".las") != 0)
              continue;
            p = (char *) malloc(strlen(argv[optind + 2]) + n + 2);
            sprintf(p, "%s/%s", argv[optind + 2], e->d_name);
            if (nfiles == cap) { cap = 2 * cap + 16; files = (char **) realloc(files, sizeof(char *) * (size_t) cap); }
            files[nfiles++] = p;
          }
        closedir(d);
        qsort(files, (size_t) nfiles, sizeof(char *), by_name);
      }
    else
      { if (d) closedir(d);
        for (i = optind + 2; i < argc; i++)
          { if (nfiles == cap) { cap = 2 * cap + 16; files = (char **) realloc(files, sizeof(char *) * (size_t) cap); }
            files[nfiles++] = argv[i];
          }
      }
  }
  if (nfiles == 0)
    die("no overlap files to merge in", argv[optind + 2]);

  src  = (Src *) calloc((size_t) nfiles, sizeof(Src));
  heap = (int *) malloc(sizeof(int) * (size_t) (nfiles + 1));
  hsize = 0;
  for (i = 0; i < nfiles; i++)
    { Src *s = src + i;
      int  ts;
      if (sort_flag)
        sort_file_in_place(files[i]);
      s->name = files[i];
      if ((s->f = fopen(files[i], "r")) == NULL)
        die("Cannot open file for reading:", files[i]);
      if (fread(&s->left, sizeof(int64), 1, s->f) != 1 || fread(&ts, sizeof(int), 1, s->f) != 1)
        die("failed to read header of", files[i]);
      if (tspace < 0)
        tspace = ts;
      else if (ts != tspace)
        die("trace spacing differs in", files[i]);
      s->tbytes = (ts <= TRACE_XOVR) ? 1 : 2;
      total += s->left;
      if (verbose)
        printf("%s, novl: %lld\n", files[i], (long long) s->left);
      if (next_record(s))
        heap[++hsize] = i;
    }

  if ((of = fopen(out, "w")) == NULL)
    die("Cannot open output file", out);
  fwrite(&total, sizeof(int64), 1, of);
  fwrite(&tspace, sizeof(int), 1, of);

  /* selection over the heads: the first (lowest file index) of the smallest records wins */
  while (hsize > 0)
    { int best = 1, h;
      Src *s;
      for (h = 2; h <= hsize; h++)
        { const Src *a = src + heap[h], *b = src + heap[best];
          if (after(&b->ovl, &a->ovl) || (!after(&a->ovl, &b->ovl) && heap[h] < heap[best]))
            best = h;
        }
      s = src + heap[best];
      fwrite(((char *) &s->ovl) + sizeof(void *), OVL_IO, 1, of);
      fwrite(s->trace, (size_t) (s->ovl.path.tlen * s->tbytes), 1, of);
      written += 1;
      if (!next_record(s))
        { fclose(s->f);
          heap[best] = heap[hsize--];
        }
    }
  if (written != total)
    die("record count mismatch writing", out);
  fclose(of);
  if (verbose)
    printf("%s, novl: %lld\n", out, (long long) total);
  return 0;
}

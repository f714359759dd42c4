/* las.c -- Align_Spec, the per-thread overlap buffers and the .las writer.
 *
 * Host side of the overlap path.  The
 * observable behaviour follows the reference routines cited at each function;
 * the on-disk layout is SURVEY.md App. C (.las = int64 novl, int32 tspace, then
 * 40-byte records + tlen trace values of 1 or 2 bytes).
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>

#include "damar_align.h"
#include "damar_host.h"

#define TRIM_BITS   15
#define TRIM_SIZE   (1 << TRIM_BITS)
#define PATH_LEN    60
#define FRACTION    1000

typedef struct
{ double ave_corr;
  int    trace_space;
  int    reach;
  float  freq[4];
  int    ave_path;
  int16 *score;           /* SCORE[32768] followed by TABLE[32768] */
  int16 *table;
  int    nthreads;
  Overlap_IO_Buffer *iobuf;
  int    symmetric;
  int    only_identity;
} Spec;

/* align.c:198-199 */
static const double bias_factor[10] = { .690, .690, .690, .690, .780, .850, .900, .933, .966, 1.000 };

/* align.c:234-318: TABLE[x] = (score of the 15 columns in x) - (largest score of a
 * proper prefix of them), so TABLE[x] >= 0 iff every suffix scores non-negative;
 * SCORE[x] = total.  Bit 14 of x is the oldest column. */
Align_Spec *New_Align_Spec(double ave_corr, int trace_space, float *freq, int nthreads,
                           int symmetric, int only_identity, int no_trace_points, int reach)
{ Spec  *s = (Spec *) malloc(sizeof(Spec));
  double match;
  int    bias, mscore, dscore, x, i;

  if (s == NULL)
    { fprintf(stderr, "damar: out of memory (alignment specification)\n");
      exit(1);
    }
  s->ave_corr    = ave_corr;
  s->trace_space = trace_space;
  s->reach       = reach;
  for (i = 0; i < 4; i++)
    s->freq[i] = freq[i];

  match = freq[0] + freq[3];
  if (match > .5)
    match = 1. - match;
  bias = (int) ((match + .025) * 20. - 1.);
  if (match < .2)
    { fprintf(stderr, "Warning: Base bias worse than 80/20%% ! (New_Align_Spec)\n");
      fprintf(stderr, "         Capping bias at this ratio.\n");
      bias = 3;
    }
  s->ave_path = (int) (PATH_LEN * (1. - bias_factor[bias] * (1. - ave_corr)));
  mscore = (int) (FRACTION * bias_factor[bias] * (1. - ave_corr));
  dscore = FRACTION - mscore;

  s->score = (int16 *) malloc(sizeof(int16) * 2 * TRIM_SIZE);
  if (s->score == NULL)
    { fprintf(stderr, "damar: out of memory (trim tables)\n");
      exit(1);
    }
  s->table = s->score + TRIM_SIZE;
  for (x = 0; x < TRIM_SIZE; x++)
    { int sc = 0, mx = 0;
      for (i = TRIM_BITS - 1; i >= 0; i--)
        { if (sc > mx)
            mx = sc;
          if ((x >> i) & 1)
            sc += mscore;
          else
            sc -= dscore;
        }
      s->table[x] = (int16) (sc - mx);
      s->score[x] = (int16) sc;
    }

  s->nthreads      = nthreads;
  s->symmetric     = symmetric;
  s->only_identity = only_identity;
  s->iobuf = (Overlap_IO_Buffer *) malloc(sizeof(Overlap_IO_Buffer) * (size_t) (nthreads > 0 ? nthreads : 1));
  for (i = 0; i < nthreads; i++)
    { Overlap_IO_Buffer *ob = CreateOverlapBuffer(nthreads, (trace_space <= TRACE_XOVR) ? 1 : 2, no_trace_points);
      if (ob == NULL)
        exit(1);
      s->iobuf[i] = *ob;
      free(ob);
    }
  return (Align_Spec *) s;
}

void Free_Align_Spec(Align_Spec *spec)
{ Spec *s = (Spec *) spec;
  int   i;
  for (i = 0; i < s->nthreads; i++)
    { free(s->iobuf[i].ovls);
      if (!s->iobuf[i].no_trace)
        free(s->iobuf[i].trace);
    }
  free(s->iobuf);
  free(s->score);
  free(s);
}

int    Trace_Spacing(Align_Spec *spec)        { return ((Spec *) spec)->trace_space; }
double Average_Correlation(Align_Spec *spec)  { return ((Spec *) spec)->ave_corr; }
float *Base_Frequencies(Align_Spec *spec)     { return ((Spec *) spec)->freq; }
int    Overlap_If_Possible(Align_Spec *spec)  { return ((Spec *) spec)->reach; }
int    Num_Threads(Align_Spec *spec)          { return ((Spec *) spec)->nthreads; }
int    Only_Identity(Align_Spec *spec)        { return ((Spec *) spec)->only_identity; }
int    Symmetric(Align_Spec *spec)            { return ((Spec *) spec)->symmetric; }
Overlap_IO_Buffer *OVL_IO_Buffer(Align_Spec *spec) { return ((Spec *) spec)->iobuf; }

const int16 *damar_spec_score_table(Align_Spec *spec) { return ((Spec *) spec)->score; }
const int16 *damar_spec_trim_table(Align_Spec *spec)  { return ((Spec *) spec)->table; }
int          damar_spec_ave_path(Align_Spec *spec)    { return ((Spec *) spec)->ave_path; }

/* align.c:5969-6018 */
Overlap_IO_Buffer *CreateOverlapBuffer(int nthreads, int tbytes, int no_trace)
{ Overlap_IO_Buffer *b = (Overlap_IO_Buffer *) calloc(1, sizeof(Overlap_IO_Buffer));
  if (b == NULL)
    return NULL;
  b->omax = 500000 / nthreads + 1;
  b->ovls = (Overlap *) calloc((size_t) b->omax, sizeof(Overlap));
  b->no_trace = no_trace;
  if (b->ovls == NULL)
    return NULL;
  if (no_trace)
    return b;
  if (tbytes < 1 || tbytes > 2)
    { fprintf(stderr, "[ERROR] - Unsupported size of trace: %d!\n", tbytes);
      return NULL;
    }
  b->tbytes = tbytes;
  b->tmax   = (uint64) b->omax * 150;
  b->trace  = malloc((size_t) b->tmax * (size_t) tbytes);
  if (b->trace == NULL)
    return NULL;
  return b;
}

/* align.c:6020-6102.  Traces are kept as byte offsets while the pool may still
 * move; they are turned into pointers when the buffer is drained. */
int AddOverlapToBuffer(Overlap_IO_Buffer *b, Overlap *ovl, int tbytes)
{ Overlap *o;
  int      keep;

  if (b == NULL)
    { fprintf(stderr, "[ERROR] - Cannot add overlap to Overlap_IO_Buffer. Buffer is NULL!\n");
      return 1;
    }
  if (b->otop == b->omax)
    { b->omax = (int) (b->omax * 1.2) + 1000;
      b->ovls = (Overlap *) realloc(b->ovls, sizeof(Overlap) * (size_t) b->omax);
      if (b->ovls == NULL)
        { fprintf(stderr, "[ERROR] - Cannot increase overlap buffer size to %d!\n", b->omax);
          return 1;
        }
    }
  keep = (ovl->path.trace != NULL && !b->no_trace);
  if (keep)
    { uint64 need = b->ttop + (uint64) tbytes * (uint64) ovl->path.tlen;
      if (need >= b->tmax * (uint64) b->tbytes)
        { while (need >= b->tmax * (uint64) b->tbytes)
            b->tmax = (uint64) (b->tmax * 1.2) + 1000;
          b->trace = realloc(b->trace, (size_t) b->tmax * (size_t) b->tbytes);
          if (b->trace == NULL)
            { fprintf(stderr, "[ERROR] - Cannot increase trace point buffer size to %llu!\n",
                      (unsigned long long) b->tmax);
              return 1;
            }
        }
    }
  o = b->ovls + b->otop;
  memset(o, 0, sizeof(Overlap));
  o->aread      = ovl->aread;
  o->bread      = ovl->bread;
  o->flags      = ovl->flags;
  o->path.abpos = ovl->path.abpos;
  o->path.aepos = ovl->path.aepos;
  o->path.bbpos = ovl->path.bbpos;
  o->path.bepos = ovl->path.bepos;
  o->path.diffs = ovl->path.diffs;
  if (keep)
    { o->path.tlen  = ovl->path.tlen;
      memcpy(((char *) b->trace) + b->ttop, ovl->path.trace, (size_t) tbytes * (size_t) ovl->path.tlen);
      o->path.trace = (void *) (uintptr_t) (b->ttop + 1);      /* offset+1, resolved on write */
      b->ttop += (uint64) tbytes * (uint64) ovl->path.tlen;
    }
  else
    { o->path.trace = NULL;
      o->path.tlen  = 0;
    }
  b->otop += 1;
  return 0;
}

/* Append every overlap of src (with its trace) to dst, in order: what AddOverlapToBuffer would
 * have produced had the records been added to dst directly. */
int damar_append_overlap_buffer(Overlap_IO_Buffer *dst, const Overlap_IO_Buffer *src)
{ int    i;
  uint64 base;

  if (src->otop == 0)
    return 0;
  if (dst->otop + src->otop > dst->omax)
    { dst->omax = (int) ((dst->otop + src->otop) * 1.2) + 1000;
      dst->ovls = (Overlap *) realloc(dst->ovls, sizeof(Overlap) * (size_t) dst->omax);
      if (dst->ovls == NULL)
        return 1;
    }
  base = dst->ttop;
  if (!dst->no_trace && src->ttop > 0)
    { uint64 need = dst->ttop + src->ttop;
      if (need >= dst->tmax * (uint64) dst->tbytes)
        { while (need >= dst->tmax * (uint64) dst->tbytes)
            dst->tmax = (uint64) (dst->tmax * 1.2) + 1000;
          dst->trace = realloc(dst->trace, (size_t) dst->tmax * (size_t) dst->tbytes);
          if (dst->trace == NULL)
            return 1;
        }
      memcpy(((char *) dst->trace) + dst->ttop, src->trace, (size_t) src->ttop);
      dst->ttop += src->ttop;
    }
  memcpy(dst->ovls + dst->otop, src->ovls, sizeof(Overlap) * (size_t) src->otop);
  for (i = 0; i < src->otop; i++)
    { Overlap *o = dst->ovls + dst->otop + i;
      if (o->path.trace != NULL)
        o->path.trace = (void *) (uintptr_t) ((uintptr_t) o->path.trace + base);
    }
  dst->otop += src->otop;
  return 0;
}

/* align.c:3375-3396 */
int Compress_TraceTo8(Overlap *ovl, int check)
{ uint16 *t16 = (uint16 *) ovl->path.trace;
  uint8  *t8  = (uint8 *) ovl->path.trace;
  int     j;
  for (j = 0; j < ovl->path.tlen; j++)
    { if (check && t16[j] > 255)
        { fprintf(stderr, "damar: Compression of trace to bytes fails, value too big\n");
          exit(1);
        }
      t8[j] = (uint8) t16[j];
    }
  return 0;
}

/* 40 bytes on disk: the Overlap minus its leading pointer (align.c:3345-3373). */
int Write_Overlap(FILE *out, Overlap *ovl, int tbytes)
{ int32_t rec[10];
  rec[0] = ovl->path.tlen;
  rec[1] = ovl->path.diffs;
  rec[2] = ovl->path.abpos;
  rec[3] = ovl->path.bbpos;
  rec[4] = ovl->path.aepos;
  rec[5] = ovl->path.bepos;
  rec[6] = (int32_t) ovl->flags;
  rec[7] = ovl->aread;
  rec[8] = ovl->bread;
  rec[9] = 0;
  if (fwrite(rec, sizeof(rec), 1, out) != 1)
    return 1;
  if (ovl->path.trace != NULL && ovl->path.tlen > 0)
    if (fwrite(ovl->path.trace, (size_t) tbytes, (size_t) ovl->path.tlen, out) != (size_t) ovl->path.tlen)
      return 1;
  return 0;
}

typedef struct                   /* one gathered record: where it lies, and its place in the gather */
{ const Overlap *ovl;
  const char    *trace;          /* its trace bytes (NULL: none) */
  int            seq;
} Keyed;

/* align.c:6104-6164 SORT_OVL; the reference's last key is the record's address in
 * the gathered array, here its gather index (same order for a stable gather). */
static int by_overlap(const void *x, const void *y)
{ const Keyed *kl = (const Keyed *) x, *kr = (const Keyed *) y;
  const Overlap *l = kl->ovl, *r = kr->ovl;
  int cl, cr;
  if (l->aread != r->aread) return l->aread - r->aread;
  if (l->bread != r->bread) return l->bread - r->bread;
  cl = COMP(l->flags);
  cr = COMP(r->flags);
  if (cl != cr) return cl - cr;
  if (l->path.abpos != r->path.abpos) return l->path.abpos - r->path.abpos;
  if (l->path.aepos != r->path.aepos) return l->path.aepos - r->path.aepos;
  if (l->path.bbpos != r->path.bbpos) return l->path.bbpos - r->path.bbpos;
  if (l->path.bepos != r->path.bepos) return l->path.bepos - r->path.bepos;
  return (kl->seq < kr->seq) ? -1 : (kl->seq > kr->seq);
}

static FILE *open_las(const char *path, int tspace)
{ FILE *out = fopen(path, "w");
  int64 zero = 0;
  if (out == NULL)
    { fprintf(stderr, "[ERROR] - Write_Overlap_Buffer: Cannot open file %s for writing\n", path);
      exit(1);
    }
  setvbuf(out, NULL, _IOFBF, 1 << 20);           /* (a record is two small fwrites: 4 KB of stdio buffer is a write() per 25 records) */
  fwrite(&zero, sizeof(int64), 1, out);
  fwrite(&tspace, sizeof(int), 1, out);
  return out;
}

static void close_las(FILE *out, int64 n)
{ rewind(out);
  fwrite(&n, sizeof(int64), 1, out);
  fclose(out);
}

/* "NAME.7" -> root "NAME", id 7; no dot -> id 0 (align.c:6206-6228) */
static int split_block_name(const char *name, char *root, size_t cap)
{ const char *dot = strrchr(name, '.');
  if (dot == NULL)
    { root[0] = '\0';
      return 0;
    }
  snprintf(root, cap, "%.*s", (int) (dot - name), name);
  return atoi(dot + 1);
}

/* The order of by_overlap (a total order, so any sorting method gives the same sequence):
 * records are dealt into their A-read's bucket first, which leaves only a handful per bucket
 * to order by the remaining keys. */
static Keyed *sort_keyed(Keyed *all, int n)
{ int     lo, hi, i, j, k;
  int    *first;
  Keyed  *out;

  if (n < 64)
    { qsort(all, (size_t) n, sizeof(Keyed), by_overlap);
      return all;
    }
  lo = hi = all[0].ovl->aread;
  for (i = 1; i < n; i++)
    { if (all[i].ovl->aread < lo) lo = all[i].ovl->aread;
      if (all[i].ovl->aread > hi) hi = all[i].ovl->aread;
    }
  first = (int *) calloc((size_t) (hi - lo) + 2, sizeof(int));
  out   = (Keyed *) malloc(sizeof(Keyed) * (size_t) n);
  if (first == NULL || out == NULL)
    { free(first);  free(out);
      qsort(all, (size_t) n, sizeof(Keyed), by_overlap);
      return all;
    }
  for (i = 0; i < n; i++)
    first[all[i].ovl->aread - lo + 1] += 1;
  for (i = 1; i <= hi - lo + 1; i++)
    first[i] += first[i - 1];
  for (i = 0; i < n; i++)
    out[first[all[i].ovl->aread - lo]++] = all[i];
  /* first[b] is now the end of bucket b */
  for (i = 0, k = 0; k <= hi - lo; k++)
    { int e = first[k];
      if (e - i > 24)
        qsort(out + i, (size_t) (e - i), sizeof(Keyed), by_overlap);
      else
        for (j = i + 1; j < e; j++)
          { Keyed x = out[j];
            int   r = j;
            while (r > i && by_overlap(&out[r - 1], &x) > 0)
              { out[r] = out[r - 1];  r -= 1; }
            out[r] = x;
          }
      i = e;
    }
  free(first);
  free(all);
  return out;
}

static void write_keyed(FILE *out, const Keyed *k, int tbytes)
{ Overlap o = *k->ovl;
  o.path.trace = (void *) k->trace;
  Write_Overlap(out, &o, tbytes);
}

typedef struct { const char *path; int tspace, tbytes; const Keyed *recs; int n; } FilePart;

static void *write_file_part(void *arg)
{ const FilePart *f = (const FilePart *) arg;
  FILE *out = open_las(f->path, f->tspace);
  int   j;
  for (j = 0; j < f->n; j++)
    write_keyed(out, f->recs + j, f->tbytes);
  close_las(out, f->n);
  return NULL;
}

/* align.c:6166-6367 on an explicit set of per-thread buffers */
static void write_buffers(const damar_write_params *s, Overlap_IO_Buffer *iobuf,
                          const char *dir1, const char *dir2, const char *ablock, const char *bblock, int lastRead)
{ int     tspace = s->trace_space;
  int     tbytes = iobuf[0].tbytes;
  int     i, j, n = 0, total = 0;
  Keyed  *all;
  char    aroot[2048], broot[2048];
  int     aid, bid;
  char    path1[4300], path2[4300];
  FILE   *out;

  for (i = 0; i < s->nthreads; i++)
    total += iobuf[i].otop;
  all = (Keyed *) malloc(sizeof(Keyed) * (size_t) (total > 0 ? total : 1));
  if (all == NULL)
    { fprintf(stderr, "[ERROR] - Write_Overlap_Buffer: Cannot create file overlap buffer for all threads\n");
      exit(1);
    }
  for (i = 0; i < s->nthreads; i++)
    { Overlap_IO_Buffer *b = iobuf + i;
      for (j = 0; j < b->otop; j++)
        { if (s->only_identity && b->ovls[j].aread != b->ovls[j].bread)
            continue;
          all[n].ovl = b->ovls + j;                       /* (the records stay where they are: 16 bytes per record are sorted) */
          all[n].trace = (b->ovls[j].path.trace != NULL) ? ((const char *) b->trace) + ((uintptr_t) b->ovls[j].path.trace - 1) : NULL;
          all[n].seq = n;
          n += 1;
        }
    }
  all = sort_keyed(all, n);

  aid = split_block_name(ablock, aroot, sizeof(aroot));
  bid = split_block_name(bblock, broot, sizeof(broot));

  if (strcmp(ablock, bblock) == 0 || s->symmetric == 0)
    { if (aid > 0)
        snprintf(path1, sizeof(path1), "%s/%s.%d.%s.%d.las", dir1, aroot, aid, broot, bid);
      else
        snprintf(path1, sizeof(path1), "%s.las", ablock);
      out = open_las(path1, tspace);
      for (j = 0; j < n; j++)
        write_keyed(out, all + j, tbytes);
      close_las(out, n);
    }
  else
    { if (aid > 0)
        { if (bid > 0)
            { snprintf(path1, sizeof(path1), "%s/%s.%d.%s.%d.las", dir1, aroot, aid, broot, bid);
              snprintf(path2, sizeof(path2), "%s/%s.%d.%s.%d.las", dir2, broot, bid, aroot, aid);
            }
          else
            { snprintf(path1, sizeof(path1), "%s/%s.%d.%s.las", dir1, aroot, aid, bblock);
              snprintf(path2, sizeof(path2), "%s.%s.%d.las", bblock, aroot, aid);
            }
        }
      else
        { if (bid > 0)
            { snprintf(path1, sizeof(path1), "%s.%s.%d.las", ablock, broot, bid);
              snprintf(path2, sizeof(path2), "%s/%s.%d.%s.las", dir2, broot, bid, ablock);
            }
          else
            { snprintf(path1, sizeof(path1), "%s.%s.las", ablock, bblock);
              snprintf(path2, sizeof(path2), "%s.%s.las", bblock, ablock);
            }
        }
      /* records whose aread lies in the lower-numbered block go to its file */
      { const char *first = (bid < aid) ? path2 : path1;
        const char *second = (bid < aid) ? path1 : path2;
        FilePart part2;
        pthread_t th;
        int       par;
        for (j = 0; j < n; j++)
          if (all[j].ovl->aread > lastRead)
            break;
        /* the two files are written side by side: the second by a thread of its own (the last block pair of a command
           is written with nothing else left to do: its two files one after the other were a tenth of the drain) */
        part2.path = second;  part2.tspace = tspace;  part2.tbytes = tbytes;  part2.recs = all + j;  part2.n = n - j;
        par = (n - j > 4096) && pthread_create(&th, NULL, write_file_part, &part2) == 0;
        { FilePart part1;
          part1.path = first;  part1.tspace = tspace;  part1.tbytes = tbytes;  part1.recs = all;  part1.n = j;
          write_file_part(&part1);
        }
        if (par)
          pthread_join(th, NULL);
        else
          write_file_part(&part2);
      }
    }
  free(all);
}


void Write_Overlap_Buffer(Align_Spec *spec, char *dir1, char *dir2, char *ablock, char *bblock, int lastRead)
{ Spec *s = (Spec *) spec;
  damar_write_params p;
  p.trace_space = s->trace_space;  p.nthreads = s->nthreads;
  p.symmetric = s->symmetric;      p.only_identity = s->only_identity;
  write_buffers(&p, s->iobuf, dir1, dir2, ablock, bblock, lastRead);
}

/* For a writer thread: take the filled buffers away from the Align_Spec (which continues with
 * fresh, empty ones) so that sorting and writing them can overlap with the next block pair. */
Overlap_IO_Buffer *damar_detach_overlap_buffers(Align_Spec *spec, damar_write_params *p)
{ Spec *s = (Spec *) spec;
  Overlap_IO_Buffer *old = s->iobuf;
  int i;
  p->trace_space = s->trace_space;  p->nthreads = s->nthreads;
  p->symmetric = s->symmetric;      p->only_identity = s->only_identity;
  s->iobuf = (Overlap_IO_Buffer *) calloc((size_t) s->nthreads, sizeof(Overlap_IO_Buffer));
  for (i = 0; i < s->nthreads; i++)
    { Overlap_IO_Buffer *b = CreateOverlapBuffer(s->nthreads, old[i].tbytes, old[i].no_trace);
      if (b == NULL)
        { fprintf(stderr, "[ERROR] - Cannot allocate overlap buffers\n");
          exit(1);
        }
      s->iobuf[i] = *b;
      free(b);
    }
  return old;
}

void damar_write_detached(const damar_write_params *p, Overlap_IO_Buffer *bufs,
                          const char *dir1, const char *dir2, const char *ablock, const char *bblock, int lastRead)
{ int i;
  write_buffers(p, bufs, dir1, dir2, ablock, bblock, lastRead);
  for (i = 0; i < p->nthreads; i++)
    { free(bufs[i].ovls);
      free(bufs[i].trace);
    }
  free(bufs);
}

/* align.c:6369-6380 */
void Reset_Overlap_Buffer(Align_Spec *spec)
{ Spec *s = (Spec *) spec;
  int   i;
  for (i = 0; i < s->nthreads; i++)
    { s->iobuf[i].otop = 0;
      s->iobuf[i].ttop = 0;
    }
}

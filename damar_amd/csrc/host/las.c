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
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>
#include <time.h>

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

/* align.c:198-199: how much of the error rate a base composition leaves, by how far A+T is from one half */
static const double bias_factor[10] = { .690, .690, .690, .690, .780, .850, .900, .933, .966, 1.000 };

/* The two 15-column tables of the trim test (align.c:234-318).  A word x holds 15 columns of an alignment, bit 14 the
 * oldest, 1 = match.  SCORE[x] = sum over its columns of (+mscore for a match, -dscore otherwise); TABLE[x] = SCORE[x]
 * minus the largest score of a proper prefix (the empty one included), i.e. TABLE[x] >= 0 iff no suffix of x scores
 * negative.  Built column by column: dropping the NEWEST column of x gives x >> 1 with one column less, so a word's
 * prefix scores are those of (x >> 1) taken as a 14-column word, and so on down to the empty word. */
static void trim_tables(int mscore, int dscore, int16 *score, int16 *table)
{ int *total = (int *) malloc(sizeof(int) * TRIM_SIZE);        /* score of the columns seen so far */
  int *best  = (int *) malloc(sizeof(int) * TRIM_SIZE);        /* largest score of a proper prefix of them */
  int  cols, x;
  if (total == NULL || best == NULL)
    { fprintf(stderr, "damar: out of memory (trim tables)\n");
      exit(1);
    }
  total[0] = best[0] = 0;                                       /* words of 0 columns: index 0 only */
  for (cols = 1; cols <= TRIM_BITS; cols++)                     /* words of `cols` columns live at indexes < 2^cols */
    for (x = (1 << cols) - 1; x >= 0; x--)                      /* (downwards: x >> 1 < x is still the shorter word's entry) */
      { const int shorter = x >> 1;
        const int before  = total[shorter];
        best[x]  = best[shorter] > before ? best[shorter] : before;
        total[x] = before + ((x & 1) ? mscore : -dscore);
      }
  for (x = 0; x < TRIM_SIZE; x++)
    { score[x] = (int16) total[x];
      table[x] = (int16) (total[x] - best[x]);
    }
  free(total);
  free(best);
}

static Overlap_IO_Buffer *set_take(int nthreads, int tbytes, int no_trace);
static void               set_give(Overlap_IO_Buffer *bufs, int nthreads);

Align_Spec *New_Align_Spec(double ave_corr, int trace_space, float *freq, int nthreads,
                           int symmetric, int only_identity, int no_trace_points, int reach)
{ Spec  *s = (Spec *) calloc(1, sizeof(Spec));
  const int tbytes = (trace_space <= TRACE_XOVR) ? 1 : 2;
  double at, left;
  int    step;

  if (s == NULL || (s->score = (int16 *) malloc(sizeof(int16) * 2 * TRIM_SIZE)) == NULL)
    { fprintf(stderr, "damar: out of memory (alignment specification)\n");
      exit(1);
    }
  s->table = s->score + TRIM_SIZE;
  memcpy(s->freq, freq, sizeof(s->freq));
  s->ave_corr = ave_corr;  s->trace_space = trace_space;  s->reach = reach;
  s->nthreads = nthreads;  s->symmetric = symmetric;      s->only_identity = only_identity;

  /* composition: the smaller of A+T and C+G, in steps of 5 % from 5 % on; below 20 % it is taken as 20 % */
  at = (double) freq[0] + freq[3];
  if (at > .5)
    at = 1. - at;
  step = (int) ((at + .025) * 20. - 1.);
  if (at < .2)
    { fprintf(stderr, "Warning: Base bias worse than 80/20%% ! (New_Align_Spec)\n");
      fprintf(stderr, "         Capping bias at this ratio.\n");
      step = 3;
    }
  left = bias_factor[step] * (1. - ave_corr);                   /* the share of columns that may differ */
  s->ave_path = (int) (PATH_LEN * (1. - left));
  { /* the two tables depend on the match score alone: a plan builds a specification per block pair (its overlap buffers
       are the pair's own), the tables of the last one are copied while the score stays what it was (128 KB instead of
       half a million table steps) */
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    static int    have = 0, last_mscore = 0;
    static int16  last[2 * TRIM_SIZE];
    const int mscore = (int) (FRACTION * left);
    pthread_mutex_lock(&mu);
    if (!have || last_mscore != mscore)
      { trim_tables(mscore, FRACTION - mscore, last, last + TRIM_SIZE);
        have = 1;  last_mscore = mscore;
      }
    memcpy(s->score, last, sizeof(last));
    pthread_mutex_unlock(&mu);
  }

  s->iobuf = set_take(nthreads, tbytes, no_trace_points);
  return (Align_Spec *) s;
}

void Free_Align_Spec(Align_Spec *spec)
{ Spec *s = (Spec *) spec;
  set_give(s->iobuf, s->nthreads);
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

/* A thread's buffer of overlaps and their traces (align.c:5969-6018): room for 500 000 / nthreads records to start with and
 * 150 trace values per record; both grow by a fifth + 1000 when they are full (AddOverlapToBuffer). */
Overlap_IO_Buffer *CreateOverlapBuffer(int nthreads, int tbytes, int no_trace)
{ Overlap_IO_Buffer *buf;
  const int records = 500000 / nthreads + 1;

  if (!no_trace && tbytes != 1 && tbytes != 2)
    { fprintf(stderr, "[ERROR] - Unsupported size of trace: %d!\n", tbytes);
      return NULL;
    }
  buf = (Overlap_IO_Buffer *) calloc(1, sizeof(Overlap_IO_Buffer));
  /* (the records are not cleared: every one is written as a whole before it is read -- AddOverlapToBuffer copies the
     struct, the writer assembles its 40 bytes field by field -- and clearing 24 MB per block pair was 1.3 ms of the
     thread that feeds the GPU) */
  if (buf == NULL || (buf->ovls = (Overlap *) malloc((size_t) records * sizeof(Overlap))) == NULL)
    return NULL;
  buf->omax = records;
  buf->no_trace = no_trace;
  if (!no_trace)
    { buf->tbytes = tbytes;
      buf->tmax   = 150ull * (uint64) records;
      if ((buf->trace = malloc((size_t) buf->tmax * (size_t) tbytes)) == NULL)
        return NULL;
    }
  return buf;
}

/* room for `more` further trace bytes behind ttop (the pool's size is tmax VALUES of tbytes bytes) */
static int trace_room(Overlap_IO_Buffer *b, uint64 more)
{ const uint64 want = b->ttop + more;
  if (want < b->tmax * (uint64) b->tbytes)
    return 0;
  do
    b->tmax = (uint64) (b->tmax * 1.2) + 1000;
  while (want >= b->tmax * (uint64) b->tbytes);
  b->trace = realloc(b->trace, (size_t) b->tmax * (size_t) b->tbytes);
  return b->trace == NULL;
}

/* align.c:6020-6102.  A record's trace is kept as its byte offset in the pool + 1 while the pool may still move; the
 * offsets become pointers when the buffer is drained (write_buffers). */
int AddOverlapToBuffer(Overlap_IO_Buffer *b, Overlap *ovl, int tbytes)
{ const int traced = b != NULL && ovl->path.trace != NULL && !b->no_trace;
  Overlap  *slot;

  if (b == NULL)
    { fprintf(stderr, "[ERROR] - Cannot add overlap to Overlap_IO_Buffer. Buffer is NULL!\n");
      return 1;
    }
  if (b->otop == b->omax)
    { b->omax = (int) (b->omax * 1.2) + 1000;
      if ((b->ovls = (Overlap *) realloc(b->ovls, sizeof(Overlap) * (size_t) b->omax)) == NULL)
        { fprintf(stderr, "[ERROR] - Cannot increase overlap buffer size to %d!\n", b->omax);
          return 1;
        }
    }
  slot  = b->ovls + b->otop++;
  *slot = *ovl;                                   /* reads, flags, end points, differences */
  slot->path.trace = NULL;
  slot->path.tlen  = 0;
  if (traced)
    { const uint64 bytes = (uint64) tbytes * (uint64) ovl->path.tlen;
      if (trace_room(b, bytes))
        { fprintf(stderr, "[ERROR] - Cannot increase trace point buffer size to %llu!\n", (unsigned long long) b->tmax);
          return 1;
        }
      memcpy((char *) b->trace + b->ttop, ovl->path.trace, (size_t) bytes);
      slot->path.tlen  = ovl->path.tlen;
      slot->path.trace = (void *) (uintptr_t) (b->ttop + 1);
      b->ttop += bytes;
    }
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
    { if (trace_room(dst, src->ttop))
        return 1;
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

/* A trace of 16-bit values narrowed to bytes in place (align.c:3375-3396); with `check`, a value that does not fit a byte
 * ends the program.  The largest value is looked for first: nothing is narrowed of a trace that does not fit. */
int Compress_TraceTo8(Overlap *ovl, int check)
{ const int     n    = ovl->path.tlen;
  const uint16 *wide = (const uint16 *) ovl->path.trace;
  uint8        *out  = (uint8 *) ovl->path.trace;
  int           k;
  if (check)
    { uint16 top = 0;
      for (k = 0; k < n; k++)
        if (wide[k] > top)
          top = wide[k];
      if (top > 255)
        { fprintf(stderr, "damar: Compression of trace to bytes fails, value too big\n");
          exit(1);
        }
    }
  for (k = 0; k < n; k++)
    out[k] = (uint8) wide[k];
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

typedef struct                   /* one gathered record: where it lies, its place in the gather, and its sort keys */
{ const Overlap *ovl;
  const char    *trace;          /* its trace bytes (NULL: none) */
  int            aread, bread;   /* the keys of the order below, copied while the records are gathered -- they are read in */
  uint64         k2, k3;         /* order there: sorting then stays inside this array instead of chasing 35 000 pointers into */
  int            seq;            /* the threads' buffers.  k2 = comp | abpos | aepos, k3 = bbpos | bepos (none is negative) */
} Keyed;

static void keyed_set(Keyed *k, const Overlap *v, const char *trace, int seq)
{ k->ovl = v;  k->trace = trace;  k->seq = seq;
  k->aread = v->aread;  k->bread = v->bread;
  k->k2 = ((uint64) (COMP(v->flags) ? 1 : 0) << 63) | ((uint64) (uint32) v->path.abpos << 32) | (uint64) (uint32) v->path.aepos;
  k->k3 = ((uint64) (uint32) v->path.bbpos << 32) | (uint64) (uint32) v->path.bepos;
}

/* align.c:6104-6164 SORT_OVL (aread, bread, COMP, abpos, aepos, bbpos, bepos); the reference's last key is the record's
 * address in the gathered array, here its gather index (same order for a stable gather). */
static int by_overlap(const void *x, const void *y)
{ const Keyed *l = (const Keyed *) x, *r = (const Keyed *) y;
  if (l->aread != r->aread) return l->aread - r->aread;
  if (l->bread != r->bread) return l->bread - r->bread;
  if (l->k2 != r->k2) return (l->k2 < r->k2) ? -1 : 1;
  if (l->k3 != r->k3) return (l->k3 < r->k3) ? -1 : 1;
  return (l->seq < r->seq) ? -1 : (l->seq > r->seq);
}

/* A .las file under construction: the number of records (8 bytes, filled in when the file is closed), the trace spacing
 * (4), then the records.  Records are assembled in a buffer of the writer's own and leave with one write() per 4 MB:
 * through stdio a record was two fwrite calls, which was most of a writer thread's time (1.7 M records per config-2
 * step). */
typedef struct { int fd; char *buf; size_t fill, cap; const char *path; } LasOut;

static void las_fail(const LasOut *o, const char *what)
{ fprintf(stderr, "[ERROR] - Write_Overlap_Buffer: Cannot %s file %s\n", what, o->path);
  exit(1);
}

/* What the writers have produced since the process started (bytes, files, records): the output rate of a long plan is
   stated beside its compute rate.  DAMAR_LAS_KEEP names a text file of path endings; with it set only
   the .las files whose path ends in one of them are written where they belong, every other file goes through the same
   record assembly and write() calls into /dev/null -- a measurement aid for plans whose output (config 4: ~55 GB) is not
   wanted on the box, with the sampled files kept for their md5s. */
static int64 LAS_total[4];               /* bytes, files, records, aligned bp = sum of aepos - abpos over the records */
void damar_las_totals(int64 *out)
{ int i;
  for (i = 0; i < 4; i++)
    out[i] = __atomic_load_n(&LAS_total[i], __ATOMIC_RELAXED);
}

static char **LAS_keep;
static int    LAS_nkeep = -1;          /* -1: DAMAR_LAS_KEEP not read yet */
static pthread_mutex_t LAS_keep_mu = PTHREAD_MUTEX_INITIALIZER;

static int las_is_kept(const char *path)
{ int i;
  size_t lp = strlen(path);
  pthread_mutex_lock(&LAS_keep_mu);
  if (LAS_nkeep < 0)
    { const char *lst = getenv("DAMAR_LAS_KEEP");
      LAS_nkeep = 0;
      if (lst != NULL && lst[0] != '\0')
        { FILE *f = fopen(lst, "r");
          char  ln[4400];
          if (f == NULL)
            { fprintf(stderr, "damar: cannot read DAMAR_LAS_KEEP=%s\n", lst);
              exit(1);
            }
          LAS_keep = (char **) malloc(sizeof(char *));
          LAS_keep[0] = NULL;                        /* (set, possibly empty: everything else is discarded) */
          while (fgets(ln, sizeof(ln), f) != NULL)
            { size_t n = strlen(ln);
              while (n > 0 && (ln[n - 1] == '\n' || ln[n - 1] == ' '))
                ln[--n] = '\0';
              if (n == 0)
                continue;
              LAS_keep = (char **) realloc(LAS_keep, sizeof(char *) * (size_t) (LAS_nkeep + 2));
              LAS_keep[LAS_nkeep++] = strdup(ln);
            }
          fclose(f);
        }
    }
  pthread_mutex_unlock(&LAS_keep_mu);
  if (LAS_keep == NULL)
    return 1;
  for (i = 0; i < LAS_nkeep; i++)
    { size_t lk = strlen(LAS_keep[i]);
      if (lk <= lp && strcmp(path + (lp - lk), LAS_keep[i]) == 0)
        return 1;
    }
  return 0;
}

static void las_flush(LasOut *o)
{ size_t done = 0;
  __atomic_fetch_add(&LAS_total[0], (int64) o->fill, __ATOMIC_RELAXED);
  while (done < o->fill)
    { const ssize_t w = write(o->fd, o->buf + done, o->fill - done);
      if (w < 0)
        { if (errno == EINTR)
            continue;
          las_fail(o, "write to");
        }
      done += (size_t) w;
    }
  o->fill = 0;
}

static void las_put(LasOut *o, const void *src, size_t n)
{ if (o->fill + n > o->cap)
    { las_flush(o);
      if (n > o->cap)                              /* (a trace longer than the buffer: straight through) */
        { LasOut big = *o;
          big.buf = (char *) src;  big.fill = n;
          las_flush(&big);
          return;
        }
    }
  memcpy(o->buf + o->fill, src, n);
  o->fill += n;
}

/* a file's 4 MB of assembly buffer: kept between files (a fresh one is 1 000 page faults while it fills, per file) */
#define LAS_BUF   ((size_t) 4 << 20)
#define LAS_BUFS  8
static char *LAS_buf[LAS_BUFS];
static int   LAS_nbuf;
static pthread_mutex_t LAS_buf_mu = PTHREAD_MUTEX_INITIALIZER;

static char *las_buf_take(void)
{ char *b = NULL;
  pthread_mutex_lock(&LAS_buf_mu);
  if (LAS_nbuf > 0)
    b = LAS_buf[--LAS_nbuf];
  pthread_mutex_unlock(&LAS_buf_mu);
  return b != NULL ? b : (char *) malloc(LAS_BUF);
}

static void las_buf_give(char *b)
{ pthread_mutex_lock(&LAS_buf_mu);
  if (LAS_nbuf < LAS_BUFS)
    { LAS_buf[LAS_nbuf++] = b;
      b = NULL;
    }
  pthread_mutex_unlock(&LAS_buf_mu);
  free(b);
}

static LasOut las_open(const char *path, int tspace)
{ LasOut  o;
  int64   none = 0;
  o.path = path;
  o.cap  = LAS_BUF;
  o.fill = 0;
  o.buf  = las_buf_take();
  o.fd   = las_is_kept(path) ? open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666) : open("/dev/null", O_WRONLY);
  if (o.fd < 0 || o.buf == NULL)
    las_fail(&o, "open for writing");
  las_put(&o, &none, sizeof(none));
  las_put(&o, &tspace, sizeof(tspace));
  return o;
}

/* the 40 bytes of a record on disk (the Overlap minus its leading pointer, align.c:3345-3373) and its trace */
static void las_record(LasOut *o, const Keyed *k, int tbytes)
{ const Overlap *v = k->ovl;
  int32_t rec[10];
  rec[0] = v->path.tlen;   rec[1] = v->path.diffs;
  rec[2] = v->path.abpos;  rec[3] = v->path.bbpos;  rec[4] = v->path.aepos;  rec[5] = v->path.bepos;
  rec[6] = (int32_t) v->flags;
  rec[7] = v->aread;       rec[8] = v->bread;       rec[9] = 0;
  las_put(o, rec, sizeof(rec));
  if (k->trace != NULL && v->path.tlen > 0)
    las_put(o, k->trace, (size_t) tbytes * (size_t) v->path.tlen);
}

static void las_close(LasOut *o, int64 n)
{ las_flush(o);
  __atomic_fetch_add(&LAS_total[1], 1, __ATOMIC_RELAXED);
  __atomic_fetch_add(&LAS_total[2], n, __ATOMIC_RELAXED);
  if (pwrite(o->fd, &n, sizeof(n), 0) != (ssize_t) sizeof(n) || close(o->fd) != 0)
    las_fail(o, "finish");
  las_buf_give(o->buf);
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
/* The two arrays a gather is sorted through are the calling thread's own and are kept between block pairs while they are
   small: a writer thread serves thousands of pairs of some 35 000 records, and 2 x 1.7 MB from malloc are mapped, faulted
   in page by page and unmapped again for each of them. */
#define SCRATCH_KEEP  ((size_t) 64 << 20)
static __thread Keyed  *KS_buf[2];
static __thread size_t  KS_cap[2];

static Keyed *scratch(int which, size_t n)
{ if (KS_cap[which] < n)
    { const size_t cap = n + n / 4 + 1024;
      free(KS_buf[which]);
      KS_buf[which] = (Keyed *) malloc(sizeof(Keyed) * cap);
      KS_cap[which] = (KS_buf[which] != NULL) ? cap : 0;
    }
  return KS_buf[which];
}

static void scratch_done(void)
{ int w;
  for (w = 0; w < 2; w++)
    if (KS_cap[w] * sizeof(Keyed) > SCRATCH_KEEP)
      { free(KS_buf[w]);  KS_buf[w] = NULL;  KS_cap[w] = 0; }
}

/* (all = scratch(0, ..); the result is all itself or scratch 1) */
static Keyed *sort_keyed(Keyed *all, int n)
{ int     lo, hi, i, j, k;
  int    *first;
  Keyed  *out;

  if (n < 64)
    { qsort(all, (size_t) n, sizeof(Keyed), by_overlap);
      return all;
    }
  lo = hi = all[0].aread;
  for (i = 1; i < n; i++)
    { if (all[i].aread < lo) lo = all[i].aread;
      if (all[i].aread > hi) hi = all[i].aread;
    }
  first = (int *) calloc((size_t) (hi - lo) + 2, sizeof(int));
  out   = scratch(1, (size_t) n);
  if (first == NULL || out == NULL)
    { free(first);
      qsort(all, (size_t) n, sizeof(Keyed), by_overlap);
      return all;
    }
  for (i = 0; i < n; i++)
    first[all[i].aread - lo + 1] += 1;
  for (i = 1; i <= hi - lo + 1; i++)
    first[i] += first[i - 1];
  for (i = 0; i < n; i++)
    out[first[all[i].aread - lo]++] = all[i];
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
  return out;
}

typedef struct { const char *path; int tspace, tbytes; const Keyed *recs; int n; } FilePart;

static void *write_file_part(void *arg)
{ const FilePart *f = (const FilePart *) arg;
  LasOut out = las_open(f->path, f->tspace);
  int64  bp = 0;
  int    j;
  for (j = 0; j < f->n; j++)
    { if (j + 12 < f->n)                            /* the records lie where the tails left them, in work-item order */
        { __builtin_prefetch(f->recs[j + 12].ovl);
          __builtin_prefetch(f->recs[j + 12].trace);
        }
      las_record(&out, f->recs + j, f->tbytes);
      bp += (int64) (uint32) f->recs[j].k2 - (int64) (uint32) ((f->recs[j].k2 >> 32) & 0x7fffffffu);
    }
  __atomic_fetch_add(&LAS_total[3], bp, __ATOMIC_RELAXED);
  las_close(&out, f->n);
  return NULL;
}

/* align.c:6166-6367 on an explicit set of per-thread buffers */
/* DAMAR_HOSTPROF: where a writer's time goes (gathering the records with their keys, sorting, writing), summed over the
   writers and printed when the process ends */
static double W_ms[3];
static int    W_prof = -1;
static pthread_mutex_t W_mu = PTHREAD_MUTEX_INITIALIZER;
static double w_now(void)
{ struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
static void w_report(void)
{ fprintf(stderr, "damar .las writers, ms: gather %.1f, sort %.1f, write %.1f\n", W_ms[0], W_ms[1], W_ms[2]);
}

static void write_buffers(const damar_write_params *s, Overlap_IO_Buffer *iobuf,
                          const char *dir1, const char *dir2, const char *ablock, const char *bblock, int lastRead)
{ int     tspace = s->trace_space;
  int     tbytes = iobuf[0].tbytes;
  int     i, j, n = 0, total = 0;
  Keyed  *all;
  char    aroot[2048], broot[2048];
  int     aid, bid;
  char    path1[4300], path2[4300];

  double  w0, w1, w2;
  pthread_mutex_lock(&W_mu);
  if (W_prof < 0)
    { W_prof = getenv("DAMAR_HOSTPROF") != NULL;
      if (W_prof)
        atexit(w_report);
    }
  pthread_mutex_unlock(&W_mu);
  w0 = W_prof ? w_now() : 0.;
  for (i = 0; i < s->nthreads; i++)
    total += iobuf[i].otop;
  all = scratch(0, (size_t) (total > 0 ? total : 1));
  if (all == NULL)
    { fprintf(stderr, "[ERROR] - Write_Overlap_Buffer: Cannot create file overlap buffer for all threads\n");
      exit(1);
    }
  for (i = 0; i < s->nthreads; i++)
    { Overlap_IO_Buffer *b = iobuf + i;
      for (j = 0; j < b->otop; j++)
        { if (s->only_identity && b->ovls[j].aread != b->ovls[j].bread)
            continue;
          /* (the records stay where they are: a reference with its keys is sorted) */
          keyed_set(all + n, b->ovls + j,
                    (b->ovls[j].path.trace != NULL) ? ((const char *) b->trace) + ((uintptr_t) b->ovls[j].path.trace - 1) : NULL, n);
          n += 1;
        }
    }
  w1 = W_prof ? w_now() : 0.;
  all = sort_keyed(all, n);
  w2 = W_prof ? w_now() : 0.;

  aid = split_block_name(ablock, aroot, sizeof(aroot));
  bid = split_block_name(bblock, broot, sizeof(broot));

  if (strcmp(ablock, bblock) == 0 || s->symmetric == 0)
    { if (aid > 0)
        snprintf(path1, sizeof(path1), "%s/%s.%d.%s.%d.las", dir1, aroot, aid, broot, bid);
      else
        snprintf(path1, sizeof(path1), "%s.las", ablock);
      { FilePart whole;
        whole.path = path1;  whole.tspace = tspace;  whole.tbytes = tbytes;  whole.recs = all;  whole.n = n;
        write_file_part(&whole);
      }
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
          if (all[j].aread > lastRead)
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
  scratch_done();
  if (W_prof)
    { const double w3 = w_now();
      pthread_mutex_lock(&W_mu);
      W_ms[0] += w1 - w0;  W_ms[1] += w2 - w1;  W_ms[2] += w3 - w2;
      pthread_mutex_unlock(&W_mu);
    }
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
/* Sets of buffers that are done with (written out, or never filled) wait here for the next block pair: a fresh set is
   16 x 12 MB of address space whose pages are faulted in as the tails fill them and unmapped again behind every pair --
   a plan makes an Align_Spec per block pair and detaches its buffers for the writer. */
#define SET_POOL      8
#define SET_POOL_MAX  (1ull << 30)     /* a set that has grown beyond this is freed, not kept */
static struct { Overlap_IO_Buffer *bufs; int nthreads, tbytes, no_trace; } SET_pool[SET_POOL];
static int SET_n;
static pthread_mutex_t SET_mu = PTHREAD_MUTEX_INITIALIZER;

static Overlap_IO_Buffer *set_take(int nthreads, int tbytes, int no_trace)
{ Overlap_IO_Buffer *bufs;
  int i;
  pthread_mutex_lock(&SET_mu);
  for (i = 0; i < SET_n; i++)
    if (SET_pool[i].nthreads == nthreads && SET_pool[i].no_trace == no_trace && (no_trace || SET_pool[i].tbytes == tbytes))
      { bufs = SET_pool[i].bufs;
        SET_pool[i] = SET_pool[--SET_n];
        pthread_mutex_unlock(&SET_mu);
        return bufs;
      }
  pthread_mutex_unlock(&SET_mu);
  bufs = (Overlap_IO_Buffer *) calloc((size_t) (nthreads > 0 ? nthreads : 1), sizeof(Overlap_IO_Buffer));
  for (i = 0; bufs != NULL && i < nthreads; i++)
    { Overlap_IO_Buffer *one = CreateOverlapBuffer(nthreads, tbytes, no_trace);
      if (one == NULL)
        { bufs = NULL;  break; }
      bufs[i] = *one;
      free(one);
    }
  if (bufs == NULL)
    { fprintf(stderr, "[ERROR] - Cannot allocate overlap buffers\n");
      exit(1);
    }
  return bufs;
}

static void set_give(Overlap_IO_Buffer *bufs, int nthreads)
{ uint64 held = 0;
  int    i;
  for (i = 0; i < nthreads; i++)
    { bufs[i].otop = 0;  bufs[i].ttop = 0;
      held += (uint64) bufs[i].omax * sizeof(Overlap) + (bufs[i].no_trace ? 0 : bufs[i].tmax * (uint64) bufs[i].tbytes);
    }
  pthread_mutex_lock(&SET_mu);
  if (SET_n < SET_POOL && nthreads > 0 && held <= SET_POOL_MAX)
    { SET_pool[SET_n].bufs = bufs;  SET_pool[SET_n].nthreads = nthreads;
      SET_pool[SET_n].tbytes = bufs[0].tbytes;  SET_pool[SET_n].no_trace = bufs[0].no_trace;
      SET_n += 1;
      pthread_mutex_unlock(&SET_mu);
      return;
    }
  pthread_mutex_unlock(&SET_mu);
  for (i = 0; i < nthreads; i++)
    { free(bufs[i].ovls);
      if (!bufs[i].no_trace)
        free(bufs[i].trace);
    }
  free(bufs);
}

Overlap_IO_Buffer *damar_detach_overlap_buffers(Align_Spec *spec, damar_write_params *p)
{ Spec *s = (Spec *) spec;
  Overlap_IO_Buffer *old = s->iobuf;
  p->trace_space = s->trace_space;  p->nthreads = s->nthreads;
  p->symmetric = s->symmetric;      p->only_identity = s->only_identity;
  s->iobuf = set_take(s->nthreads, old[0].tbytes, old[0].no_trace);
  return old;
}

void damar_write_detached(const damar_write_params *p, Overlap_IO_Buffer *bufs,
                          const char *dir1, const char *dir2, const char *ablock, const char *bblock, int lastRead)
{ write_buffers(p, bufs, dir1, dir2, ablock, bblock, lastRead);
  set_give(bufs, p->nthreads);
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

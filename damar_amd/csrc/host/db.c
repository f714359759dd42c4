/* db.c -- read-block loader, reverse complement and synthetic DB generator.
 *
 * Host side of the daligner overlap path: everything here runs on the CPU
 * before the first kernel launch.  Written from the on-disk formats
 *  and the behaviour of the reference routines cited per
 * function; no reference code is reused.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <strings.h>
#include <math.h>
#include <errno.h>
#include <sys/stat.h>
#include <unistd.h>

#include <zlib.h>

#include "damar_db.h"

static void *xmalloc(size_t n, const char *what)
{ void *p = malloc(n ? n : 1);
  if (p == NULL)
    { fprintf(stderr, "damar: out of memory (%s, %zu bytes)\n", what, n);
      exit(1);
    }
  return p;
}

/* "dir/name.suffix" -> "name"  (db/DB.c:175-210 Root with a suffix) */
char *damar_root(const char *name, const char *suffix)
{ const char *base = strrchr(name, '/');
  size_t      len, slen;
  char       *out;

  base = (base == NULL) ? name : base + 1;
  len  = strlen(base);
  slen = (suffix == NULL) ? 0 : strlen(suffix);
  if (slen > 0 && len > slen && strcasecmp(base + (len - slen), suffix) == 0)
    len -= slen;
  out = (char *) xmalloc(len + 1, "root");
  memcpy(out, base, len);
  out[len] = '\0';
  return out;
}

static char *dir_of(const char *name)
{ const char *slash = strrchr(name, '/');
  char       *out;
  if (slash == NULL)
    return strdup(".");
  out = (char *) xmalloc((size_t) (slash - name) + 1, "dir");
  memcpy(out, name, (size_t) (slash - name));
  out[slash - name] = '\0';
  return out;
}

/* d<run:3>_<block:5>, "." for the unsplit DB (db/DB.c:1851-1933 getDir) */
char *damar_get_dir(int run, int block)
{ char *out = (char *) xmalloc(40, "dir name");
  if (block == 0)
    strcpy(out, ".");
  else
    sprintf(out, "d%03d_%05d", run, block);
  return out;
}

/* Open_DB (db/DB.c:457-680) + Read_All_Sequences (db/DB.c:1547-1608). */
static uint32 unpack4[256];            /* the four bases of a .bps byte, first base in the low byte */

static void unpack4_init(void)
{ int b;
  if (unpack4[255] != 0)
    return;
  for (b = 0; b < 256; b++)
    unpack4[b] = (uint32) ((b >> 6) & 3) | ((uint32) ((b >> 4) & 3) << 8) | ((uint32) ((b >> 2) & 3) << 16) | ((uint32) (b & 3) << 24);
}

/* Sequence arrays are hundreds of MB that are written once, front to back: on 2 MB pages (where the system hands them out
   on request) first touch costs a 512th of the page faults. */
#include <sys/mman.h>
static void *big_alloc(size_t n, const char *what)
{ void *p = NULL;
  if (n >= ((size_t) 8 << 20) && posix_memalign(&p, (size_t) 2 << 20, n) == 0 && p != NULL)
    {
#ifdef MADV_HUGEPAGE
      (void) madvise(p, n, MADV_HUGEPAGE);
#endif
      return p;
    }
  return xmalloc(n, what);
}

/* the bases of `nbytes` .bps bytes (four per byte, first base in the top bits), one byte per base */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("bmi2")))
static void unpack_bmi2(const unsigned char *src, size_t nbytes, char *dst)
{ size_t j = 0;
  for (; j + 8 <= nbytes; j += 8)
    { uint64 x, r;
      memcpy(&x, src + j, 8);
      /* reverse the four 2-bit groups inside every byte: the first base goes to the low bits, where pdep starts */
      r = ((x & 0x0303030303030303ull) << 6) | ((x & 0x0c0c0c0c0c0c0c0cull) << 2) |
          ((x >> 2) & 0x0c0c0c0c0c0c0c0cull) | ((x >> 6) & 0x0303030303030303ull);
      { uint64 o0 = _pdep_u64(r, 0x0303030303030303ull), o1 = _pdep_u64(r >> 16, 0x0303030303030303ull),
               o2 = _pdep_u64(r >> 32, 0x0303030303030303ull), o3 = _pdep_u64(r >> 48, 0x0303030303030303ull);
        memcpy(dst + 4 * j, &o0, 8);  memcpy(dst + 4 * j + 8, &o1, 8);
        memcpy(dst + 4 * j + 16, &o2, 8);  memcpy(dst + 4 * j + 24, &o3, 8);
      }
    }
  for (; j < nbytes; j++)
    memcpy(dst + 4 * j, &unpack4[src[j]], 4);
}
#endif

static void unpack_bytes(const unsigned char *src, size_t nbytes, char *dst)
{
#if defined(__x86_64__)
  static int have = -1;
  if (have < 0)
    have = (__builtin_cpu_supports("bmi2") && getenv("DAMAR_DB_NO_BMI2") == NULL) ? 1 : 0;       /* (the variable: a test hook) */
  if (have)
    { unpack_bmi2(src, nbytes, dst);
      return;
    }
#endif
  { size_t j;
    for (j = 0; j < nbytes; j++)
      memcpy(dst + 4 * j, &unpack4[src[j]], 4);
  }
}

static int read_block_impl(const char *name, HITS_DB *block, damar_packed *pk)
{ char   *root = damar_root(name, ".db");
  char   *dir  = dir_of(name);
  char   *dot;
  int     part = 0;
  char    path[4096];
  FILE   *stub = NULL, *idx = NULL, *bps = NULL;
  int     nfiles, nblocks, ufirst, ulast, i;
  long long bsize;
  HITS_READ *reads;

  unpack4_init();
  dot = strrchr(root, '.');
  if (dot != NULL && dot[1] != '\0' && dot[1] != '-')
    { char *end;
      long  v = strtol(dot + 1, &end, 10);
      if (*end == '\0' && v > 0)
        { part = (int) v;
          *dot = '\0';
        }
    }

  snprintf(path, sizeof(path), "%s/%s.db", dir, root);
  if ((stub = fopen(path, "r")) == NULL)
    { fprintf(stderr, "damar: cannot open database stub %s\n", path);
      goto fail;
    }
  snprintf(path, sizeof(path), "%s/.%s.idx", dir, root);
  if ((idx = fopen(path, "r")) == NULL)
    { fprintf(stderr, "damar: cannot open index %s\n", path);
      goto fail;
    }
  if (fread(block, sizeof(HITS_DB), 1, idx) != 1)
    { fprintf(stderr, "damar: index %s is junk\n", path);
      goto fail;
    }

  if (fscanf(stub, "files = %9d\n", &nfiles) != 1)
    goto junk;
  for (i = 0; i < nfiles; i++)
    { char a[4096], b[4096];
      int  last;
      if (fscanf(stub, "  %9d %4095s %4095s\n", &last, a, b) != 3)
        goto junk;
    }
  nblocks = 0;
  if (fscanf(stub, "blocks = %9d\n", &nblocks) != 1)
    { if (part != 0)
        { fprintf(stderr, "damar: DB %s is not partitioned, cannot request block %d\n", root, part);
          goto fail;
        }
    }
  else
    { if (fscanf(stub, "size = %9lld\n", &bsize) != 1)
        goto junk;
      if (part > nblocks)
        { fprintf(stderr, "damar: DB %s has only %d blocks\n", root, nblocks);
          goto fail;
        }
    }
  if (part > 0)
    { for (i = 1; i <= part; i++)
        if (fscanf(stub, " %9d\n", &ufirst) != 1)
          goto junk;
      if (fscanf(stub, " %9d\n", &ulast) != 1)
        goto junk;
    }
  else
    { ufirst = 0;
      ulast  = block->ureads;
    }

  block->nreads = ulast - ufirst;
  block->part   = part;
  block->ufirst = ufirst;
  block->tracks = NULL;

  reads = (HITS_READ *) xmalloc(sizeof(HITS_READ) * (size_t) (block->nreads + 2), "read index");
  reads += 1;
  if (fseeko(idx, (off_t) (sizeof(HITS_DB) + sizeof(HITS_READ) * (size_t) ufirst), SEEK_SET) != 0 ||
      fread(reads, sizeof(HITS_READ), (size_t) block->nreads, idx) != (size_t) block->nreads)
    { fprintf(stderr, "damar: index of %s is truncated\n", root);
      goto fail;
    }
  if (part > 0)
    { int64 tot = 0;
      int   mx  = 0;
      for (i = 0; i < block->nreads; i++)
        { tot += reads[i].rlen;
          if (reads[i].rlen > mx)
            mx = reads[i].rlen;
        }
      block->totlen = tot;
      block->maxlen = mx;
    }
  ((int *) reads)[-1] = block->nreads;
  block->reads = reads;

  snprintf(path, sizeof(path), "%s/.%s", dir, root);        /* db/DB.c Open_DB: path = pwd "/." root */
  block->path = strdup(path);
  strcat(path, ".bps");
  if ((bps = fopen(path, "r")) == NULL)
    { fprintf(stderr, "damar: cannot open bases %s\n", path);
      goto fail;
    }

  { char  *seq;
    int64  o = 0;
    /* the reads of a block lie back to back in the .bps file: ONE read of the stretch (a seek + read per read was 11 600
       system calls for a 135 Mbp block), then every read is unpacked out of it, whole bytes eight at a time */
    const int64 f0 = block->nreads > 0 ? reads[0].boff : 0;
    int64  f1 = f0;
    unsigned char *raw = NULL;
    for (i = 0; i < block->nreads; i++)
      { const int64 e = reads[i].boff + ((reads[i].rlen + 3) >> 2);
        if (reads[i].boff < f0)
          { f1 = -1;  break; }
        if (e > f1) f1 = e;
      }
    if (f1 >= f0 && f1 - f0 <= 2 * (block->totlen / 4 + block->nreads) + 1024 &&        /* (a stretch, not a scatter) */
        getenv("DAMAR_DB_READ_BY_READ") == NULL)                                          /* (test hook: the read-by-read path) */
      { raw = (unsigned char *) big_alloc((size_t) (f1 - f0) + 64, "packed bases");
        if (fseeko(bps, (off_t) f0, SEEK_SET) != 0 || (f1 > f0 && fread(raw, (size_t) (f1 - f0), 1, bps) != 1))
          { fprintf(stderr, "damar: read of %s failed\n", path);
            free(raw);
            goto fail;
          }
      }

    if (pk != NULL && raw != NULL && f1 - f0 < 0xffffffffll)
      { /* the caller keeps the block as it lies in the file */
        { static int64 serial = 0;
          pk->serial = __atomic_add_fetch(&serial, 1, __ATOMIC_RELAXED);
        }
        pk->raw  = raw;
        pk->nraw = f1 - f0;
        pk->foff = (uint32 *) xmalloc(sizeof(uint32) * (size_t) (block->nreads + 1), "read offsets");
        for (i = 0; i < block->nreads; i++)
          { pk->foff[i] = (uint32) (reads[i].boff - f0);
            reads[i].boff = o;
            o += reads[i].rlen + 1;
          }
        reads[block->nreads].boff = o;
        block->bases  = NULL;
        block->loaded = 0;
        fclose(bps);
        fclose(idx);
        fclose(stub);
        free(dir);
        free(root);
        return 0;
      }
    seq = (char *) big_alloc((size_t) (block->totlen + block->nreads + 4) + 64, "block bases");
    *seq++ = 4;
    for (i = 0; i < block->nreads; i++)
      { int    len  = reads[i].rlen;
        int    clen = (len + 3) >> 2;
        char  *s = seq + o;
        const unsigned char *p;
        unsigned char *own = NULL;

        if (raw != NULL)
          p = raw + (reads[i].boff - f0);
        else
          { own = (unsigned char *) xmalloc((size_t) clen + 8, "packed read");
            if (fseeko(bps, (off_t) reads[i].boff, SEEK_SET) != 0 ||
                (clen > 0 && fread(own, (size_t) clen, 1, bps) != 1))
              { fprintf(stderr, "damar: read of %s failed\n", path);
                free(own);
                goto fail;
              }
            p = own;
          }
        if (clen > 0)
          { unsigned byte = p[clen - 1];          /* the last, possibly partial, byte base by base */
            int      q = 4 * (clen - 1);
            unpack_bytes(p, (size_t) (clen - 1), s);
            s[q] = (char) ((byte >> 6) & 3);
            if (q + 1 < len) s[q + 1] = (char) ((byte >> 4) & 3);
            if (q + 2 < len) s[q + 2] = (char) ((byte >> 2) & 3);
            if (q + 3 < len) s[q + 3] = (char) (byte & 3);
          }
        free(own);
        s[len] = 4;
        reads[i].boff = o;
        o += len + 1;
      }
    free(raw);
    reads[block->nreads].boff = o;
    block->bases  = (void *) seq;
    block->loaded = 1;
  }

  fclose(bps);
  fclose(idx);
  fclose(stub);
  free(dir);
  free(root);
  return 0;

junk:
  fprintf(stderr, "damar: stub file of %s is junk\n", root);
fail:
  if (bps) fclose(bps);
  if (idx) fclose(idx);
  if (stub) fclose(stub);
  free(dir);
  free(root);
  return -1;
}

int damar_read_block(const char *name, HITS_DB *block)
{ return read_block_impl(name, block, NULL);
}

int damar_read_block_packed(const char *name, HITS_DB *block, damar_packed *pk)
{ int r;
  pk->raw = NULL;  pk->nraw = 0;  pk->foff = NULL;  pk->serial = 0;
  r = read_block_impl(name, block, pk);
  if (r != 0)
    return -1;
  return pk->raw != NULL ? 0 : 1;
}

void damar_free_packed(damar_packed *pk)
{ free(pk->raw);
  free(pk->foff);
  pk->raw = NULL;  pk->foff = NULL;  pk->nraw = 0;
}

void damar_unpack_read(const damar_packed *pk, const HITS_DB *block, int r, int comp, char *dst)
{ const int len = block->reads[r].rlen;
  const unsigned char *p = pk->raw + pk->foff[r];
  int x;
  for (x = 0; x < len; x++)
    { const int b = (p[x >> 2] >> (6 - 2 * (x & 3))) & 3;
      if (comp)
        dst[len - 1 - x] = (char) (3 - b);
      else
        dst[x] = (char) b;
    }
  dst[-1] = 4;
  dst[len] = 4;
}

static void free_tracks(HITS_TRACK *t)
{ while (t != NULL)
    { HITS_TRACK *n = t->next;
      free(t->name);
      free(t->anno);
      free(t->data);
      free(t);
      t = n;
    }
}

void damar_close_block(HITS_DB *block)
{ free_tracks(block->tracks);
  block->tracks = NULL;
  if (block->loaded && block->bases != NULL)
    free(((char *) block->bases) - 1);
  block->bases = NULL;
  if (block->reads != NULL)
    free(block->reads - 1);
  block->reads = NULL;
  if (block->path != NULL)
    free(block->path);
  block->path = NULL;
}

/* daligner.c:511-570: reverse-complement every read; freq[] is mirrored too. */
/* daligner.c:511-570: reverse complement of one read in place.  Eight bases per step from both ends:
   the complement of a base 0..3 is 3 - x = x ^ 3, the reversal a byte swap of the 64-bit word. */
static void rc_inplace(char *s, int len)
{ char *a = s, *b = s + len;
  while (b - a >= 16)
    { uint64 x, y;
      b -= 8;
      memcpy(&x, a, 8);
      memcpy(&y, b, 8);
      x = __builtin_bswap64(x) ^ 0x0303030303030303ull;
      y = __builtin_bswap64(y) ^ 0x0303030303030303ull;
      memcpy(a, &y, 8);
      memcpy(b, &x, 8);
      a += 8;
    }
  b -= 1;
  while (a < b)
    { char c = *a;
      *a++ = (char) (3 - *b);
      *b-- = (char) (3 - c);
    }
  if (a == b)
    *a = (char) (3 - *a);
}

/* dst[j] = 3 - src[len - 1 - j], eight bases at a time */
static void rc_copy(char *dst, const char *src, int len)
{ int j = 0;
  for (; j + 8 <= len; j += 8)
    { uint64 x;
      memcpy(&x, src + len - 8 - j, 8);
      x = __builtin_bswap64(x) ^ 0x0303030303030303ull;
      memcpy(dst + j, &x, 8);
    }
  for (; j < len; j++)
    dst[j] = (char) (3 - src[len - 1 - j]);
}

/* The mask intervals of a block seen from the other strand (what daligner.c:572-626 leaves behind): an interval
   [b, e) of a read of rlen bases is [rlen - e, rlen - b) there, and a read's intervals come in the opposite order.  A
   read's stretch of the data array is its interval ends in ascending order; seen as ONE list of positions p_0 < p_1 < ...
   < p_{n-1}, the other strand's list is rlen - p_{n-1}, ..., rlen - p_0: entry q of the result is rlen minus entry
   n - 1 - q of the source.  (tano, tata) -> (anno, data); the two may be the same arrays, so a stretch is flipped by
   exchanging its two ends' partners, the middle entry of an odd stretch with itself. */
static void mirror_track(const HITS_DB *block, const int64 *tano, const int *tata, int64 *anno, int *data)
{ const int nreads = block->nreads;
  int r;
  for (r = 0; r <= nreads; r++)
    anno[r] = tano[r];
  for (r = 0; r < nreads; r++)
    { const int   rlen = block->reads[r].rlen;
      const int64 first = tano[r], n = tano[r + 1] - first;
      int64 q;
      for (q = 0; 2 * q < n; q++)
        { const int64 lo = first + q, hi = first + (n - 1 - q);
          const int   plo = tata[lo], phi = tata[hi];
          data[lo] = rlen - phi;
          data[hi] = rlen - plo;
        }
    }
}

/* Reverse-complemented copy of a loaded block into *out (daligner.c:511-628 on a copy): own bases and own
   mask tracks, everything else (reads, path) shared with `block`, which must outlive it.  Re-entrant: the
   command-line driver prepares the next block on a second thread.  Release with damar_free_complement. */
void damar_complement_copy(const HITS_DB *block, HITS_DB *out)
{ int64 n = block->reads[block->nreads].boff;
  char *seq = NULL;
  const HITS_TRACK *src;
  float x;
  int   i;

  *out = *block;
  out->tracks = NULL;
  x = out->freq[0]; out->freq[0] = out->freq[3]; out->freq[3] = x;
  x = out->freq[1]; out->freq[1] = out->freq[2]; out->freq[2] = x;
  if (block->bases != NULL)                        /* (a packed block has none: the GPU complements it, damar_block_upload_packed) */
    { seq = (char *) big_alloc((size_t) n + 1 + 64, "complement block");
      *seq++ = 4;
      out->bases = (void *) seq;
      for (i = 0; i < block->nreads; i++)         /* every read reversed and complemented straight out of the forward block */
        { const int64 bo = block->reads[i].boff;
          rc_copy(seq + bo, ((const char *) block->bases) + bo, block->reads[i].rlen);
          seq[bo + block->reads[i].rlen] = 4;
        }
    }
  for (src = block->tracks; src != NULL; src = src->next)
    { const int64 *tano = (const int64 *) src->anno;
      HITS_TRACK  *trg = (HITS_TRACK *) xmalloc(sizeof(HITS_TRACK), "mask header");
      trg->name = strdup(src->name);
      trg->size = 4;
      trg->data = xmalloc(sizeof(int) * (size_t) (tano[block->nreads] + 1), "mask data");
      trg->anno = xmalloc(sizeof(int64) * (size_t) (block->nreads + 1), "mask index");
      trg->next = out->tracks;
      out->tracks = trg;
      mirror_track(block, tano, (const int *) src->data, (int64 *) trg->anno, (int *) trg->data);
    }
}

void damar_free_complement(HITS_DB *c)
{ if (c->bases != NULL)
    free(((char *) c->bases) - 1);
  free_tracks(c->tracks);
  c->bases = NULL;
  c->tracks = NULL;
}

HITS_DB *damar_complement_block(HITS_DB *block, int inplace)
{ static HITS_DB cstore;
  static int     have = 0;
  HITS_TRACK *t;
  float x;
  int   i;

  if (!inplace)
    { /* the copy is a static record (like the reference's static cblock, daligner.c:529): its mask tracks
         are released by the next call, its bases belong to the caller */
      if (have)
        free_tracks(cstore.tracks);
      damar_complement_copy(block, &cstore);
      have = 1;
      return &cstore;
    }
  x = block->freq[0]; block->freq[0] = block->freq[3]; block->freq[3] = x;
  x = block->freq[1]; block->freq[1] = block->freq[2]; block->freq[2] = x;
  for (i = 0; i < block->nreads; i++)
    rc_inplace((char *) block->bases + block->reads[i].boff, block->reads[i].rlen);
  for (t = block->tracks; t != NULL; t = t->next)
    mirror_track(block, (const int64 *) t->anno, (const int *) t->data, (int64 *) t->anno, (int *) t->data);
  return block;
}

/* daligner.c:442-497 read_DB's mask part: load every named interval track of the block
 * (db/DB.c:1113 Load_Track: <path>.<track>.anno = int tracklen, int size, size-byte offsets;
 * <path>.<track>.data = int pairs [beg,end) per read; per-block files <path>.<part>.<track>.*
 * take precedence) and leave ONE track on block->tracks: anno = int64[nreads+1] in units of
 * ints, data = the union of the intervals (daligner.c:263-439 Merge_Size / Merge_Tracks; the
 * union is formed by sorting: overlapping or abutting intervals merge, which selects the same
 * k-mers as the reference's heap sweep whatever order it meets equal coordinates in).
 * Returns 0, or -1 after printing what is wrong.  lib/tracks.c's compressed .a2/.d2 form is
 * tried first, as track_load does. */
/* lib/tracks.c:20-131 track_load, compressed form: <path>.<track>.a2 = 64-byte header {u16 version,
 * u16 size, u32 pad, u64 len, u64 clen, u64 cdlen, 4 reserved u64} + clen bytes, <path>.<track>.d2 =
 * cdlen bytes; both payloads are runs of {u64 n, n bytes of zlib stream}, each stream inflating to
 * at most 8 MiB (lib/compression.c).  anno = u64 byte offsets for ALL reads of the database; the
 * block's slice is cut out here.  Returns 1 if loaded, 0 if there is no .a2 file, -1 on error. */
static int inflate_chunks(const unsigned char *in, uint64 ilen, unsigned char *out, uint64 olen)
{ uint64 ip = 0, op = 0;
  while (ip < ilen)
    { uint64 clen;
      uLongf dlen;
      if (ip + 8 > ilen)
        return -1;
      memcpy(&clen, in + ip, 8);
      ip += 8;
      if (ip + clen > ilen)
        return -1;
      dlen = (uLongf) (olen - op);
      if (uncompress(out + op, &dlen, in + ip, (uLong) clen) != Z_OK)
        return -1;
      ip += clen;
      op += dlen;
    }
  return 0;
}

static int load_a2(const HITS_DB *block, const char *name, int64 **offs_out, int **data_out)
{ char   path[4400];
  FILE  *af, *df;
  struct { uint16 version, size; uint32 pad; uint64 len, clen, cdlen, r1, r2, r3, r4; } h;
  unsigned char *cbuf = NULL, *dbuf = NULL;
  uint64 *anno = NULL;
  int64  *offs;
  int    *data;
  int     i, nreads = block->nreads, rc = -1;

  snprintf(path, sizeof(path), "%s.%s.a2", block->path, name);
  if ((af = fopen(path, "r")) == NULL)
    return 0;
  snprintf(path, sizeof(path), "%s.%s.d2", block->path, name);
  df = fopen(path, "r");
  if (fread(&h, sizeof(h), 1, af) != 1 || h.size != 8 || df == NULL)
    { fprintf(stderr, "damar: could not read header / data of track %s\n", name);
      goto done;
    }
  if ((block->part == 0 && h.len != (uint64) nreads) || h.len < (uint64) (block->ufirst + nreads))
    { fprintf(stderr, "damar: invalid track length in header of track %s\n", name);
      goto done;
    }
  cbuf = (unsigned char *) xmalloc((size_t) h.clen + 8, "track");
  anno = (uint64 *) xmalloc(8 * (size_t) (h.len + 1), "track");
  if ((h.clen > 0 && fread(cbuf, (size_t) h.clen, 1, af) != 1) ||
      inflate_chunks(cbuf, h.clen, (unsigned char *) anno, 8 * (h.len + 1)))
    { fprintf(stderr, "damar: failed to read anno track %s\n", name);
      goto done;
    }
  free(cbuf);
  cbuf = (unsigned char *) xmalloc((size_t) h.cdlen + 8, "track");
  dbuf = (unsigned char *) xmalloc((size_t) anno[h.len] + 8, "track");
  if ((h.cdlen > 0 && fread(cbuf, (size_t) h.cdlen, 1, df) != 1) ||
      inflate_chunks(cbuf, h.cdlen, dbuf, anno[h.len]))
    { fprintf(stderr, "damar: failed to read data track %s\n", name);
      goto done;
    }
  { const uint64 o0 = anno[block->ufirst], dlen = anno[block->ufirst + nreads] - o0;
    offs = (int64 *) xmalloc(sizeof(int64) * (size_t) (nreads + 1), "mask index");
    data = (int *) xmalloc((size_t) dlen + 8, "mask data");
    for (i = 0; i <= nreads; i++)
      offs[i] = (int64) ((anno[block->ufirst + i] - o0) / sizeof(int));
    memcpy(data, dbuf + o0, (size_t) dlen);
    *offs_out = offs;
    *data_out = data;
  }
  rc = 1;
done:
  free(cbuf);  free(dbuf);  free(anno);
  fclose(af);
  if (df) fclose(df);
  return rc;
}

typedef struct { int beg, end; } Ival;

static int ival_cmp(const void *x, const void *y)
{ const Ival *a = (const Ival *) x, *b = (const Ival *) y;
  if (a->beg != b->beg) return (a->beg < b->beg) ? -1 : 1;
  if (a->end != b->end) return (a->end < b->end) ? -1 : 1;
  return 0;
}

int damar_load_masks(HITS_DB *block, char **names, int n)
{ int64 **offs;
  int   **dats;
  int     t, i, nreads = block->nreads, rc = -1;
  char    path[4400];

  if (n <= 0)
    return 0;
  offs = (int64 **) calloc((size_t) n, sizeof(int64 *));
  dats = (int **) calloc((size_t) n, sizeof(int *));
  for (t = 0; t < n; t++)
    { FILE *af = NULL, *df = NULL;
      int   tracklen, size, ispart = 0, ureads;
      { int got = load_a2(block, names[t], &offs[t], &dats[t]);       /* lib/tracks.c tries this form first */
        if (got < 0)
          goto done;
        if (got > 0)
          continue;
      }
      if (block->part > 0)
        { snprintf(path, sizeof(path), "%s.%d.%s.anno", block->path, block->part, names[t]);
          af = fopen(path, "r");
          ispart = (af != NULL);
        }
      if (af == NULL)
        { snprintf(path, sizeof(path), "%s.%s.anno", block->path, names[t]);
          af = fopen(path, "r");
        }
      if (af == NULL)
        { fprintf(stderr, "damar: Track '%s' does not exist\n", names[t]);
          goto done;
        }
      if (ispart)
        snprintf(path, sizeof(path), "%s.%d.%s.data", block->path, block->part, names[t]);
      else
        snprintf(path, sizeof(path), "%s.%s.data", block->path, names[t]);
      df = fopen(path, "r");
      if (fread(&tracklen, sizeof(int), 1, af) != 1 || fread(&size, sizeof(int), 1, af) != 1 ||
          (size != 4 && size != 8) || df == NULL)
        { fprintf(stderr, "damar: Track '%s' annotation file is junk\n", names[t]);
          fclose(af);  if (df) fclose(df);
          goto done;
        }
      ureads = ispart ? nreads : block->ureads;
      if (tracklen != ureads)
        { fprintf(stderr, "damar: Track '%s' not same size as database (track: %d, db: %d)!\n",
                  names[t], tracklen, ureads);
          fclose(af);  fclose(df);
          goto done;
        }
      if (!ispart && block->part > 0)
        fseeko(af, (off_t) size * block->ufirst, SEEK_CUR);
      offs[t] = (int64 *) xmalloc(sizeof(int64) * (size_t) (nreads + 1), "mask index");
      for (i = 0; i <= nreads; i++)
        { int64 v = 0;
          int   v4;
          if ((size == 8 ? fread(&v, 8, 1, af) : fread(&v4, 4, 1, af)) != 1)
            { fprintf(stderr, "damar: Track '%s' annotation file is junk\n", names[t]);
              fclose(af);  fclose(df);
              goto done;
            }
          offs[t][i] = (size == 8) ? v : (int64) v4;
        }
      { int64 o0 = offs[t][0], dlen = offs[t][nreads] - offs[t][0];
        dats[t] = (int *) xmalloc((size_t) dlen + 8, "mask data");
        fseeko(df, (off_t) o0, SEEK_SET);
        if (dlen > 0 && fread(dats[t], (size_t) dlen, 1, df) != 1)
          { fprintf(stderr, "damar: Track '%s' data file size mismatch. Expected %lld\n", names[t], (long long) dlen);
            fclose(af);  fclose(df);
            goto done;
          }
        for (i = 0; i <= nreads; i++)
          offs[t][i] = (offs[t][i] - o0) / (int64) sizeof(int);     /* daligner.c:475-477: units of ints */
      }
      fclose(af);
      fclose(df);
    }

  { int64  total = 0, top = 0;
    int64 *anno;
    int   *data;
    Ival  *iv;
    int    cap = 0;
    HITS_TRACK *trk;
    for (t = 0; t < n; t++)
      total += offs[t][nreads];
    anno = (int64 *) xmalloc(sizeof(int64) * (size_t) (nreads + 1), "mask index");
    data = (int *) xmalloc(sizeof(int) * (size_t) (total + 2), "mask data");
    iv = NULL;
    for (i = 0; i < nreads; i++)
      { int m = 0, q;
        anno[i] = top;
        for (t = 0; t < n; t++)
          m += (int) ((offs[t][i + 1] - offs[t][i]) / 2);
        if (m > cap)
          { cap = m + 64;
            iv = (Ival *) realloc(iv, sizeof(Ival) * (size_t) cap);
          }
        m = 0;
        for (t = 0; t < n; t++)
          { int64 a;
            for (a = offs[t][i]; a + 1 < offs[t][i + 1]; a += 2)
              { iv[m].beg = dats[t][a];  iv[m].end = dats[t][a + 1];  m += 1; }
          }
        if (n > 1)
          qsort(iv, (size_t) m, sizeof(Ival), ival_cmp);
        for (q = 0; q < m; q++)
          { if (q > 0 && top > anno[i] && iv[q].beg <= data[top - 1])
              { if (iv[q].end > data[top - 1])
                  data[top - 1] = iv[q].end;
              }
            else
              { data[top++] = iv[q].beg;
                data[top++] = iv[q].end;
              }
          }
      }
    anno[nreads] = top;
    free(iv);
    trk = (HITS_TRACK *) xmalloc(sizeof(HITS_TRACK), "mask header");
    trk->name = strdup(n > 1 ? "merge" : names[0]);
    trk->size = 8;
    trk->anno = (void *) anno;
    trk->data = (void *) data;
    trk->next = NULL;
    free_tracks(block->tracks);
    block->tracks = trk;
  }
  rc = 0;
done:
  for (t = 0; t < n; t++)
    { free(offs[t]);
      free(dats[t]);
    }
  free(offs);
  free(dats);
  return rc;
}

/****************************************************************************************
 *  Synthetic reads: db/simulator.c semantics, written straight into DB files
 ****************************************************************************************/

void damar_sim_defaults(damar_sim_params *p)
{ p->genome_mbp = 1.0;
  p->coverage   = 20.;
  p->bias       = .5;
  p->seed       = 1;
  p->rmean      = 10000;
  p->rsdev      = 2000;
  p->rshort     = 4000;
  p->erate      = .15;
  p->block_mbp  = 200;
  p->min_len    = 1000;
  p->tandem_frac = 0.;
  p->max_blocks = 0;
}

/* glibc's srand48/drand48 restated inline (X' = 0x5DEECE66D X + 0xB mod 2^48, value X / 2^48: the
   library builds the same double from the 48 bits): the stream is the one db/simulator.c draws from,
   at a third of the cost of the library call -- config 4 draws 5e10 numbers */
static uint64_t sim_x48;
static inline void   sim_srand48(long seed) { sim_x48 = (((uint64_t) (uint32_t) seed) << 16) | 0x330Eull; }
static inline double sim_drand48(void)
{ sim_x48 = (sim_x48 * 0x5DEECE66Dull + 0xBull) & 0xFFFFFFFFFFFFull;
  return (double) sim_x48 * (1.0 / 281474976710656.0);
}
#define srand48 sim_srand48
#define drand48 sim_drand48

/* second, independent stream for the tandem implants (-T): xorshift64* */
static uint64_t sim_tx;
static inline uint64_t tan_next(void)
{ sim_tx ^= sim_tx >> 12;  sim_tx ^= sim_tx << 25;  sim_tx ^= sim_tx >> 27;
  return sim_tx * 0x2545F4914F6CDD1Dull;
}
static inline int tan_range(int lo, int hi)       /* uniform in [lo, hi] */
{ return lo + (int) (tan_next() % (uint64_t) (hi - lo + 1)); }

/* plain simulator reads hold no tandem repeats, so datander finds nothing in them.
   Overwrite a stretch of the read with `copies` copies of a random unit of 50-500 bp, each copy
   with 3 % substitutions (the recipe of SURVEY App. E, applied per read). */
static void tandem_implant(char *seq, int len)
{ int unit = tan_range(50, 500), copies = tan_range(5, 40), span, pos, c, i;
  char u[500];
  if (unit * copies > len / 2)
    copies = (len / 2) / unit;
  if (copies < 3)
    return;
  span = unit * copies;
  pos  = tan_range(0, len - span);
  for (i = 0; i < unit; i++)
    u[i] = (char) (tan_next() & 3);
  for (c = 0; c < copies; c++)
    for (i = 0; i < unit; i++)
      { char b = u[i];
        if (tan_next() % 100 < 3)
          b = (char) ((b + 1 + (int) (tan_next() % 3)) & 3);
        seq[pos + c * unit + i] = b;
      }
}

#define NORM_STEPS 60000
#define NORM_MAX   6.0

typedef struct
{ double cdf[NORM_STEPS + 1];   /* upper half of the N(0,1) cdf, simulator.c:153-182 */
  double step;
} NormTable;

static void norm_init(NormTable *t)
{ double sum = 0., del = NORM_MAX / NORM_STEPS;
  int    i;
  t->step = del;
  for (i = 0; i < NORM_STEPS; i++)
    { double x = i * del;
      t->cdf[i] = sum;
      sum += exp(-.5 * x * x) * del;
    }
  t->cdf[NORM_STEPS] = sum;
  sum *= 2.;
  for (i = 0; i < NORM_STEPS; i++)
    t->cdf[i] /= sum;
  t->cdf[NORM_STEPS] = 1.;
}

/* simulator.c:184-227 */
static double norm_sample(const NormTable *t, double x)
{ double y = (x >= .5) ? x - .5 : .5 - x;
  int    l = 0, r = NORM_STEPS;
  while (l < r)
    { int m = (l + r) >> 1;
      if (y < t->cdf[m])
        r = m;
      else
        l = m + 1;
    }
  y = (r - (t->cdf[r] - y) / (t->cdf[r] - t->cdf[r - 1])) * t->step;
  return (x < .5) ? -y : y;
}

typedef struct
{ FILE  *bps, *idx;
  int64  off;
  int64  totlen;
  int64  count[4];
  int    maxlen;
  int    nreads;
  HITS_READ *recs;
  int    rmax;
} DbOut;

static void dbout_add(DbOut *o, const char *seq, int len)
{ int   clen = (len + 3) >> 2, i;
  unsigned char *buf = (unsigned char *) xmalloc((size_t) clen + 4, "pack");
  HITS_READ hr;

  memset(buf, 0, (size_t) clen + 4);
  for (i = 0; i < len; i++)
    { o->count[(int) seq[i]] += 1;
      buf[i >> 2] |= (unsigned char) (seq[i] << (6 - 2 * (i & 3)));
    }
  fwrite(buf, 1, (size_t) clen, o->bps);
  free(buf);

  memset(&hr, 0, sizeof(hr));
  hr.rlen  = len;
  hr.boff  = o->off;
  hr.coff  = -1;
  hr.flags = DB_BEST;
  fwrite(&hr, sizeof(hr), 1, o->idx);
  if (o->nreads >= o->rmax)
    { o->rmax = (int) (1.5 * o->rmax) + 1024;
      o->recs = (HITS_READ *) realloc(o->recs, sizeof(HITS_READ) * (size_t) o->rmax);
    }
  o->recs[o->nreads] = hr;
  o->off    += clen;
  o->totlen += len;
  if (len > o->maxlen)
    o->maxlen = len;
  o->nreads += 1;
}

int damar_sim_write_db(const damar_sim_params *p, const char *dir, const char *root)
{ int     genome = (int) (p->genome_mbp * 1000000.);
  char   *src;
  char    path[4096];
  DbOut   out;
  HITS_DB db;
  NormTable *nt;
  double  nmean, nsdev;
  int64   want, have;
  char   *rbuf = NULL;
  int     rcap = 0;
  int     i, nblocks, blk_done = 0;
  int64   blk_tot = 0;
  FILE   *stub;

  mkdir(dir, 0755);
  memset(&out, 0, sizeof(out));
  snprintf(path, sizeof(path), "%s/.%s.bps", dir, root);
  out.bps = fopen(path, "w");
  snprintf(path, sizeof(path), "%s/.%s.idx", dir, root);
  out.idx = fopen(path, "w");
  if (out.bps == NULL || out.idx == NULL)
    { fprintf(stderr, "damar: cannot create DB files in %s\n", dir);
      return -1;
    }
  memset(&db, 0, sizeof(db));
  fwrite(&db, sizeof(db), 1, out.idx);

  /* simulator.c:100-129 random_genome */
  { double pra = p->bias / 2., prc = (1. - p->bias) / 2. + pra, prg = (1. - p->bias) / 2. + prc;
    src = (char *) xmalloc((size_t) genome + 1, "genome");
    srand48(p->seed);
    sim_tx = 0x9E3779B97F4A7C15ull ^ ((uint64_t) (uint32_t) p->seed * 0xD1B54A32D192ED03ull);
    for (i = 0; i < genome; i++)
      { double x = drand48();
        src[i] = (char) ((x < pra) ? 0 : (x < prc) ? 1 : (x < prg) ? 2 : 3);
      }
    src[genome] = 4;
  }

  /* simulator.c:242-352 shotgun */
  nsdev = (1. * p->rsdev) / p->rmean;
  nsdev = log(1. + nsdev * nsdev);
  nmean = log(1. * p->rmean) - .5 * nsdev;
  nsdev = sqrt(nsdev);
  if (genome < p->rshort)
    { fprintf(stderr, "damar: genome shorter than the shortest read\n");
      return -1;
    }
  nt = (NormTable *) xmalloc(sizeof(NormTable), "normal table");
  norm_init(nt);

  want = (int64) (p->coverage * genome);
  have = 0;
  while (have < want)
    { int   len, sdl, ins, del, elen, j;
      char *s, *t;

      len = (int) exp(nmean + nsdev * norm_sample(nt, drand48()));
      if (len > genome)
        len = genome;
      if (len < p->rshort)
        continue;

      sdl = (int) (len * p->erate);
      ins = del = 0;
      for (j = 0; j < sdl; j++)
        { double x = drand48();
          if (x < .73333)
            ins += 1;
          else if (x < .93333)
            del += 1;
        }
      sdl -= ins;
      elen = len + (ins - del);
      s = src + (int) (drand48() * ((genome - len) + .9999999));

      if (elen > rcap)
        { rcap = ((int) (1.2 * elen)) + 1000;
          rbuf = (char *) realloc(rbuf, (size_t) rcap + 3);
        }
      t = rbuf;
      while ((len + 1) * drand48() < ins)
        { *t++ = (char) (4. * drand48());
          ins -= 1;
        }
      for (; len > 0; len--)
        { if (len * drand48() >= sdl)
            *t++ = *s;
          else if (sdl * drand48() >= del)
            { double x = 3. * drand48();
              if (x >= *s)
                x += 1.;
              *t++ = (char) x;
              sdl -= 1;
            }
          else
            { del -= 1;
              sdl -= 1;
            }
          s += 1;
          while (len * drand48() < ins)
            { *t++ = (char) (4. * drand48());
              ins -= 1;
            }
        }
      *t = 4;

      if (drand48() >= .5)             /* strand flip; note simulator.c:133-145 uses s <= t */
        { char *a = rbuf, *b = rbuf + (elen - 1);
          while (a <= b)
            { char c = *a;
              *a++ = (char) (3 - *b);
              *b-- = (char) (3 - c);
            }
        }

      if (p->tandem_frac > 0. && (double) (tan_next() >> 11) * (1.0 / 9007199254740992.0) < p->tandem_frac)
        tandem_implant(rbuf, elen);

      if (elen >= p->min_len)          /* FA2db -x */
        { dbout_add(&out, rbuf, elen);
          if (p->max_blocks > 0)       /* stop once the first max_blocks blocks are complete (they do not
                                          depend on what would follow: the generator is sequential) */
            { blk_tot += elen;
              if (blk_tot >= p->block_mbp * 1000000ll)
                { blk_tot = 0;
                  if (++blk_done >= p->max_blocks)
                    break;
                }
            }
        }
      have += elen;
    }
  free(rbuf);
  free(nt);
  free(src);

  db.ureads = out.nreads;
  for (i = 0; i < 4; i++)
    db.freq[i] = (float) ((1. * out.count[i]) / out.totlen);
  db.totlen = out.totlen;
  db.maxlen = out.maxlen;
  db.nreads = 0;   /* FA2db leaves the block fields zero; Open_DB recomputes them */
  rewind(out.idx);
  fwrite(&db, sizeof(db), 1, out.idx);
  fclose(out.idx);
  fclose(out.bps);

  /* stub + DBsplit partition (db/DBsplit.c:201-234) */
  snprintf(path, sizeof(path), "%s/%s.db", dir, root);
  if ((stub = fopen(path, "w")) == NULL)
    return -1;
  fprintf(stub, "files = %9d\n", 1);
  fprintf(stub, "  %9d %s %s\n", out.nreads, "sim", "Sim");
  { int64 size = p->block_mbp * 1000000ll, tot = 0;
    int   open = 0;
    long  pos  = ftell(stub);

    nblocks = 0;
    fprintf(stub, "blocks = %9d\n", 0);
    fprintf(stub, "size = %9lld\n", (long long) p->block_mbp);
    fprintf(stub, " %9d\n", 0);
    for (i = 0; i < out.nreads; i++)
      { open += 1;
        tot  += out.recs[i].rlen;
        if (tot >= size)
          { fprintf(stub, " %9d\n", i + 1);
            tot = 0;
            open = 0;
            nblocks += 1;
          }
      }
    if (open > 0)
      { fprintf(stub, " %9d\n", out.nreads);
        nblocks += 1;
      }
    fseek(stub, pos, SEEK_SET);
    fprintf(stub, "blocks = %9d\n", nblocks);
  }
  fclose(stub);
  free(out.recs);
  return nblocks;
}

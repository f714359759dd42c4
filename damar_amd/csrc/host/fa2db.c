/* fa2db.c -- FASTA -> database, the creation path of the reference's db/FA2db.c: writes
 * <path>.db (stub), .<path>.idx (HITS_DB header + one HITS_READ per read), .<path>.bps (2-bit bases) and the two
 * tracks every FA2db run leaves, .<path>.seqID.{anno,data} (index of the read inside its FASTA file) and
 * .<path>.pacbio.{anno,data} (well, begin, end of reads with a PacBio header), byte for byte as the reference
 * does for the same input (FA2db.c:611-651 addReadToDB, 652-905 readFastaFile, 1063-1130 main; FA2x.c:63-94,
 * 154-300 tracks; fileUtils.c:8-46 isPacBioHeader) -- except for the fields of the .idx header that the
 * reference leaves uninitialised, which are zero here.
 *
 *     FA2db [-v] [-a] [-b] [-Q] [-c<track>]... [-x<int(1000)>] <path:db> (-f<file of fasta names> | <input:fasta> ...)
 *
 * Built: creating a database from .fasta / .fa files and adding files to an existing one (its block
 * partition is extended, FA2db.c:908-975; -a starts a new block), -b (longest read of a well), -x, -f, -c (header
 * arguments NAME=v1,v2,... of the reads become tracks .<path>.NAME.{anno,data}: FA2db.c:169-355 parse_header, :638-643)
 * and -Q (only reads whose readType argument -- named with -c -- says FullHqRead: FA2db.c:809-837).  Host code, C, no GPU.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <strings.h>
#include <dirent.h>

#include "damar_db.h"

#define MAX_NAME 204800                    /* db/DB.h:398 (200*1024) */
#define STR_(x) #x
#define STR(x) STR_(x)

typedef struct
{ char  *name;
  uint64 *anno;   int64 amax;               /* bytes of data per read, then offsets */
  int    *data;   int64 dtop, dmax;
} Track;

static Track T_seq = { "seqID", NULL, 0, NULL, 0, 0 }, T_pac = { "pacbio", NULL, 0, NULL, 0, 0 };
static int   T_pac_used = 0;

/* -c: tracks made of header arguments, created when a value (or, adding to a database, an .anno file) first names them */
static Track **T_hdr = NULL;
static int     T_nhdr = 0;
static char  **C_name = NULL;                 /* the names given with -c */
static int     C_n = 0;

static Track *hdr_track(const char *name)
{ int i;
  for (i = 0; i < T_nhdr; i++)
    if (strcmp(T_hdr[i]->name, name) == 0)
      return T_hdr[i];
  T_hdr = (Track **) realloc(T_hdr, sizeof(Track *) * (size_t) (T_nhdr + 1));
  T_hdr[T_nhdr] = (Track *) calloc(1, sizeof(Track));
  T_hdr[T_nhdr]->name = strdup(name);
  return T_hdr[T_nhdr++];
}

/* the header arguments of one read that -c asked for (FA2db.c:197-355) */
typedef struct { char *name;  int nval, vmax;  int *val; } HdrArg;
typedef struct { HdrArg *arg;  int n, max; } HdrArgs;

static void hdr_arg_value(HdrArg *a, int v)
{ if (a->nval >= a->vmax)
    { a->vmax = a->vmax * 2 + 16;
      a->val = (int *) realloc(a->val, sizeof(int) * (size_t) a->vmax);
    }
  a->val[a->nval++] = v;
}

/* `text` = what follows the read name in a header line: arguments NAME=v[,v...] separated by single blanks.  Arguments
   whose name was not given with -c are skipped; a name must be letters, digits and '_' up to the '='; values are decimal
   integers ("RQ=0.851" reads as 851; the characters of a "chemistry" argument are its values, up to the first comma). */
static void parse_header_args(const char *text, HdrArgs *h)
{ const char *c = text;
  h->n = 0;
  while (*c != '\0' && *c != '\n')
    { const char *name = c;
      size_t nlen;
      int i, wanted = 0;
      while ((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || *c == '_')
        c++;
      if (*c != '=')
        { fprintf(stderr, "malformed track name: '%s'\n", c);
          exit(1);
        }
      nlen = (size_t) (c - name);
      for (i = 0; i < C_n; i++)
        if (strlen(C_name[i]) == nlen && strncmp(C_name[i], name, nlen) == 0)
          wanted = 1;
      if (!wanted)
        { c++;
          while (*c != '\0' && *c != '\n' && *c != ' ')
            c++;
          if (*c != ' ')
            break;
          c++;
          continue;
        }
      if (h->n >= h->max)
        { const int old = h->max;
          h->max = h->max * 2 + 8;
          h->arg = (HdrArg *) realloc(h->arg, sizeof(HdrArg) * (size_t) h->max);
          memset(h->arg + old, 0, sizeof(HdrArg) * (size_t) (h->max - old));
        }
      { HdrArg *a = h->arg + h->n++;
        int more, blank;
        free(a->name);
        a->name = strndup(name, nlen);
        a->nval = 0;
        do
          { const char *v;
            c++;                                    /* past '=' or ',' */
            if (strcmp(a->name, "RQ") == 0 && strncasecmp(c, "0.", 2) == 0)
              c += 2;
            v = c;
            while (*c != '\0' && *c != '\n' && *c != ' ' && *c != ',')
              c++;
            more = (*c == ',');
            blank = (*c == ' ');
            if (strcmp(a->name, "chemistry") == 0)
              { const char *q;
                for (q = v; q < c; q++)
                  hdr_arg_value(a, (int) *q);
                if (more)                             /* (what follows the first comma goes with the rest of the line) */
                  return;
                break;
              }
            { char  num[64], *end;
              size_t l = (size_t) (c - v);
              long   val;
              if (l >= sizeof(num)) l = sizeof(num) - 1;
              memcpy(num, v, l);
              num[l] = '\0';
              val = strtol(num, &end, 10);
              if (*end != '\0')
                { printf("non-numeric value %s\n", num);
                  exit(1);
                }
              hdr_arg_value(a, (int) val);
            }
          }
        while (more);
        if (!blank)
          break;
        c++;
      }
    }
}

static void track_add(Track *t, int64 read, int value)                   /* FA2x.c:63-94 */
{ if (read >= t->amax)
    { int64 n = (int64) (read * 1.2 + 1000);
      t->anno = (uint64 *) realloc(t->anno, sizeof(uint64) * (size_t) n);
      memset(t->anno + t->amax, 0, sizeof(uint64) * (size_t) (n - t->amax));
      t->amax = n;
    }
  if (t->dtop >= t->dmax)
    { t->dmax = (int64) (t->dtop * 1.2 + 1000);
      t->data = (int *) realloc(t->data, sizeof(int) * (size_t) t->dmax);
    }
  t->anno[read] += (strcmp(t->name, "chemistry") == 0) ? sizeof(char) : sizeof(int);      /* FA2x.c:83-90: characters */
  t->data[t->dtop++] = value;
}

/* FA2x.c:96-300: the reads up to `first` belong to an existing database.  Their offsets stay in the track's
   .anno file if it is there and consistent (its last offset is where the new data starts); the offsets of the
   new reads are written over that last entry, the data are appended. */
static void track_write(Track *t, const char *dir, const char *root, int first, int ureads)
{ char   path[2 * MAX_NAME + 64];
  FILE  *f;
  uint64 off = 0;
  int    i, tlen = ureads, tsize = (int) sizeof(uint64), exists = 0;

  if (ureads + 1 > t->amax)
    { t->anno = (uint64 *) realloc(t->anno, sizeof(uint64) * (size_t) (ureads + 1));
      memset(t->anno + t->amax, 0, sizeof(uint64) * (size_t) (ureads + 1 - t->amax));
      t->amax = ureads + 1;
    }
  snprintf(path, sizeof(path), "%s/.%s.%s.anno", dir, root, t->name);
  if ((f = fopen(path, "r")) != NULL)
    { int    olen, osize;
      uint64 last;
      exists = 1;
      if (fread(&olen, sizeof(int), 1, f) == 1 && olen == first && fread(&osize, sizeof(int), 1, f) == 1 &&
          fseek(f, (long) (sizeof(uint64) * (size_t) first), SEEK_CUR) == 0 && fread(&last, sizeof(uint64), 1, f) == 1)
        off = last;
      fclose(f);
    }
  for (i = first; i <= ureads; i++)
    { uint64 c = t->anno[i];
      t->anno[i] = off;
      off += c;
    }
  if (exists)
    { if ((f = fopen(path, "r+")) == NULL)
        { fprintf(stderr, "[ERROR] - Cannot open file %s for appending track %s\n", path, t->name);
          exit(1);
        }
      fwrite(&tlen, sizeof(int), 1, f);
      fwrite(&tsize, sizeof(int), 1, f);
      fflush(f);
      fseeko(f, -(off_t) sizeof(uint64), SEEK_END);
      fwrite(t->anno + first, sizeof(uint64), (size_t) (tlen + 1 - first), f);
    }
  else
    { if ((f = fopen(path, "w")) == NULL)
        { fprintf(stderr, "[WARNING] Cannot create file %s. Skip track %s.\n", path, t->name);
          return;
        }
      fwrite(&tlen, sizeof(int), 1, f);
      fwrite(&tsize, sizeof(int), 1, f);
      fwrite(t->anno, sizeof(uint64), (size_t) ureads + 1, f);
    }
  fclose(f);
  snprintf(path, sizeof(path), "%s/.%s.%s.data", dir, root, t->name);
  if ((f = fopen(path, "a")) == NULL)
    return;
  if (strcmp(t->name, "chemistry") == 0)              /* FA2x.c:272-281: its values go out as characters */
    { int64 v;
      for (v = 0; v < t->dtop; v++)
        fputc((char) t->data[v], f);
    }
  else
    fwrite(t->data, sizeof(int), (size_t) t->dtop, f);
  fclose(f);
}

static int is_pacbio_header(const char *h)                               /* fileUtils.c:8-46 */
{ const char *end = strchr(h, ' '), *p;
  int n = 0;
  if (end == NULL)
    end = h + strlen(h) - 1;
  for (p = strchr(h, '/'); p != NULL && p < end; p = strchr(p + 1, '/'))
    n += 1;
  return n == 2;
}

static char **F_seen = NULL;
static int    F_nseen = 0;

static int file_seen(const char *core)             /* remembers it if it is new */
{ int j;
  for (j = 0; j < F_nseen; j++)
    if (strcmp(F_seen[j], core) == 0)
      return 1;
  F_seen = (char **) realloc(F_seen, sizeof(char *) * (size_t) (F_nseen + 1));
  F_seen[F_nseen++] = strdup(core);
  return 0;
}

static const char *base_name(const char *path)
{ const char *s = strrchr(path, '/');
  return s ? s + 1 : path;
}

typedef struct
{ FILE *idx, *bps, *stub;
  int64 off, totlen, count[4];
  int   ureads, maxlen, minlen, verbose, nadded, best, fullhq;
} Out;

static void add_read(Out *o, char *seq, int len, int seqid, int pac, int well, int beg, int end, const HdrArgs *ha)   /* FA2db.c:611-651 */
{ static unsigned char *buf = NULL;
  static int bmax = 0;
  int clen = (len + 3) >> 2, i;
  HITS_READ hr;

  if (clen + 4 > bmax)
    { bmax = 2 * clen + 1024;
      buf = (unsigned char *) realloc(buf, (size_t) bmax);
    }
  memset(buf, 0, (size_t) clen + 4);
  for (i = 0; i < len; i++)
    { int x;
      switch (seq[i])
      { case 'c': case 'C': x = 1; break;
        case 'g': case 'G': x = 2; break;
        case 't': case 'T': x = 3; break;
        default:            x = 0;                  /* FA2db.c:84-87: everything else counts as 'a' */
      }
      o->count[x] += 1;
      buf[i >> 2] |= (unsigned char) (x << (6 - 2 * (i & 3)));
    }
  memset(&hr, 0, sizeof(hr));
  hr.boff  = o->off;
  hr.rlen  = len;
  hr.coff  = -1;
  hr.flags = DB_BEST;
  fwrite(buf, 1, (size_t) clen, o->bps);
  fwrite(&hr, sizeof(hr), 1, o->idx);
  track_add(&T_seq, o->ureads, seqid);
  if (pac)
    { track_add(&T_pac, o->ureads, well);
      track_add(&T_pac, o->ureads, beg);
      track_add(&T_pac, o->ureads, end);
      T_pac_used = 1;
    }
  if (ha != NULL)
    { int a, v;
      for (a = 0; a < ha->n; a++)
        for (v = 0; v < ha->arg[a].nval; v++)
          track_add(hdr_track(ha->arg[a].name), o->ureads, ha->arg[a].val[v]);
    }
  o->off    += clen;
  o->ureads += 1;
  o->totlen += len;
  if (len > o->maxlen)
    o->maxlen = len;
}

/* -Q: the read carries exactly one readType value and it is FullHqRead0 / FullHqRead1 (1, 2) */
static int full_hq_read(const HdrArgs *h)
{ int i;
  for (i = 0; i < h->n; i++)
    if (strcmp(h->arg[i].name, "readType") == 0)
      return h->arg[i].nval == 1 && (h->arg[i].val[0] == 1 || h->arg[i].val[0] == 2);
  return 0;
}

static void read_fasta(Out *o, const char *name)                          /* FA2db.c:652-905 */
{ char   path[2 * MAX_NAME + 16], core[MAX_NAME + 8], prolog[MAX_NAME + 8];
  char  *line = NULL, *seq = NULL, header[MAX_NAME + 8];
  size_t lcap = 0;
  ssize_t n;
  int    smax = 0, rlen = 0, seqid = -1, have = 0;
  int    cnt[2] = { -1, -1 }, nxt = 0;
  char  *bseq = NULL;                                 /* -b: the best read of the current well */
  int    bmax = 0, blen = 0, bseqid = 0, bpac = 0, bwell = 0, bbeg = 0, bend = 0;
  HdrArgs hargs[2];                                   /* (the reference's two alternating read records) */
  int     bargs = 0;
  FILE  *in;
  const char *b = base_name(name);
  size_t bl = strlen(b);

  if (bl > 6 && strcmp(b + bl - 6, ".fasta") == 0) bl -= 6;
  else if (bl > 3 && strcmp(b + bl - 3, ".fa") == 0) bl -= 3;
  if (bl >= MAX_NAME)
    { fprintf(stderr, "File name over %d chars: '%.200s'\n", MAX_NAME, b);
      exit(1);
    }
  memcpy(core, b, bl);
  core[bl] = '\0';
  if (file_seen(core))                               /* FA2db.c:676-688: a file cannot be added twice */
    { fprintf(stderr, "File %s.fasta is already in database\n", core);
      exit(1);
    }
  snprintf(path, sizeof(path), "%.*s%s.fasta", (int) (b - name), name, core);
  if ((in = fopen(path, "r")) == NULL)
    { snprintf(path, sizeof(path), "%.*s%s.fa", (int) (b - name), name, core);
      if ((in = fopen(path, "r")) == NULL)
        { fprintf(stderr, "FA2db: cannot open %s.fasta or %s.fa\n", core, core);
          exit(1);
        }
    }
  if ((n = getline(&line, &lcap, in)) < 1)
    { fprintf(stderr, "Skipping '%s', file is empty!\n", core);
      fclose(in);
      free(line);
      return;
    }
  if (o->verbose)
    fprintf(stderr, "Adding '%s' ...\n", core);
  o->nadded += 1;
  if (n > MAX_NAME - 2 || line[n - 1] != '\n')
    { fprintf(stderr, "File %s.fasta, Line 1: Fasta line is too long (> %d chars)\n", core, MAX_NAME - 2);
      exit(1);
    }
  if (line[0] != '>')
    { fprintf(stderr, "File %s.fasta, Line 1: First header in fasta file is missing\n", core);
      exit(1);
    }
  if (is_pacbio_header(line + 1))
    { const char *slash = strchr(line + 1, '/');
      snprintf(prolog, sizeof(prolog), "%.*s", (int) (slash - (line + 1)), line + 1);
    }
  else
    strcpy(prolog, "DAZZ_READ");

  memset(hargs, 0, sizeof(hargs));
  for (;;)                                           /* line holds a header here, or n < 0 at the end */
    { int pac = 0, well = -1, beg = -1, end = -1;
      const int cur = nxt;                           /* the record this header is parsed into */

      if (n < 0)
        break;
      snprintf(header, sizeof(header), "%s", line + 1);
      /* FA2db.c:744-750: the index inside the file is counted per read record of the reference's two
         alternating records; without -b only one of them is ever parsed into, so it is the plain index,
         with -b each counts the headers parsed into it -- reproduced as it is */
      cnt[nxt] += 1;
      seqid = cnt[nxt];
      if (is_pacbio_header(header))
        { const char *slash = strchr(header, '/');
          pac = (sscanf(slash + 1, "%d/%d_%d\n", &well, &beg, &end) == 3);
        }
      if (C_n > 0)
        { const char *c = header;                     /* past the read name and ONE blank */
          while (*c != '\0' && *c != '\n' && *c != ' ')
            c++;
          if (*c == ' ' || *c == '\n')
            c++;
          parse_header_args(c, &hargs[cur]);
        }
      rlen = 0;
      have = 0;
      while ((n = getline(&line, &lcap, in)) >= 0)
        { if (line[0] == '>')
            { have = 1;
              break;
            }
          if (n > 0 && line[n - 1] == '\n')
            n -= 1;
          if (rlen + n + 1 > smax)
            { smax = (int) (1.2 * (rlen + n)) + 1000;
              seq = (char *) realloc(seq, (size_t) smax);
            }
          memcpy(seq + rlen, line, (size_t) n);
          rlen += (int) n;
        }
      if (rlen < o->minlen)
        { if (o->verbose > 1)
            fprintf(stderr, "Warning: skipping read of length %d\n", rlen);
        }
      else if (C_n > 0 && o->fullhq && !full_hq_read(&hargs[cur]))
        ;                                            /* -Q (FA2db.c:809-837): not a FullHqRead, or it does not say */
      else if (!o->best)
        add_read(o, seq, rlen, seqid, pac, well, beg, end, C_n > 0 ? &hargs[cur] : NULL);
      else
        { /* -b (FA2db.c:858-893): of consecutive reads of one well only the longest enters the database (the
             first of equally long ones; reads without a PacBio header all count as well -1) */
          if (bseq != NULL && bwell == well)
            { if (blen < rlen)
                goto take;
            }
          else
            { if (bseq != NULL)
                add_read(o, bseq, blen, bseqid, bpac, bwell, bbeg, bend, C_n > 0 ? &hargs[bargs] : NULL);
            take:
              if (rlen + 1 > bmax)
                { bmax = rlen + rlen / 4 + 1000;
                  bseq = (char *) realloc(bseq, (size_t) bmax);
                }
              memcpy(bseq, seq, (size_t) rlen);
              blen = rlen;  bseqid = seqid;  bpac = pac;  bwell = well;  bbeg = beg;  bend = end;
              bargs = cur;
              nxt ^= 1;                              /* the record just parsed is the best one now: parse into the other */
            }
        }
      if (!have)
        break;
    }
  if (o->best && bseq != NULL)
    add_read(o, bseq, blen, bseqid, bpac, bwell, bbeg, bend, C_n > 0 ? &hargs[bargs] : NULL);
  free(bseq);
  fprintf(o->stub, "  %9d %s %s\n", o->ureads, core, prolog);
  fclose(in);
  free(line);
  free(seq);
}

int main(int argc, char *argv[])
{ Out     o;
  HITS_DB db;
  char   *root, *dir, path[2 * MAX_NAME + 16], newstub[2 * MAX_NAME + 16];
  int     c, i, ofiles = 0, first = 0, newblock = 0;
  FILE   *flist = NULL, *istub;

  memset(&o, 0, sizeof(o));
  o.minlen = 1000;
  opterr = 0;
  while ((c = getopt(argc, argv, "vabQx:c:f:")) != -1)
    switch (c)
    { case 'v': o.verbose += 1; break;
      case 'x': o.minlen = atoi(optarg); break;
      case 'a': newblock = 1; break;
      case 'b': o.best = 1; break;
      case 'f':
        if ((flist = fopen(optarg, "r")) == NULL)
          { fprintf(stderr, "Cannot open file of inputs '%s'\n", optarg);
            exit(1);
          }
        break;
      case 'Q': o.fullhq = 1; break;
      case 'c':
        C_name = (char **) realloc(C_name, sizeof(char *) * (size_t) (C_n + 1));
        C_name[C_n++] = optarg;
        break;
      default:
        fprintf(stderr, "usage: FA2db [-vabQ] [-c<track>] [-x<int(1000)>] <path:db> (-f<file> | <input:fasta> ...)\n");
        exit(1);
    }
  if (o.minlen < 0)
    { fprintf(stderr, "invalid min read length of %d\n", o.minlen);
      exit(1);
    }
  if ((flist == NULL && argc - optind < 2) || argc - optind < 1)
    { fprintf(stderr, "usage: FA2db [-vabQ] [-c<track>] [-x<int(1000)>] <path:db> (-f<file> | <input:fasta> ...)\n");
      exit(1);
    }
  root = damar_root(argv[optind], ".db");
  { const char *s = strrchr(argv[optind], '/');
    dir = s ? strndup(argv[optind], (size_t) (s - argv[optind])) : strdup(".");
  }
  snprintf(path, sizeof(path), "%s/%s.db", dir, root);
  snprintf(newstub, sizeof(newstub), "%s/%s.dbx", dir, root);       /* FA2db.c:571: the new image replaces the old */
  istub = fopen(path, "r");
  if ((o.stub = fopen(newstub, "w+")) == NULL)
    { fprintf(stderr, "FA2db: cannot create %s\n", newstub);
      exit(1);
    }
  memset(&db, 0, sizeof(db));
  if (istub == NULL)
    { snprintf(path, sizeof(path), "%s/.%s.idx", dir, root);
      o.idx = fopen(path, "w+");
      snprintf(path, sizeof(path), "%s/.%s.bps", dir, root);
      o.bps = fopen(path, "w+");
      if (o.idx == NULL || o.bps == NULL)
        { fprintf(stderr, "FA2db: cannot create the database files of %s\n", root);
          exit(1);
        }
      fwrite(&db, sizeof(db), 1, o.idx);             /* place holder, rewritten below (FA2db.c:1114-1131) */
      fprintf(o.stub, "files = %9d\n", 0);
    }
  else                                               /* FA2db.c:533-590: add to an existing database */
    { if (fscanf(istub, "files = %9d\n", &ofiles) != 1)
        { fprintf(stderr, "FA2db: stub file of %s is junk\n", root);
          exit(1);
        }
      snprintf(path, sizeof(path), "%s/.%s.idx", dir, root);
      o.idx = fopen(path, "r+");
      snprintf(path, sizeof(path), "%s/.%s.bps", dir, root);
      o.bps = fopen(path, "r+");
      if (o.idx == NULL || o.bps == NULL || fread(&db, sizeof(db), 1, o.idx) != 1)
        { fprintf(stderr, "FA2db: cannot open the database files of %s\n", root);
          exit(1);
        }
      fseeko(o.bps, 0, SEEK_END);
      fseeko(o.idx, 0, SEEK_END);
      first = o.ureads = db.ureads;
      o.off = ftello(o.bps);
      fprintf(o.stub, "files = %9d\n", 0);
      for (i = 0; i < ofiles; i++)
        { int  last;
          char fname[MAX_NAME + 8], prolog[MAX_NAME + 8];
          if (fscanf(istub, "  %9d %" STR(MAX_NAME) "s %" STR(MAX_NAME) "s\n", &last, fname, prolog) != 3)
            { fprintf(stderr, "FA2db: stub file of %s is junk\n", root);
              exit(1);
            }
          file_seen(fname);
          fprintf(o.stub, "  %9d %s %s\n", last, fname, prolog);
        }
      o.nadded = ofiles;
    }
  if (flist != NULL)                                   /* fileUtils.c:62-95: one name per line */
    { char nm[MAX_NAME + 8];
      while (fgets(nm, sizeof(nm), flist) != NULL)
        { size_t l = strlen(nm);
          if (l > 0 && nm[l - 1] == '\n')
            nm[l - 1] = '\0';
          if (nm[0] != '\0')
            read_fasta(&o, nm);
        }
      fclose(flist);
    }
  else
    for (i = optind + 1; i < argc; i++)
      read_fasta(&o, argv[i]);

  if (istub == NULL)                                   /* FA2db.c:1093-1110 */
    { for (c = 0; c < 4; c++)
        db.freq[c] = (float) ((1. * o.count[c]) / o.totlen);
      db.totlen = o.totlen;
      db.maxlen = o.maxlen;
    }
  else
    { for (c = 0; c < 4; c++)
        db.freq[c] = (float) ((db.freq[c] * db.totlen + (1. * o.count[c])) / (db.totlen + o.totlen));
      db.totlen += o.totlen;
      if (o.maxlen > db.maxlen)
        db.maxlen = o.maxlen;
    }
  db.ureads = o.ureads;

  { int nblock;                                        /* FA2db.c:908-975: extend an existing block partition */
    if (istub != NULL && fscanf(istub, "blocks = %9d\n", &nblock) == 1)
      { long long size;
        long  pos = ftell(o.stub);
        int   ufirst = 0, ireads = 0;
        int64 tot = 0;
        HITS_READ rec;
        if (o.verbose)
          fprintf(stderr, "Updating block partition ...\n");
        fprintf(o.stub, "blocks = %9d\n", 0);
        if (fscanf(istub, "size = %9lld\n", &size) != 1)
          { fprintf(stderr, "FA2db: stub file of %s is junk\n", root);
            exit(1);
          }
        fprintf(o.stub, "size = %9lld\n", size);
        size *= 1000000ll;
        if (!newblock)
          nblock -= 1;
        for (i = 0; i <= nblock; i++)
          { if (fscanf(istub, " %9d\n", &ufirst) != 1)
              { fprintf(stderr, "FA2db: stub file of %s is junk\n", root);
                exit(1);
              }
            fprintf(o.stub, " %9d\n", ufirst);
          }
        fflush(o.idx);
        fseeko(o.idx, (off_t) (sizeof(HITS_DB) + sizeof(HITS_READ) * (size_t) ufirst), SEEK_SET);
        for (i = ufirst; i < o.ureads; i++)
          { if (fread(&rec, sizeof(HITS_READ), 1, o.idx) != 1)
              { fprintf(stderr, "FA2db: index of %s is truncated\n", root);
                exit(1);
              }
            ireads += 1;
            tot += rec.rlen;
            if (tot >= size)
              { fprintf(o.stub, " %9d\n", i + 1);
                tot = 0;
                ireads = 0;
                nblock += 1;
              }
          }
        if (ireads > 0)
          { fprintf(o.stub, " %9d\n", o.ureads);
            nblock += 1;
          }
        fseek(o.stub, pos, SEEK_SET);
        fprintf(o.stub, "blocks = %9d\n", nblock);
      }
  }
  rewind(o.idx);
  fwrite(&db, sizeof(db), 1, o.idx);
  rewind(o.stub);                                      /* files actually in the database (empty ones are skipped) */
  fprintf(o.stub, "files = %9d\n", o.nadded);
  if (istub != NULL)
    fclose(istub);
  fclose(o.stub);
  fclose(o.idx);
  fclose(o.bps);
  snprintf(path, sizeof(path), "%s/%s.db", dir, root);
  rename(newstub, path);
  if (first > 0 && C_n > 0)
    { /* adding to a database with -c: every track the database has is carried on over the new reads, whether they
         name it or not (FA2db.c:406-494 looks the .anno files up in the directory) */
      DIR *dp = opendir(dir);
      struct dirent *e;
      const size_t rl = strlen(root);
      while (dp != NULL && (e = readdir(dp)) != NULL)
        { const char *n = e->d_name;
          const size_t l = strlen(n);
          if (n[0] == '.' && l > rl + 7 && strncmp(n + 1, root, rl) == 0 && n[1 + rl] == '.' && strcmp(n + l - 5, ".anno") == 0)
            { char tn[512];
              snprintf(tn, sizeof(tn), "%.*s", (int) (l - rl - 7), n + rl + 2);
              if (strchr(tn, '.') != NULL || strcmp(tn, "seqID") == 0)
                continue;
              if (strcmp(tn, "pacbio") == 0)
                T_pac_used = 1;
              else
                (void) hdr_track(tn);
            }
        }
      if (dp != NULL)
        closedir(dp);
    }
  if (o.ureads > first)                                /* FA2x.c:192: no read added, no track touched */
    { int t;
      track_write(&T_seq, dir, root, first, o.ureads);
      if (T_pac_used)
        track_write(&T_pac, dir, root, first, o.ureads);
      for (t = 0; t < T_nhdr; t++)
        track_write(T_hdr[t], dir, root, first, o.ureads);
    }
  return 0;
}

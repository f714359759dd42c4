/* daligner.c -- host driver of the MI355X overlapper: same command line, block-pair
 * loop, output directories and .las files as the reference's dalign/daligner.c:662-1077,
 * calling the three-function filter interface of libdamar_hip.so (damar_filter.h).
 * Host code stays C; every heavy step behind Sort_Kmers / Match_Filter runs on the GPU.
 *
 * The B blocks of the line are read, checked and reverse-complemented one or two ahead on a second
 * thread while the GPU works on the current one.  Rejected explicitly: -D (dynamic mask server).
 * -H is accepted and has no effect, exactly like the reference.
 *
 * Plan mode, `daligner [options] -P <plan file | ->`: every `daligner ...` line of an HPCdaligner plan
 * (dalign/HPCdaligner.c:628-788) in ONE process.  The reference runs one process per line
 * (daligner.c:662-1077), so every line reads its blocks again and sorts their k-mers again; here a block
 * is read, reverse-complemented, uploaded and indexed once and stays resident (HBM: 1.25 B per base and
 * strand + 8 B per k-mer) for all the lines that name it; least recently used idle blocks are released beyond
 * DAMAR_PLAN_BLOCKS blocks (default 64) or beyond a byte budget (DAMAR_PLAN_GB, default 55 % of the GPU's memory: the
 * seed arenas and the alignment scratch need the rest).  Options in front of -P apply to every line; a line's own
 * options are parsed on top of them.  Output files are those of the separate commands.
 *
 * Node mode, `daligner [options] -P <plan> -G <n | i,j,...> [-L]`: the plan's block pairs over several GPUs of one node --
 * the scheduler north_star asks for in place of dalign/daligner.c:958 under the work list of dalign/HPCdaligner.c:628-788.
 * The parent parses the plan, cuts the block pairs into one region per GPU (an A range x subject range of the plan's
 * triangle: a worker that stays inside its region builds few k-mer indexes), forks one worker per GPU BEFORE any HIP call
 * and waits; a worker pulls units (one A block against up to 8 subject blocks, both orientations) from its region's
 * cursor -- a C11 atomic in a page shared by the processes -- and, when that is empty, from the other regions' (work
 * stealing without victims).  No data moves between the workers: every pair writes its own files.  When the plan has
 * fewer than two pairs per GPU, cross pairs are split by B-read range and the parent merges the parts (LAmerge).
 * With -L the parent also runs the plan's own LAmerge lines (HPCdaligner.c:790-808) once the pairs are done.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <stdatomic.h>
#include <limits.h>
#include <dirent.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>

#include "damar_filter.h"
#include "damar_hip.h"
#include "damar_gate.h"

static void usage(void)
{ fprintf(stderr, "usage:\n");
  fprintf(stderr, "daligner [-vbAIOT] [-k<int(14)>] [-w<int(6)>] [-h<int(35)>] [-t<int>] [-M<int>] [-m<track>]+\n");
  fprintf(stderr, "         [-e<double(.70)] [-l<int(1000)>] [-s<int(100)>] [-H<int>] [-j<int>]\n");
  fprintf(stderr, "         [-r<int(1)>] [-g<gpu ordinal(0)>] <subject:db> <target:db> ...\n");
  fprintf(stderr, "daligner [options] -P <HPCdaligner plan file, or - for stdin> [-G <GPUs: n | i,j,...>] [-L]\n");
}

/* Output directories are named relative to the working directory, as the reference names them; a worker that runs one
   PART of a split block pair writes under OUT_base instead (absolute: the files are written later, on another thread). */
static char *OUT_base = NULL;

static char *out_dir(int run, int part)
{ char *d = damar_get_dir(run, part);
  if (OUT_base != NULL)
    { char *x = (char *) malloc(strlen(OUT_base) + strlen(d) + 2);
      sprintf(x, "%s/%s", OUT_base, d);
      free(d);
      d = x;
    }
  return d;
}

static void make_subdir(const HITS_DB *block, int run)      /* daligner.c:630-660 */
{ char *d = out_dir(run, block->part);
  struct stat s;
  if (stat(d, &s) != 0)
    { if (errno == ENOENT)
        mkdir(d, S_IRWXU | S_IRGRP | S_IXGRP | S_IROTH | S_IXOTH);
      else
        { fprintf(stderr, "Cannot create output directory: %s\n", d);
          exit(1);
        }
    }
  else if (!S_ISDIR(s.st_mode))
    { fprintf(stderr, "Output directory name: \"%s\" exist - but its not a directory\n", d);
      exit(1);
    }
  free(d);
}

static int check_reads_ok(const HITS_DB *b, const char *name, int kmer)     /* daligner.c:499-504 */
{ int i;
  for (i = 0; i < b->nreads; i++)
    if (b->reads[i].rlen < kmer)
      { fprintf(stderr, "[ERROR] - daligner: Block %s contains reads < %dbp long !  Run DBsplit.\n", name, kmer);
        return 0;
      }
  return 1;
}

static void check_reads(const HITS_DB *b, const char *name, int kmer)
{ if (!check_reads_ok(b, name, kmer))
    exit(1);
}

/* DAMAR_CLIPROF=1: wall clock of the driver's phases on stderr at exit */
#include <time.h>
static double P_ms[12];
static const char *P_name[12] = { "read_block(2nd thread)", "Sort_Kmers", "Match_Filter", "complement(2nd thread)", "write_submit", "drain", "wait_for_block", "upload(2nd thread)",
                                  "New_Align_Spec", "line_setup", "block_get", "subdirs" };
static double wall_ms(void)
{ struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
#define TIMED(slot, stmt) do { double t0_ = wall_ms(); stmt; P_ms[slot] += wall_ms() - t0_; } while (0)
static double T_start;                                  /* DAMAR_CLIPROF: when main() was entered */
static void mark(const char *what)
{ if (getenv("DAMAR_CLIPROF"))
    fprintf(stderr, "cli: +%.1f ms %s\n", wall_ms() - T_start, what);
}

/* ---- B blocks prepared ahead (read_DB + Merge_Tracks + complement_DB of daligner.c:958-1034) ---- */
typedef struct
{ int     same;          /* the B block is the A block */
  HITS_DB blk;           /* forward block (unused if same) */
  HITS_DB cblk;          /* reverse-complemented copy */
} Prepared;

#define PF_DEPTH 2
static struct
{ char **names;  int n;
  const char *afile;  HITS_DB *ablock;
  char **mask;  int mtop, kmer;
  Prepared *items;
  int produced, consumed;
  pthread_mutex_t mu;
  pthread_cond_t  cv;
} PF;

/* a failure on the reader thread: the main thread may be inside HIP calls and the tail workers running, so
   no exit handlers (the library's die() leaves the same way) */
static void reader_fail(void)
{ fflush(stdout);
  fflush(stderr);
  _exit(1);
}

static void *prepare_blocks(void *arg)
{ int i;
  (void) arg;
  for (i = 0; i < PF.n; i++)
    { Prepared *it = PF.items + i;
      pthread_mutex_lock(&PF.mu);
      while (PF.produced - PF.consumed >= PF_DEPTH)
        pthread_cond_wait(&PF.cv, &PF.mu);
      pthread_mutex_unlock(&PF.mu);
      it->same = (strcmp(PF.afile, PF.names[i]) == 0);
      if (!it->same)
        { double t0 = wall_ms();
          if (damar_read_block(PF.names[i], &it->blk))
            reader_fail();
          if (damar_load_masks(&it->blk, PF.mask, PF.mtop))
            { printf("[ERROR] - Unable to load track!\n");
              reader_fail();
            }
          if (check_reads_ok(&it->blk, PF.names[i], PF.kmer) == 0)
            reader_fail();
          P_ms[0] += wall_ms() - t0;
          t0 = wall_ms();
          damar_block_preload(&it->blk);          /* to HBM on its own stream; Sort_Kmers picks it up */
          P_ms[7] += wall_ms() - t0;
        }
      { double t0 = wall_ms();
        damar_complement_copy(it->same ? PF.ablock : &it->blk, &it->cblk);
        P_ms[3] += wall_ms() - t0;
        t0 = wall_ms();
        damar_block_preload(&it->cblk);
        P_ms[7] += wall_ms() - t0;
      }
      pthread_mutex_lock(&PF.mu);
      PF.produced += 1;
      pthread_cond_broadcast(&PF.cv);
      pthread_mutex_unlock(&PF.mu);
    }
  return NULL;
}

static Prepared *next_prepared(int i)
{ double t0 = wall_ms();
  pthread_mutex_lock(&PF.mu);
  while (PF.produced <= i)
    pthread_cond_wait(&PF.cv, &PF.mu);
  pthread_mutex_unlock(&PF.mu);
  P_ms[6] += wall_ms() - t0;
  return PF.items + i;
}

static void done_with(int i)
{ (void) i;
  pthread_mutex_lock(&PF.mu);
  PF.consumed += 1;
  pthread_cond_broadcast(&PF.cv);
  pthread_mutex_unlock(&PF.mu);
}

typedef struct
{ int    kmer, hitmin, binshift, maxreps, spacing, runid, notrace, nthreads, only_id, gpu;
  double ecorr;
  int    minover, identity, symmetric, biased, verbose;
  int    have_mem;  int mem_gb;
  char  *mask[64];
  int    mtop;
  char  *plan;
  char  *gpus;          /* -G: node mode */
  int    lamerge;       /* -L: run the plan's LAmerge lines too */
} Opts;

static void default_opts(Opts *o)
{ memset(o, 0, sizeof(*o));
  o->kmer = 14;  o->hitmin = 35;  o->binshift = 6;  o->ecorr = .70;  o->spacing = 100;  o->runid = 1;
  o->nthreads = 4;  o->gpu = -1;  o->minover = 1000;  o->symmetric = 1;
}

/* daligner.c:721-812; returns the index of the first non-option argument */
static int parse_opts(int argc, char *argv[], Opts *o)
{ int c;
  opterr = 0;
  optind = 1;
  while ((c = getopt(argc, argv, "vbOTAIk:w:h:t:M:e:l:s:H:D:m:r:j:g:P:G:L")) != -1)
    switch (c)
    { case 'v': o->verbose = 1; break;
      case 'T': o->notrace = 1; break;
      case 'I': o->identity = 1; break;
      case 'O': o->identity = 1; o->only_id = 1; break;
      case 'A': o->symmetric = 0; break;
      case 'k': o->kmer = atoi(optarg); break;
      case 'w': o->binshift = atoi(optarg); break;
      case 'h': o->hitmin = atoi(optarg); break;
      case 't': o->maxreps = atoi(optarg); break;
      case 'H': break;
      case 'e': o->ecorr = atof(optarg); break;
      case 'l': o->minover = atoi(optarg); break;
      case 's': o->spacing = atoi(optarg); break;
      case 'j': o->nthreads = atoi(optarg); break;
      case 'r': o->runid = atoi(optarg); break;
      case 'g': o->gpu = atoi(optarg); break;
      case 'P': o->plan = optarg; break;
      case 'G': o->gpus = optarg; break;
      case 'L': o->lamerge = 1; break;
      case 'M':
        o->mem_gb = atoi(optarg);
        if (o->mem_gb < 0)
          fprintf(stderr, "invalid memory limit of (%d)\n", o->mem_gb);
        o->have_mem = 1;
        break;
      case 'm':
        if (o->mtop >= 64)
          { fprintf(stderr, "daligner: too many -m tracks\n");
            exit(1);
          }
        o->mask[o->mtop++] = optarg;
        break;
      case 'b': o->biased = 1; break;
      case 'D':
        fprintf(stderr, "daligner: option -%c is not supported by this build\n", c);
        exit(1);
      default:
        fprintf(stderr, "Unsupported option: %s\n", argv[optind - 1]);
        usage();
        exit(1);
    }
  if (o->kmer < 0 || o->binshift < 0 || o->hitmin < 0 || o->maxreps < 0 || o->minover < 0 || o->spacing < 0 || o->runid < 0)
    { fprintf(stderr, "daligner: negative option value\n");
      exit(1);
    }
  if (o->ecorr < .5 || o->ecorr >= 1.)
    { fprintf(stderr, "Average correlation must be in [.5,1.) (%g)\n", o->ecorr);
      exit(1);
    }
  return optind;
}

/* the device is chosen once per process, after all options are known: -g, then DAMAR_DEVICE, then 0; only
   then may -M override the memory limit (damar_hip_init sets MEM_LIMIT / MEM_PHYSICAL on its first call) */
static int GATE_gpu = 0;             /* the GPU whose teardown gate this process passed (damar_gate.h) */

static void select_device(const Opts *o)
{ int gpu = o->gpu;
  if (gpu < 0 && getenv("DAMAR_DEVICE") != NULL)
    gpu = atoi(getenv("DAMAR_DEVICE"));
  GATE_gpu = gpu < 0 ? 0 : gpu;
  damar_gate_wait(GATE_gpu);         /* (not into the teardown of the command before this one) */
  damar_hip_init(gpu < 0 ? 0 : gpu);
  if (o->have_mem)
    MEM_LIMIT = (uint64) o->mem_gb * 0x40000000ull;
}

static void apply_opts(const Opts *o)
{ VERBOSE = o->verbose;
  IDENTITY = o->identity;
  SYMMETRIC = o->symmetric;
  BIASED = o->biased;
  MINOVER = 2 * o->minover;                        /* daligner.c:861 */
  if (Set_Filter_Params(o->kmer, o->binshift, o->maxreps, o->hitmin, o->nthreads))
    { fprintf(stderr, "Illegal combination of filter parameters\n");
      exit(1);
    }
}

static int symmetric_for(const char *afile, const char *aroot, char **bfiles, int nb)     /* daligner.c:911-946 */
{ int i;
  for (i = 0; i < nb; i++)
    if (strcmp(afile, bfiles[i]) != 0)
      { char *broot = damar_root(bfiles[i], ".db");
        char *ad = strrchr(aroot, '.'), *bd = strrchr(broot, '.');
        size_t la = ad ? (size_t) (ad - aroot + 1) : strlen(aroot);
        size_t lb = bd ? (size_t) (bd - broot + 1) : strlen(broot);
        int differ = strncmp(aroot, broot, la > lb ? la : lb) != 0;
        if (differ && VERBOSE)
          printf("[WARNING] - Daligner is performed on different databases (%s - %s). SYMMETRIC option is disabled!\n",
                 aroot, broot);
        free(broot);
        if (differ)
          return 0;
      }
  return 1;
}

static int plan_main(const Opts *base, const char *planfile);

int main(int argc, char *argv[])
{ HITS_DB ablock;
  char   *afile, *aroot;
  void   *aindex, *bindex;
  int     alen, blen;
  Align_Spec *spec;
  Opts    O;
  int     kmer, runid, nthreads;
  char  **mask;
  int     mtop, i;

  T_start = wall_ms();
  default_opts(&O);
  optind = parse_opts(argc, argv, &O);
  if (O.plan != NULL)
    return plan_main(&O, O.plan);
  kmer = O.kmer;  runid = O.runid;  nthreads = O.nthreads;  mask = O.mask;  mtop = O.mtop;
  if (optind + 2 > argc)
    { fprintf(stderr, "[ERROR] - at least one target and one subject block are required\n\n");
      usage();
      exit(1);
    }
  select_device(&O);
  apply_opts(&O);

  afile = argv[optind++];
  { double t0_ = wall_ms();
    if (damar_read_block(afile, &ablock))
      exit(1);
    P_ms[0] += wall_ms() - t0_;
  }
  if (damar_load_masks(&ablock, mask, mtop))
    { printf("[ERROR] - Unable to load track!\n");
      exit(1);
    }
  check_reads(&ablock, afile, kmer);
  aroot = damar_root(afile, ".db");

  if (SYMMETRIC)
    SYMMETRIC = symmetric_for(afile, aroot, argv + optind, argc - optind);

  make_subdir(&ablock, runid);
  spec = New_Align_Spec(O.ecorr, O.spacing, ablock.freq, nthreads, SYMMETRIC, O.only_id, O.notrace, 1);

  /* The host tail (redundancy handling, sort, .las write) of a block pair runs on a worker
     thread while the GPU starts on the next pair; B blocks stay alive until it is done. */
  damar_set_async(1);
  { const int nb = argc - optind;
    Prepared *pending[8];
    int       npending = 0, k;
    pthread_t th;

    PF.names = argv + optind;  PF.n = nb;
    PF.afile = afile;  PF.ablock = &ablock;
    PF.mask = mask;  PF.mtop = mtop;  PF.kmer = kmer;
    PF.items = (Prepared *) calloc((size_t) nb + 1, sizeof(Prepared));
    PF.produced = PF.consumed = 0;
    pthread_mutex_init(&PF.mu, NULL);
    pthread_cond_init(&PF.cv, NULL);
    /* (the device was selected above, before the second thread makes its first HIP call) */
    if (pthread_create(&th, NULL, prepare_blocks, NULL) != 0)
      { fprintf(stderr, "daligner: cannot start the block reader thread\n");
        exit(1);
      }

    aindex = NULL;
    alen = 0;
    for (k = 0; k < nb; k++)
      { char     *bfile = argv[optind + k];
        Prepared *it;
        char     *broot = NULL;

        if (k == 0)
          { if (VERBOSE)
              printf("\nBuilding index for %s\n", aroot);
            TIMED(1, aindex = Sort_Kmers(&ablock, &alen));
          }
        it = next_prepared(k);
        if (!it->same)
          { char *d1 = NULL, *d2 = NULL;
            int   last;
            broot = damar_root(bfile, ".db");
            if (SYMMETRIC)
              make_subdir(&it->blk, runid);
            if (VERBOSE)
              printf("\nBuilding index for %s\n", broot);
            TIMED(1, bindex = Sort_Kmers(&it->blk, &blen));
            TIMED(2, Match_Filter(aroot, &ablock, broot, &it->blk, aindex, alen, bindex, blen, 0, spec));
            if (VERBOSE)
              printf("\nBuilding index for c(%s)\n", broot);
            TIMED(1, bindex = Sort_Kmers(&it->cblk, &blen));
            TIMED(2, Match_Filter(aroot, &ablock, broot, &it->cblk, aindex, alen, bindex, blen, 1, spec));

            last = (it->blk.part < ablock.part) ? it->blk.ufirst + it->blk.nreads - 1
                                                : ablock.ufirst + ablock.nreads - 1;
            if (ablock.part > 0)  d1 = damar_get_dir(runid, ablock.part);
            if (it->blk.part > 0) d2 = damar_get_dir(runid, it->blk.part);
            TIMED(4, damar_write_overlaps(spec, d1, d2, aroot, broot, last));
            free(d1);
            free(d2);
            free(broot);
          }
        else
          { char *d1 = NULL;
            TIMED(2, Match_Filter(aroot, &ablock, aroot, &ablock, aindex, alen, aindex, alen, 0, spec));
            if (VERBOSE)
              printf("\nBuilding index for c(%s)\n", aroot);
            TIMED(1, bindex = Sort_Kmers(&it->cblk, &blen));
            TIMED(2, Match_Filter(aroot, &ablock, aroot, &it->cblk, aindex, alen, bindex, blen, 1, spec));
            if (ablock.part > 0) d1 = damar_get_dir(runid, ablock.part);
            TIMED(4, damar_write_overlaps(spec, d1, NULL, aroot, aroot, ablock.ufirst + ablock.nreads - 1));
            free(d1);
          }
        /* the host tail of this pair may still read the two B blocks on its thread: they are released after
           a drain, a few pairs later; the reader thread may go on to the next block right away */
        pending[npending++] = it;
        done_with(k);
        if (npending >= 4)
          { TIMED(5, damar_async_drain());
            while (npending > 0)
              { Prepared *p = pending[--npending];
                damar_free_complement(&p->cblk);
                if (!p->same)
                  damar_close_block(&p->blk);
              }
          }
      }
    TIMED(5, damar_async_drain());
    while (npending > 0)
      { Prepared *p = pending[--npending];
        damar_free_complement(&p->cblk);
        if (!p->same)
          damar_close_block(&p->blk);
      }
    pthread_join(th, NULL);
    free(PF.items);
  }
  damar_set_async(0);
  if (getenv("DAMAR_CLIPROF"))
    { fprintf(stderr, "cli: wall ms:");
      for (i = 0; i < 12; i++)
        fprintf(stderr, " %s=%.1f", P_name[i], P_ms[i]);
      fprintf(stderr, "\n");
    }
  return 0;
}

/* Plan mode and node mode fork their workers on the premise that this process has not touched the GPU yet.  Under
   rocprofv3 (and anything else that preloads a tool library into the process) the HIP runtime is up before main(), and a
   child of such a process must not use it: plan mode then runs in-process, node mode refuses. */

/* ---------------------------------------------------------------------------------------------------
 * Plan mode: many `daligner` lines, one process, blocks and k-mer indexes resident (see the header).
 * The C mirror of damar_amd/driver.py's Plan on the device-resident entry points of damar_hip.h.
 * ------------------------------------------------------------------------------------------------- */

typedef struct
{ char    *name;
  HITS_DB  blk, cblk;
  damar_dev_block *dev[2];          /* forward / complement bases in HBM */
  damar_dev_index *idx[2];          /* their k-mer indexes */
  int      ilen[2];
  long     used;                    /* LRU stamp */
  int      busy;                    /* named by the pair being computed */
  int      ready;                   /* 0 while the reader thread is still preparing it */
  int      hostready;               /* read and complemented on the host (the reader threads' first stage) */
  damar_packed pk;                  /* packed != 0: the block is kept as its stretch of the .bps file (blk.bases == NULL), */
  int      packed;                  /* the GPU unpacks and reverse-complements it (damar_block_upload_packed) */
} PBlock;

static PBlock *PB;                  /* fixed capacity (PB_max + 8): entries never move while the reader thread fills them */
static int     PB_n, PB_cap, PB_max = 64;
static long    PB_clock;
static int     PB_builds;
static pthread_mutex_t PB_mu = PTHREAD_MUTEX_INITIALIZER;
static int PB_dev_ready = 0;        /* the device is selected: the reader thread may upload */
static pthread_cond_t  PB_cv = PTHREAD_COND_INITIALIZER;
static int     PB_ahead;            /* entries [0, PB_ahead) are prepared by the reader thread, in this order */
static Opts    PB_opts;

/* read, mask, check, reverse-complement and (on the reader thread) upload one block, both strands */
static int S_loads_;
static void pblock_load_host(PBlock *b, const Opts *o, int background)
{ double t0 = wall_ms();
  __atomic_fetch_add(&S_loads_, 1, __ATOMIC_RELAXED);
  static int unpacked = -1;          /* DAMAR_DB_UNPACKED=1: blocks unpacked and complemented on the host as until round 4 (test hook) */
  if (unpacked < 0)
    unpacked = getenv("DAMAR_DB_UNPACKED") != NULL;
  b->packed = 0;
  if (unpacked)
    { if (damar_read_block(b->name, &b->blk))
        { if (background) reader_fail(); else exit(1); }
    }
  else
    { int r = damar_read_block_packed(b->name, &b->blk, &b->pk);
      if (r < 0)
        { if (background) reader_fail(); else exit(1); }
      b->packed = (r == 0);
    }
  if (damar_load_masks(&b->blk, (char **) o->mask, o->mtop))
    { printf("[ERROR] - Unable to load track!\n");
      if (background) reader_fail(); else exit(1);
    }
  if (!check_reads_ok(&b->blk, b->name, o->kmer))
    { if (background) reader_fail(); else exit(1); }
  P_ms[0] += wall_ms() - t0;
  t0 = wall_ms();
  damar_complement_copy(&b->blk, &b->cblk);
  P_ms[3] += wall_ms() - t0;
}

static void pblock_upload(PBlock *b)         /* on the copy stream, beside the kernels of the main thread */
{ double t0 = wall_ms();
  if (b->packed)
    { b->dev[0] = damar_block_upload_packed(&b->blk, &b->pk, 0);
      b->dev[1] = damar_block_upload_packed(&b->cblk, &b->pk, 1);
    }
  else
    { b->dev[0] = damar_block_upload_bg(&b->blk);
      b->dev[1] = damar_block_upload_bg(&b->cblk);
    }
  P_ms[7] += wall_ms() - t0;
}

static void pblock_load(PBlock *b, const Opts *o, int background)
{ pblock_load_host(b, o, background);
  if (background)
    { pthread_mutex_lock(&PB_mu);     /* (reading and complementing started before the device was up: plan_main) */
      while (!PB_dev_ready)
        pthread_cond_wait(&PB_cv, &PB_mu);
      pthread_mutex_unlock(&PB_mu);
      pblock_upload(b);
    }
}

static int PB_next;                 /* next entry a reader thread takes */

static int PB_up_next;              /* next entry to upload (in order, once it is read and the device is up) */

/* Two stages per block: read + complement (host only), then the upload.  A reader thread prefers an upload that is
   due; while the device is still coming up -- a tenth of a second in a cold process -- it reads on instead of waiting
   with its first block in hand, so that by then every block is ready to go up (plan of 4 blocks: the last block in HBM
   at +170 instead of +205 ms). */
static void *plan_reader(void *arg)
{ (void) arg;
  pthread_mutex_lock(&PB_mu);
  for (;;)
    { int i;
      if (PB_dev_ready && PB_up_next < PB_ahead && PB[PB_up_next].hostready)
        { i = PB_up_next++;
          pthread_mutex_unlock(&PB_mu);
          pblock_upload(PB + i);
          pthread_mutex_lock(&PB_mu);
          PB[i].ready = 1;
          pthread_cond_broadcast(&PB_cv);
          continue;
        }
      if (PB_next < PB_ahead)
        { i = PB_next++;
          pthread_mutex_unlock(&PB_mu);
          pblock_load_host(PB + i, &PB_opts, 1);
          pthread_mutex_lock(&PB_mu);
          PB[i].hostready = 1;
          pthread_cond_broadcast(&PB_cv);
          continue;
        }
      if (PB_up_next >= PB_ahead)
        break;
      pthread_cond_wait(&PB_cv, &PB_mu);            /* for the device, or for the block that is next to go up */
    }
  pthread_mutex_unlock(&PB_mu);
  return NULL;
}

static void *prewarm_thread(void *arg)
{ damar_prewarm(*(int *) arg);
  return NULL;
}

static void pblock_release(PBlock *b)
{ int c;
  for (c = 0; c < 2; c++)
    { if (b->idx[c] != NULL) damar_index_free(b->idx[c]);
      if (b->dev[c] != NULL) damar_block_free(b->dev[c]);
      b->idx[c] = NULL;  b->dev[c] = NULL;
    }
  if (b->packed)
    { damar_packed_forget(&b->blk);
      damar_packed_forget(&b->cblk);
      damar_free_packed(&b->pk);
      b->packed = 0;
    }
  damar_free_complement(&b->cblk);
  damar_close_block(&b->blk);
  free(b->name);
}

static void pblock_flush_indexes(void)             /* the filter parameters changed: indexes are stale */
{ int i, c;
  for (i = 0; i < PB_n; i++)
    for (c = 0; c < 2; c++)
      if (PB[i].idx[c] != NULL)
        { damar_index_free(PB[i].idx[c]);
          PB[i].idx[c] = NULL;
        }
}

/* HBM the table's blocks hold (bases of both strands and their indexes) against the byte budget: least recently used
   idle blocks go first.  A block that is needed again is read, complemented and indexed again. */
static uint64_t PB_budget = 0;

static uint64_t pblock_bytes(const PBlock *b)
{ return damar_block_bytes(b->dev[0]) + damar_block_bytes(b->dev[1]) + damar_index_bytes(b->idx[0]) + damar_index_bytes(b->idx[1]); }

static void pblock_release(PBlock *b);

static int PB_sharers = 1;          /* processes of this command on the same GPU (node mode with DAMAR_SHARE_GPU) */

static void pblock_trim(void)
{ if (PB_budget == 0)
    { /* 55 % of what this process can count on: the HBM that is free now PLUS what its own blocks already hold (the
         readers have uploaded blocks and an index has been built by the time this runs first; sampling "free" alone
         would discount them twice -- ADVICE r4).  Memory somebody else holds (the worker of the command before this one
         may still be releasing its HBM: plan_main) is not counted on.  When the workers of a node command sit on one
         GPU the others' allocations are already missing from "free": the share is then taken of the whole GPU, and
         capped by what is there */
      uint64_t fr = 0, tot = 0, own = 0, avail;
      int i;
      damar_hbm_info(&fr, &tot);
      pthread_mutex_lock(&PB_mu);
      for (i = 0; i < PB_n; i++)
        if (PB[i].ready && PB[i].name != NULL)
          own += pblock_bytes(PB + i);
      pthread_mutex_unlock(&PB_mu);
      avail = fr + own < tot ? fr + own : tot;
      if (PB_sharers > 1 && tot / (uint64_t) PB_sharers < avail)
        avail = tot / (uint64_t) PB_sharers;
      PB_budget = (uint64_t) (.55 * (double) avail);
      if (getenv("DAMAR_PLAN_GB") != NULL && atof(getenv("DAMAR_PLAN_GB")) > 0)
        PB_budget = (uint64_t) (atof(getenv("DAMAR_PLAN_GB")) * 1073741824.);
    }
  for (;;)
    { uint64_t sum = 0;
      int i, v = -1, vi = -1;
      pthread_mutex_lock(&PB_mu);                   /* (the reader threads set dev[] and ready under it) */
      for (i = 0; i < PB_n; i++)
        if (PB[i].ready && PB[i].name != NULL)
          { sum += pblock_bytes(PB + i);
            /* a victim is idle and has been used: a block the readers have just uploaded ahead of its first use stays.
               Indexes go first (8 bytes per k-mer against 1.5 per base, and a build is a few ms on the GPU while the bases
               come over PCIe): a block loses its bases only when no idle block has an index left */
            if (!PB[i].busy && PB[i].used > 0 && (PB[i].idx[0] || PB[i].idx[1]) && (vi < 0 || PB[i].used < PB[vi].used))
              vi = i;
            if (!PB[i].busy && PB[i].used > 0 && (PB[i].dev[0] || PB[i].dev[1]) && (v < 0 || PB[i].used < PB[v].used))
              v = i;
          }
      if (v < 0 && vi < 0 && sum > PB_budget)      /* nothing used and idle is left: the block the readers brought that is
                                                      needed LAST gives its bases back (they are uploaded again from the
                                                      host copy when its turn comes: pblock_index) -- ADVICE r5 */
        for (i = PB_n - 1; i >= 0 && v < 0; i--)
          if (PB[i].ready && PB[i].name != NULL && !PB[i].busy && PB[i].used == 0 && (PB[i].dev[0] || PB[i].dev[1]))
            v = i;
      pthread_mutex_unlock(&PB_mu);
      if (sum <= PB_budget || (v < 0 && vi < 0))
        return;
      if (vi >= 0)
        { int c;                                    /* (nothing in flight reads an index once damar_match_batch has returned) */
          for (c = 0; c < 2; c++)
            { if (PB[vi].idx[c] != NULL) damar_index_free(PB[vi].idx[c]);
              PB[vi].idx[c] = NULL;              /* (its buffers are parked in the library's pool: the next build takes them) */
            }
          continue;
        }
      damar_async_drain();                          /* the host tail may still read its bases */
      { int c;                                      /* device side only: the host copy stays for a cheap return */
        for (c = 0; c < 2; c++)
          { if (PB[v].dev[c] != NULL) damar_block_free(PB[v].dev[c]);
            PB[v].dev[c] = NULL;
          }
      }
      damar_pool_trim();                            /* (the freed index went to the library's pool: really give it back) */
    }
}

static PBlock *pblock_get(const char *name, const Opts *o)
{ int i;
  for (i = 0; i < PB_n; i++)
    if (strcmp(PB[i].name, name) == 0)
      { if (!PB[i].ready)                          /* the reader thread is at it */
          { double t0 = wall_ms();
            pthread_mutex_lock(&PB_mu);
            while (!PB[i].ready)
              pthread_cond_wait(&PB_cv, &PB_mu);
            pthread_mutex_unlock(&PB_mu);
            P_ms[6] += wall_ms() - t0;
          }
        PB[i].used = ++PB_clock;
        return PB + i;
      }
  if (PB_n >= PB_max)                               /* replace the least recently used idle block */
    { int v = -1;
      for (i = 0; i < PB_n; i++)
        if (!PB[i].busy && PB[i].ready && PB[i].used > 0 && (v < 0 || PB[i].used < PB[v].used))
          v = i;
      if (v >= 0)
        { damar_async_drain();                      /* the host tail may still read its bases */
          pblock_release(PB + v);
          memset(PB + v, 0, sizeof(PBlock));
          PB[v].name = strdup(name);
          PB[v].ready = 1;
          pblock_load(PB + v, o, 0);
          PB[v].used = ++PB_clock;
          return PB + v;
        }
    }
  if (PB_n >= PB_cap)
    { fprintf(stderr, "daligner: block table full (%d blocks busy at once)\n", PB_n);
      exit(1);
    }
  { PBlock *b = PB + PB_n++;
    memset(b, 0, sizeof(*b));
    b->name = strdup(name);
    b->ready = 1;
    pblock_load(b, o, 0);
    b->used = ++PB_clock;
    return b;
  }
}

static void stats_take(int lo, int hi);
static damar_dev_index *pblock_index(PBlock *b, int comp)
{ if (b->idx[comp] == NULL)
    { double t0 = wall_ms();
      if (b->dev[comp] == NULL)
        b->dev[comp] = b->packed ? damar_block_upload_packed(comp ? &b->cblk : &b->blk, &b->pk, comp)
                                 : damar_block_upload(comp ? &b->cblk : &b->blk);
      P_ms[7] += wall_ms() - t0;
      t0 = wall_ms();
      b->idx[comp] = damar_index_build(b->dev[comp], 0, &b->ilen[comp]);
      stats_take(DAMAR_T_TUPLES, DAMAR_T_MERGE);
      P_ms[1] += wall_ms() - t0;
      PB_builds += 1;
      pblock_trim();
    }
  return b->idx[comp];
}

/* What a plan has cost, summed over its calls (the library's own clocks, include/damar_hip.h DAMAR_T_*), for the one
   machine-readable line a run leaves behind: DAMAR_PLAN_STATS=<file> ("-": stderr) */
static double S_ms[DAMAR_T_COUNT];
static int64  S_seeds, S_pairs, S_work, S_resort;
static int    S_loads, S_tile, S_blocks, S_resident;
static double S_budget_gb;

static void stats_take(int lo, int hi)               /* the last call's timings [lo, hi) */
{ double t[DAMAR_T_COUNT];
  int    i;
  damar_last_timings(t);
  for (i = lo; i < hi; i++)
    S_ms[i] += t[i];
}

static void plan_stats_write(int nlines, int worker, int nworkers, double wall)
{ const char *dst = getenv("DAMAR_PLAN_STATS");
  static const char *nm[DAMAR_T_COUNT] = { "tuples", "ksort", "table", "merge", "ssort", "work", "report", "d2h", "tail" };
  int64  nf = 0, nl = 0, nrec = 0, las[4];
  double rms = 0, tail = 0, wr = 0;
  FILE  *f;
  int    i;
  if (dst == NULL)
    return;
  damar_async_counts(&nf, &rms, &nl);
  damar_async_totals(&nrec, &tail, &wr);
  S_ms[DAMAR_T_REPORT] += rms;  S_ms[DAMAR_T_TAIL] += tail;  S_ms[DAMAR_T_D2H] += damar_async_d2h_ms();
  damar_las_totals(las);
  f = (strcmp(dst, "-") == 0) ? stderr : fopen(dst, worker > 0 ? "a" : "w");
  if (f == NULL)
    return;
  fprintf(f, "{\"tool\": \"daligner -P\", \"worker\": %d, \"workers\": %d, \"plan_lines\": %d, \"block_pairs\": %lld, \"blocks\": %d, "
             "\"tile\": %d, \"bases_resident\": %d, \"budget_gb\": %.1f, \"index_builds\": %d, \"block_loads\": %d, \"wall_ms\": %.1f, "
             "\"seed_pairs\": %lld, \"work_items\": %lld, \"resorted\": %lld, \"local_alignments\": %lld, \"report_launches\": %lld, \"records\": %lld, "
             "\"aligned_bp\": %lld, \"las_bytes\": %lld, \"las_files\": %lld, \"phase_ms\": {",
          worker, nworkers, nlines, (long long) S_pairs, S_blocks, S_tile, S_resident, S_budget_gb, PB_builds, S_loads, wall,
          (long long) S_seeds, (long long) S_work, (long long) S_resort, (long long) nf, (long long) nl, (long long) las[2], (long long) las[3], (long long) las[0], (long long) las[1]);
  for (i = 0; i < DAMAR_T_COUNT; i++)
    fprintf(f, "\"%s\": %.1f, ", nm[i], S_ms[i]);
  fprintf(f, "\"write\": %.1f}, \"host_wall_ms\": {", wr);
  for (i = 0; i < 12; i++)
    fprintf(f, "%s\"%s\": %.1f", i ? ", " : "", P_name[i], P_ms[i]);
  fprintf(f, "}}\n");
  if (f != stderr)
    fclose(f);
}

static Align_Spec **PS;             /* the plan's Align_Specs, alive until the asynchronous tail has drained */
static int PS_n, PS_cap;

#define LINE_B 2                    /* blocks that can be busy beyond the table's limit: the A block and one subject block */

/* the queued tails and writes still use the spec: it is released when the plan is done (no drain per line) */
static void plan_keep_spec(Align_Spec *spec)
{ if (PS_n >= 512)                                  /* a long plan: every 512 block pairs the queue is drained and the specs
                                                      (tables and overlap buffers each) are released */
    { int i;
      damar_async_drain();
      for (i = 0; i < PS_n; i++)
        Free_Align_Spec(PS[i]);
      PS_n = 0;
    }
  if (PS_n >= PS_cap)
    { PS_cap = 2 * PS_cap + 64;
      PS = (Align_Spec **) realloc(PS, sizeof(Align_Spec *) * (size_t) PS_cap);
    }
  PS[PS_n++] = spec;
}

/* one plan line: daligner.c:948-1074 */
static void plan_line(const Opts *o, const char *afile, char **bfiles, int nb)
{ static Opts last;
  static int  have_last = 0;
  PBlock *a;
  char   *aroot;
  int     k;

  int masks_differ = have_last && last.mtop != o->mtop;
  if (have_last && !masks_differ)
    for (k = 0; k < o->mtop; k++)
      if (strcmp(last.mask[k], o->mask[k]) != 0)
        masks_differ = 1;
  if (have_last && (last.kmer != o->kmer || last.maxreps != o->maxreps || last.biased != o->biased || masks_differ))
    { damar_async_drain();
      pblock_flush_indexes();
      if (masks_differ)                           /* other mask tracks: the blocks themselves are stale
                                                     (plan_main starts no reader thread for such a plan) */
        { while (PB_n > 0)
            pblock_release(PB + --PB_n);
        }
    }
  last = *o;  have_last = 1;
  { const double t0 = wall_ms();
    apply_opts(o);
    a = pblock_get(afile, o);
    a->busy = 1;
    aroot = damar_root(afile, ".db");
    if (SYMMETRIC)
      SYMMETRIC = symmetric_for(afile, aroot, bfiles, nb);
    make_subdir(&a->blk, o->runid);
    P_ms[9] += wall_ms() - t0;
  }
  /* One damar_match_batch per subject block (both orientations: one launch of the report kernel).  The library leaves
     that launch in flight when the call returns: the next block's index builds and seed stages run beside it, and the
     write request below is queued behind the launch's own tails (include/damar_hip.h).  Every subject block has its
     own Align_Spec, i.e. its own overlap buffers, written and reset per block as the reference does
     (daligner.c:1006-1021, 1051-1056). */
  for (k = 0; k < nb; k++)
    { const int same = (strcmp(afile, bfiles[k]) == 0);
      damar_match_job jobs[2];
      PBlock     *b;
      Align_Spec *sp;
      char       *d1 = NULL, *d2 = NULL;
      double      t0;
      damar_dev_index *ai = pblock_index(a, 0);
      memset(jobs, 0, sizeof(jobs));
      TIMED(10, b = same ? a : pblock_get(bfiles[k], o));
      b->busy = 1;
      TIMED(8, sp = New_Align_Spec(o->ecorr, o->spacing, a->blk.freq, o->nthreads, SYMMETRIC, o->only_id, o->notrace, 1);
               plan_keep_spec(sp));
      if (!same && SYMMETRIC)
        TIMED(11, make_subdir(&b->blk, o->runid));
      jobs[0].ablock = jobs[1].ablock = &a->blk;
      jobs[0].aidx = jobs[1].aidx = ai;
      jobs[0].bblock = &b->blk;   jobs[0].bidx = same ? ai : pblock_index(b, 0);
      jobs[1].bblock = &b->cblk;  jobs[1].bidx = pblock_index(b, 1);
      jobs[0].self = jobs[1].self = same;
      jobs[0].comp = 0;  jobs[1].comp = 1;
      jobs[0].spec = jobs[1].spec = sp;
      t0 = wall_ms();
      damar_match_batch(jobs, 2);
      stats_take(DAMAR_T_MERGE, DAMAR_T_REPORT);
      S_seeds += jobs[0].counts[0] + jobs[1].counts[0];
      { int64 c[8];
        damar_last_counters(c);
        S_work += c[1];                               /* read pairs that passed the screen of the run heads */
        S_resort += c[7];                             /* comparisons whose seeds were sorted over all the bits after all */
      }
      S_pairs += 1;
      P_ms[2] += wall_ms() - t0;
      if (a->blk.part > 0) d1 = out_dir(o->runid, a->blk.part);
      if (same)
        { TIMED(4, damar_write_overlaps(sp, d1, NULL, aroot, aroot, a->blk.ufirst + a->blk.nreads - 1)); }
      else
        { char *broot = damar_root(bfiles[k], ".db");
          const int last_read = (b->blk.part < a->blk.part) ? b->blk.ufirst + b->blk.nreads - 1
                                                            : a->blk.ufirst + a->blk.nreads - 1;
          if (b->blk.part > 0) d2 = out_dir(o->runid, b->blk.part);
          TIMED(4, damar_write_overlaps(sp, d1, d2, aroot, broot, last_read));
          free(broot);
          b->busy = 0;
        }
      free(d1);
      free(d2);
    }
  a->busy = 0;
  free(aroot);
}

/* the `daligner ...` lines of a plan, tokenised (comments and other commands are skipped); with mtok / mn / nm also its
   `LAmerge ...` lines (HPCdaligner.c:790-808) */
static void read_plan(const char *planfile, char ****ltok_p, int **lntok_p, int *nl_p, char ****mtok_p, int **mn_p, int *nm_p)
{ FILE  *f = (strcmp(planfile, "-") == 0) ? stdin : fopen(planfile, "r");
  char  *line = NULL;
  size_t cap = 0;
  char ***ltok = NULL, ***mtok = NULL;
  int   *lntok = NULL, *mn = NULL, nl = 0, lcap = 0, nm = 0, mcap = 0, j;
  if (f == NULL)
    { fprintf(stderr, "daligner: cannot open plan %s\n", planfile);
      exit(1);
    }
  while (getline(&line, &cap, f) > 0)
    { char *tok[4096], *sp = NULL, *t;
      int   n = 0, merge;
      for (t = strtok_r(line, " \t\r\n", &sp); t != NULL && n < 4095; t = strtok_r(NULL, " \t\r\n", &sp))
        tok[n++] = t;
      if (n == 0)
        continue;
      { const char *b0 = strrchr(tok[0], '/');
        b0 = b0 ? b0 + 1 : tok[0];
        merge = (strcmp(b0, "LAmerge") == 0);
        if (strcmp(b0, "daligner") != 0 && !(merge && mtok_p != NULL))              /* comments, other commands */
          continue;
      }
      if (merge)
        { if (nm >= mcap)
            { mcap = 2 * mcap + 64;
              mtok = (char ***) realloc(mtok, sizeof(char **) * (size_t) mcap);
              mn = (int *) realloc(mn, sizeof(int) * (size_t) mcap);
            }
          mtok[nm] = (char **) malloc(sizeof(char *) * (size_t) (n + 1));
          for (j = 0; j < n; j++)
            mtok[nm][j] = strdup(tok[j]);
          mtok[nm][n] = NULL;
          mn[nm++] = n;
          continue;
        }
      if (nl >= lcap)
        { lcap = 2 * lcap + 64;
          ltok = (char ***) realloc(ltok, sizeof(char **) * (size_t) lcap);
          lntok = (int *) realloc(lntok, sizeof(int) * (size_t) lcap);
        }
      ltok[nl] = (char **) malloc(sizeof(char *) * (size_t) (n + 1));
      for (j = 0; j < n; j++)
        ltok[nl][j] = strdup(tok[j]);
      ltok[nl][n] = NULL;
      lntok[nl++] = n;
    }
  free(line);
  if (f != stdin)
    fclose(f);
  *ltok_p = ltok;  *lntok_p = lntok;  *nl_p = nl;
  if (mtok_p != NULL)
    { *mtok_p = mtok;  *mn_p = mn;  *nm_p = nm; }
}

/* ---- long plans: more blocks than HBM holds indexes for ------------------------------------------------------
   HPCdaligner writes one line per A block with all its subject blocks (HPCdaligner.c:628-788): in that order a plan of n
   blocks touches 2 n indexes per line, and a least-recently-used cache smaller than that misses on EVERY block pair
   (config 4: 255 blocks of 78 Mbp, 0.62 GB per index, 510 indexes against the ~150 that fit beside the bases).  The
   block pairs of a plan are independent, so the lines are dealt out anew: the (A, subject) plane is cut into tiles of T x T
   blocks and a tile's pairs are run together -- T forward indexes of its A blocks, 2 T of its subject blocks -- tile rows
   in order, the subject tiles of a row back and forth so that consecutive tiles share their subject blocks.  Every
   output file is the same file; only the order in which they appear changes.  DAMAR_PLAN_TILE=<T> sets T (0: plan order). */
static int block_no(const char *name, size_t *stem);
static int64 stub_block_size(const char *blockname)          /* the block size in bases from "size = ..." of the .db stub (DBsplit -s), 0 if unknown */
{ size_t stem = 0;
  char   path[4200], ln[512];
  FILE  *f;
  int64  size = 0;
  if (block_no(blockname, &stem) == 0 || stem > 4000)
    return 0;
  snprintf(path, sizeof(path), "%.*s.db", (int) stem, blockname);
  f = fopen(path, "r");
  if (f == NULL)
    return 0;
  while (fgets(ln, sizeof(ln), f) != NULL)
    { long long v;
      if (sscanf(ln, " size = %lld", &v) == 1)
        size = (int64) v * 1000000;                     /* (DBsplit writes its -s argument: Mbp, DBsplit.c:191) */
    }
  fclose(f);
  return size;
}

/* the plan's lines dealt out by tiles of T x T block numbers; returns the new number of lines */
static int tile_plan(const Opts *base, char ****ltok_p, int **lntok_p, int nl, int T)
{ char ***ltok = *ltok_p, ***out = NULL;
  int   *lntok = *lntok_p, *outn = NULL, *first, *atile;
  int    i, j, no = 0, cap = 0, lo = 0x7fffffff, hi = 0, ntile, I, J, step;
  size_t stem;
  first = (int *) malloc(sizeof(int) * (size_t) (nl + 1));
  atile = (int *) malloc(sizeof(int) * (size_t) (nl + 1));
  for (i = 0; i < nl; i++)
    { Opts o = *base;
      o.plan = NULL;
      first[i] = parse_opts(lntok[i], ltok[i], &o);
      for (j = first[i]; j < lntok[i]; j++)
        { const int n = block_no(ltok[i][j], &stem);
          if (n == 0)
            { free(first);  free(atile);             /* a plan that names whole databases: left as it is */
              return nl;
            }
          if (n < lo) lo = n;
          if (n > hi) hi = n;
        }
    }
  ntile = (hi - lo) / T + 1;
  for (i = 0; i < nl; i++)
    atile[i] = (block_no(ltok[i][first[i]], &stem) - lo) / T;
  for (I = 0; I < ntile; I++)
    for (step = 0; step < ntile; step++)
      { J = (I & 1) ? ntile - 1 - step : step;
        for (i = 0; i < nl; i++)
          if (atile[i] == I)
            { int nb = 0;
              for (j = first[i] + 1; j < lntok[i]; j++)
                if ((block_no(ltok[i][j], &stem) - lo) / T == J)
                  nb += 1;
              if (nb == 0)
                continue;
              if (no >= cap)
                { cap = 2 * cap + 256;
                  out = (char ***) realloc(out, sizeof(char **) * (size_t) cap);
                  outn = (int *) realloc(outn, sizeof(int) * (size_t) cap);
                }
              out[no] = (char **) malloc(sizeof(char *) * (size_t) (first[i] + nb + 2));
              for (j = 0; j <= first[i]; j++)
                out[no][j] = strdup(ltok[i][j]);
              outn[no] = first[i] + 1;
              for (j = first[i] + 1; j < lntok[i]; j++)
                if ((block_no(ltok[i][j], &stem) - lo) / T == J)
                  out[no][outn[no]++] = strdup(ltok[i][j]);
              out[no][outn[no]] = NULL;
              no += 1;
            }
      }
  for (i = 0; i < nl; i++)
    { for (j = 0; j < lntok[i]; j++)
        free(ltok[i][j]);
      free(ltok[i]);
    }
  free(ltok);  free(lntok);  free(first);  free(atile);
  *ltok_p = out;  *lntok_p = outn;
  return no;
}

static int node_main(const Opts *base, const char *planfile);
static int PLAN_done_fd = -1;          /* the worker of plan mode writes one byte here when every .las is closed */

static int plan_main(const Opts *base, const char *planfile)
{ char ***ltok = NULL;                 /* the plan's daligner lines, tokenised */
  int   *lntok = NULL, nl = 0;
  int    i, j, same_masks = 1;
  pthread_t reader[8];                 /* DAMAR_PLAN_READERS of them (2): a block's read + complement + upload take ~95 ms, a plan line less */
  int nreaders = 2;
  int    have_reader = 0, device_up = 0;
  const double t_plan0 = wall_ms();

  if (base->gpus != NULL)
    return node_main(base, planfile);
  if (base->lamerge)
    { fprintf(stderr, "daligner: -L needs -G (the node scheduler runs the plan's LAmerge lines)\n");
      exit(1);
    }
  if (getenv("DAMAR_PLAN_BLOCKS") != NULL && atoi(getenv("DAMAR_PLAN_BLOCKS")) >= 2)
    PB_max = atoi(getenv("DAMAR_PLAN_BLOCKS"));
  read_plan(planfile, &ltok, &lntok, &nl, NULL, NULL, NULL);

  /* The work is done by a forked child (this process has not made a HIP call yet); the command returns as soon as the
     child says that every .las is closed.  What the child still does then -- unmapping tens of GB of HBM and the pinned
     landing buffers, tearing the HIP context down: 0.4 s in the kernel driver -- changes nothing on disk and finishes
     behind the caller's back.  A GPU command that starts within that time waits for it at the teardown gate
     (damar_gate.h) instead of colliding with it; a caller that chains GPU commands without anything in between is
     better off with DAMAR_PLAN_TIDY=1 (one process that releases everything itself: 0.85 s per command against 1.0)
     or with one plan for all of them.  A child that dies before it is done is waited for and reported. */
  if (getenv("DAMAR_PLAN_DRYRUN") != NULL)
    setenv("DAMAR_PLAN_TIDY", "1", 1);
  if (getenv("DAMAR_PLAN_TIDY") == NULL && damar_profiler_preloaded())
    { fprintf(stderr, "daligner: a profiler is preloaded (the GPU runtime is up before main): running the plan in this "
                      "process (DAMAR_PLAN_TIDY=1)\n");
      setenv("DAMAR_PLAN_TIDY", "1", 1);
    }
  if (getenv("DAMAR_PLAN_TIDY") == NULL)
    { int   pfd[2];
      pid_t pid;
      fflush(NULL);
      if (pipe(pfd) != 0 || (pid = fork()) < 0)
        { fprintf(stderr, "daligner: cannot fork the worker\n");
          exit(1);
        }
      if (pid > 0)
        { char c = 0;
          ssize_t got;
          close(pfd[1]);
          do
            got = read(pfd[0], &c, 1);
          while (got < 0 && errno == EINTR);
          if (got == 1)
            return 0;
          { int st = 0;
            waitpid(pid, &st, 0);
            if (WIFEXITED(st) && WEXITSTATUS(st) != 0)
              return WEXITSTATUS(st);
            fprintf(stderr, "daligner: the worker ended before the plan was done\n");
            return 1;
          }
        }
      close(pfd[0]);
      PLAN_done_fd = pfd[1];
    }

  /* a long plan (see tile_plan): the device first -- its HBM decides the tile -- then the lines in tile order; the
     bases of every block stay resident if they fit beside the indexes of a tile, and the reader threads bring them all */
  { int nblk = 0, want_tile = -1;
    char **seen = NULL;
    for (i = 0; i < nl; i++)
      { Opts o = *base;
        int  first;
        o.plan = NULL;
        first = parse_opts(lntok[i], ltok[i], &o);
        for (j = first; j < lntok[i]; j++)
          { int k, dup = 0;
            for (k = 0; k < nblk && !dup; k++)
              dup = (strcmp(seen[k], ltok[i][j]) == 0);
            if (!dup)
              { seen = (char **) realloc(seen, sizeof(char *) * (size_t) (nblk + 1));
                seen[nblk++] = ltok[i][j];
              }
          }
      }
    S_blocks = nblk;
    if (getenv("DAMAR_PLAN_TILE") != NULL)
      want_tile = atoi(getenv("DAMAR_PLAN_TILE"));
    if (nl > 0 && nblk > 1 && (want_tile > 0 || (want_tile < 0 && nblk > PB_max)))
      { const int64 size = stub_block_size(seen[0]);
        uint64_t fr = 0, tot = 0, budget;
        double   dev1, idx1;
        int      T;
        if (getenv("DAMAR_PLAN_DRYRUN") == NULL)
          { select_device(base);
            device_up = 1;
            mark("device selected (HIP up): long plan");
            damar_hbm_info(&fr, &tot);
          }
        budget = (uint64_t) (.55 * (double) (fr < tot ? fr : tot));
        if (getenv("DAMAR_PLAN_GB") != NULL && atof(getenv("DAMAR_PLAN_GB")) > 0)
          budget = (uint64_t) (atof(getenv("DAMAR_PLAN_GB")) * 1073741824.);
        PB_budget = budget;
        S_budget_gb = (double) budget / 1073741824.;
        dev1 = 1.5 * (double) size;  idx1 = 8. * (double) size;            /* one strand: bases + two packed copies; 8 B per k-mer */
        if (size > 0 && (double) nblk * 2. * dev1 <= .45 * (double) budget)
          { T = (int) (((double) budget - (double) nblk * 2. * dev1) / (3. * idx1));
            S_resident = 1;
            if (getenv("DAMAR_PLAN_BLOCKS") == NULL)
              PB_max = nblk;
          }
        else if (size > 0)
          { T = (int) ((double) budget / (3. * (dev1 + idx1)));
            if (getenv("DAMAR_PLAN_BLOCKS") == NULL && 2 * T + 8 > PB_max)
              PB_max = 2 * T + 8 < nblk ? 2 * T + 8 : nblk;
          }
        else
          T = 32;
        if (want_tile > 0)
          T = want_tile;
        if (T < 4)
          T = 4;
        if (T < nblk)
          { nl = tile_plan(base, &ltok, &lntok, nl, T);
            S_tile = T;
          }
        if (VERBOSE || getenv("DAMAR_CLIPROF"))
          fprintf(stderr, "daligner: long plan: %d blocks of %lld bases, budget %.1f GB, tiles of %d blocks%s, %d lines\n",
                  nblk, (long long) size, S_budget_gb, S_tile, S_resident ? ", bases of every block resident" : "", nl);
      }
    free(seen);
  }
  if (getenv("DAMAR_PLAN_DRYRUN") != NULL)         /* the lines as they would be run (tests/test_host.py): no GPU needed */
    { for (i = 0; i < nl; i++)
        { for (j = 0; j < lntok[i]; j++)
            printf("%s%s", j ? " " : "", ltok[i][j]);
          printf("\n");
        }
      return 0;
    }

  /* the block table, and what the reader thread prepares ahead: the blocks in order of first use */
  PB_cap = PB_max + LINE_B + 2;           /* a group of subject blocks and the A block can be busy beyond PB_max */
  PB = (PBlock *) calloc((size_t) PB_cap, sizeof(PBlock));
  { Opts o0 = *base, o;
    int  first0 = 0;
    for (i = 0; i < nl; i++)
      { int first;
        o = *base;  o.plan = NULL;
        first = parse_opts(lntok[i], ltok[i], &o);
        if (o.plan != NULL)
          { fprintf(stderr, "daligner: -P inside a plan\n");
            exit(1);
          }
        if (first + 2 > lntok[i])
          { fprintf(stderr, "[ERROR] - at least one target and one subject block are required\n\n");
            exit(1);
          }
        if (i == 0)
          { o0 = o;  first0 = first; }
        else if (o.mtop != o0.mtop || o.kmer != o0.kmer)
          same_masks = 0;
        else
          for (j = 0; j < o.mtop; j++)
            if (strcmp(o.mask[j], o0.mask[j]) != 0)
              same_masks = 0;
      }
    (void) first0;
    if (same_masks && nl > 0 && getenv("DAMAR_PLAN_NOREADER") == NULL)
      { for (i = 0; i < nl && PB_n < PB_max; i++)
          { int first;
            o = *base;  o.plan = NULL;
            first = parse_opts(lntok[i], ltok[i], &o);
            for (j = first; j < lntok[i] && PB_n < PB_max; j++)
              { int k, seen = 0;
                for (k = 0; k < PB_n; k++)
                  if (strcmp(PB[k].name, ltok[i][j]) == 0)
                    seen = 1;
                if (!seen)
                  { PB[PB_n].name = strdup(ltok[i][j]);
                    PB[PB_n].used = 0;                   /* not used yet: pblock_get stamps it (pblock_trim spares it) */
                    PB_n += 1;
                  }
              }
          }
        PB_ahead = PB_n;
        PB_opts = o0;
        if (getenv("DAMAR_PLAN_READERS") != NULL && atoi(getenv("DAMAR_PLAN_READERS")) >= 1)
          nreaders = atoi(getenv("DAMAR_PLAN_READERS")) > 8 ? 8 : atoi(getenv("DAMAR_PLAN_READERS"));
        for (have_reader = 0; have_reader < nreaders; have_reader++)
          if (pthread_create(&reader[have_reader], NULL, plan_reader, NULL) != 0)
            { fprintf(stderr, "daligner: cannot start the block reader thread\n");
              exit(1);
            }
      }
  }

  /* The device comes up (a few hundred ms in a cold process) while the reader thread already reads and complements the
     first blocks; it uploads once the flag below is set. */
  mark("plan parsed, reader threads started");
  if (!device_up)
    { select_device(base);
      mark("device selected (HIP up)");
    }
  if (getenv("DAMAR_PREWARM_GB") && atoi(getenv("DAMAR_PREWARM_GB")) > 0)
    { pthread_t th;                     /* grow the HBM footprint next to reading the first blocks (see damar_prewarm) */
      static int gb;
      gb = atoi(getenv("DAMAR_PREWARM_GB"));
      if (pthread_create(&th, NULL, prewarm_thread, &gb) == 0)
        pthread_detach(th);
    }
  damar_set_async(1);
  pthread_mutex_lock(&PB_mu);
  PB_dev_ready = 1;
  pthread_cond_broadcast(&PB_cv);
  pthread_mutex_unlock(&PB_mu);

  for (i = 0; i < nl; i++)
    { Opts o = *base;
      int  first;
      o.plan = NULL;
      first = parse_opts(lntok[i], ltok[i], &o);
      plan_line(&o, ltok[i][first], ltok[i] + first + 1, lntok[i] - first - 1);
      if (i == 0) mark("first plan line submitted");
    }
  mark("last plan line submitted");
  TIMED(5, damar_async_drain());
  mark("drained: every .las closed");
  S_loads = __atomic_load_n(&S_loads_, __ATOMIC_RELAXED);
  plan_stats_write(nl, 0, 1, wall_ms() - t_plan0);
  if (getenv("DAMAR_PLAN_TIDY") == NULL)
    { /* every file is closed; releasing tens of GB of HBM buffer by buffer, joining the threads and tearing the HIP context
         down costs a tenth of a second of wall time and changes nothing on disk: leave that to process exit */
      if (getenv("DAMAR_CLIPROF"))
        { fprintf(stderr, "cli: %d plan lines, %d index builds; wall ms:", nl, PB_builds);
          for (i = 0; i < 12; i++)
            fprintf(stderr, " %s=%.1f", P_name[i], P_ms[i]);
          fprintf(stderr, "\n");
        }
      fflush(NULL);
      if (PLAN_done_fd >= 0)
        { char c = 1;
          damar_gate_hold(GATE_gpu);                 /* from here on this process is only tearing down */
          if (write(PLAN_done_fd, &c, 1) != 1)
            _exit(1);
          close(PLAN_done_fd);
        }
      _exit(0);                                      /* (releasing buffer by buffer first was measured: the GPU is free no sooner, scripts/b2b.py) */
    }
  while (have_reader > 0)
    pthread_join(reader[--have_reader], NULL);
  for (i = 0; i < PS_n; i++)
    Free_Align_Spec(PS[i]);
  free(PS);
  /* Releasing the blocks and indexes one by one costs 60 - 75 ms (measured: "drained" to "released" in the DAMAR_CLIPROF
     timeline) and frees nothing sooner than the process exit that follows does (tools/startup.hip: leaving main with
     32 GB allocated or with everything freed takes the same 90 - 110 ms of teardown).  DAMAR_PLAN_RELEASE=1 releases
     anyway (leak checks, callers that keep the process). */
  if (getenv("DAMAR_PLAN_RELEASE") != NULL)
    { for (i = 0; i < PB_n; i++)
        pblock_release(PB + i);
      PB_n = 0;
      free(PB);
    }
  for (i = 0; i < nl; i++)
    { for (j = 0; j < lntok[i]; j++)
        free(ltok[i][j]);
      free(ltok[i]);
    }
  free(ltok);
  free(lntok);
  mark("blocks and indexes released (only with DAMAR_PLAN_RELEASE)");
  damar_set_async(0);
  mark("host pipeline stopped");
  if (getenv("DAMAR_CLIPROF"))
    { fprintf(stderr, "cli: %d plan lines, %d index builds; wall ms:", nl, PB_builds);
      for (i = 0; i < 12; i++)
        fprintf(stderr, " %s=%.1f", P_name[i], P_ms[i]);
      fprintf(stderr, "\n");
    }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Node mode: the plan's block pairs over the GPUs of one node (see the header).  The C counterpart of
 * damar_amd/multi.py (region_units / RegionQueue / work_units / merge_parts), with fork() and a shared page
 * of C11 atomics in place of torchrun and the store.
 * ------------------------------------------------------------------------------------------------- */

#define NODE_MAXW  64
#define NODE_GROUP 8                      /* subject blocks per unit (multi.GROUP, driver.Plan.GROUP) */

typedef struct
{ int a;                                 /* A block number */
  int b[NODE_GROUP], nb;                 /* subject blocks */
  int part, nparts;                      /* nparts > 1: one B-read range of the single pair (a, b[0]) */
  int cost;
} Unit;

typedef struct
{ atomic_int cursor[NODE_MAXW];          /* next unit of each region */
  int        first[NODE_MAXW], end[NODE_MAXW];
  int        nregions;
  struct { int units, stolen, builds, numa, cpus, tails, writers, loads;  double wall_ms;
           double phase_ms[DAMAR_T_COUNT + 1];                    /* the library's clocks (DAMAR_T_*), then the writers' */
           long long pairs, seeds, aligns, records, aligned_bp, las_bytes;
         } stat[NODE_MAXW];
  atomic_int done[NODE_MAXW];            /* the worker has closed its last file (what it does after that is teardown) */
} NodeShared;

/* "<dir/>root.N" -> N (>= 1) and the length of "<dir/>root"; 0 if the name has no block number */
static int block_no(const char *name, size_t *stem)
{ const char *dot = strrchr(name, '.');
  char *end;
  long  n;
  if (dot == NULL || dot[1] == '\0')
    return 0;
  n = strtol(dot + 1, &end, 10);
  if (*end != '\0' || n < 1 || n > 1000000)
    return 0;
  *stem = (size_t) (dot - name);
  return (int) n;
}

/* pair costs as a summed-area table: cost of a rectangle of the (a, b) plane in O(1) */
static int   NB_lo, NB_hi;               /* block numbers that occur in the plan */
static long *NS;                         /* (n+1) x (n+1) prefix sums, n = NB_hi - NB_lo + 1 */
static unsigned char *NP;                /* n x n: the pair is in the plan */
#define NIDX(a, b) ((size_t) ((a) - NB_lo) * (size_t) (NB_hi - NB_lo + 1) + (size_t) ((b) - NB_lo))

static long rect_cost(int alo, int ahi, int blo, int bhi)
{ const size_t w = (size_t) (NB_hi - NB_lo + 2);
  if (alo > ahi || blo > bhi)
    return 0;
  #define SAT(i, j) NS[(size_t) (i) * w + (size_t) (j)]
  { const int a0 = alo - NB_lo, a1 = ahi - NB_lo + 1, b0 = blo - NB_lo, b1 = bhi - NB_lo + 1;
    return SAT(a1, b1) - SAT(a0, b1) - SAT(a1, b0) + SAT(a0, b0);
  }
  #undef SAT
}

typedef struct { int alo, ahi, blo, bhi; } Region;

/* k regions of about equal cost by recursive bisection, each time across the dimension that is longer in index builds
   (an A block costs one k-mer index, a subject block two: both strands) unless its best cut is badly off balance */
static int split_region(Region r, int k, Region *out)
{ int  k1 = k / 2, t, have = 0, have_alt = 0;
  long want, best_err = 0, alt_err = 0;
  int  best_dim = 0, best_t = 0, alt_dim = 0, alt_t = 0, prefer;
  /* shrink to the rows and columns that hold pairs */
  while (r.alo < r.ahi && rect_cost(r.alo, r.alo, r.blo, r.bhi) == 0) r.alo += 1;
  while (r.ahi > r.alo && rect_cost(r.ahi, r.ahi, r.blo, r.bhi) == 0) r.ahi -= 1;
  while (r.blo < r.bhi && rect_cost(r.alo, r.ahi, r.blo, r.blo) == 0) r.blo += 1;
  while (r.bhi > r.blo && rect_cost(r.alo, r.ahi, r.bhi, r.bhi) == 0) r.bhi -= 1;
  if (k <= 1)
    { out[0] = r;
      return 1;
    }
  want = rect_cost(r.alo, r.ahi, r.blo, r.bhi) * k1 / k;
  prefer = ((r.ahi - r.alo + 1) >= 2 * (r.bhi - r.blo + 1)) ? 0 : 1;          /* 0: cut the A range, 1: the subject range */
  for (int dim = 0; dim < 2; dim++)
    for (t = (dim ? r.blo : r.alo); t < (dim ? r.bhi : r.ahi); t++)
      { const long cl = dim ? rect_cost(r.alo, r.ahi, r.blo, t) : rect_cost(r.alo, t, r.blo, r.bhi);
        const long cr = dim ? rect_cost(r.alo, r.ahi, t + 1, r.bhi) : rect_cost(t + 1, r.ahi, r.blo, r.bhi);
        const long err = labs(cl - want);
        if (cl == 0 || cr == 0)
          continue;
        if (dim == prefer && (!have || err < best_err))
          { have = 1;  best_err = err;  best_dim = dim;  best_t = t; }
        if (!have_alt || err < alt_err)
          { have_alt = 1;  alt_err = err;  alt_dim = dim;  alt_t = t; }
      }
  if (!have_alt)
    { out[0] = r;
      return 1;
    }
  if (!have || (best_err > (want > 1 ? want : 1) * 15 / 100 && alt_err < best_err))
    { best_dim = alt_dim;  best_t = alt_t; }
  { Region left = r, right = r;
    int    n;
    if (best_dim == 0) { left.ahi = best_t;  right.alo = best_t + 1; }
    else               { left.bhi = best_t;  right.blo = best_t + 1; }
    n = split_region(left, k1, out);
    return n + split_region(right, k - k1, out + n);
  }
}

static int unit_cmp(const void *x, const void *y)             /* most expensive first; then plan order */
{ const Unit *u = (const Unit *) x, *v = (const Unit *) y;
  if (u->cost != v->cost) return v->cost - u->cost;
  if (u->a != v->a) return u->a - v->a;
  return v->b[0] - u->b[0];
}

/* the units of one region: per A block its subject blocks, highest first, in groups of at most g */
static int region_units(Region r, int g, Unit *out)
{ int n = 0, a, b;
  for (a = r.alo; a <= r.ahi; a++)
    { Unit *u = NULL;
      for (b = r.bhi; b >= r.blo; b--)
        if (NP[NIDX(a, b)])
          { if (u == NULL || u->nb >= g)
              { u = out + n++;
                memset(u, 0, sizeof(*u));
                u->a = a;  u->nparts = 1;
              }
            u->b[u->nb++] = b;
            u->cost += (a == b) ? 1 : 2;
          }
    }
  qsort(out, (size_t) n, sizeof(Unit), unit_cmp);
  return n;
}

#pragma GCC diagnostic ignored "-Wformat-truncation"

int lamerge_main(int argc, char *argv[]);                     /* host/lamerge.c, linked in: LAmerge without an exec */

/* LAmerge on `argv` (NULL-terminated, argv[0] is only a name) in a forked child of this process */
static pid_t spawn_merge(char **argv)
{ pid_t pid;
  int   argc = 0;
  while (argv[argc] != NULL)
    argc += 1;
  fflush(NULL);
  pid = fork();
  if (pid == 0)
    { optind = 1;
      exit(lamerge_main(argc, argv));
    }
  return pid;
}

static void remove_tree(const char *path)                     /* rm -rf of the part directories */
{ DIR *d = opendir(path);
  struct dirent *e;
  if (d != NULL)
    { while ((e = readdir(d)) != NULL)
        if (strcmp(e->d_name, ".") != 0 && strcmp(e->d_name, "..") != 0)
          { char sub[PATH_MAX];
            struct stat sb;
            snprintf(sub, sizeof(sub), "%s/%s", path, e->d_name);
            if (lstat(sub, &sb) == 0 && S_ISDIR(sb.st_mode))
              remove_tree(sub);
            else
              unlink(sub);
          }
      closedir(d);
    }
  rmdir(path);
}

static int wait_ok(pid_t pid, const char *what)
{ int st = 0;
  if (pid < 0 || waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0)
    { fprintf(stderr, "daligner: %s failed\n", what);
      return 0;
    }
  return 1;
}

static void mkdir_p(const char *path)
{ char *p = strdup(path), *q;
  for (q = p + 1; *q; q++)
    if (*q == '/')
      { *q = '\0';
        mkdir(p, 0777);
        *q = '/';
      }
  mkdir(p, 0777);
  free(p);
}

static void part_dir(char *out, size_t cap, const char *cwd, int a, int b, int part, int nparts)
{ snprintf(out, cap, "%s/_parts/%d.%d/p%dof%d", cwd, a, b, part, nparts); }

/* one worker: its GPU, its own block table and index cache, units from the shared cursors */
/* Host placement of one worker (one process per GPU on a node of 8).  Before the library creates its
   streams, host threads and pinned landing buffers, the process is bound to the NUMA node its GPU hangs on: CPU affinity
   of this thread (the threads started later inherit it) and a preferred-node memory policy (pinned buffers are faulted in
   by the thread that allocates them).  The host tail's thread counts are cut to the worker's share of the host's cores:
   at -G8 the defaults (4 tail + 2 writer threads per worker, plus readers) would be 64 unplaced threads. */
static int cpulist_to_set(const char *text, cpu_set_t *set)
{ int n = 0;
  const char *p = text;
  CPU_ZERO(set);
  while (*p)
    { char *e;
      long a = strtol(p, &e, 10), b;
      if (e == p)
        break;
      b = a;
      if (*e == '-')
        { p = e + 1;
          b = strtol(p, &e, 10);
        }
      for (; a <= b; a++)
        if (a >= 0 && a < CPU_SETSIZE)
          { CPU_SET((int) a, set);  n += 1; }
      p = (*e == ',') ? e + 1 : e;
      if (*e != ',' )
        break;
    }
  return n;
}

/* NUMA node of HIP device `gpu` WITHOUT touching the HIP runtime (a worker binds its threads and memory policy, and
   waits at the teardown gate, before its first HIP call: the runtime's helper threads and first allocations then start
   out in the right place -- ADVICE r4).  The driver lists its nodes under /sys/class/kfd/kfd/topology/nodes: the GPUs are
   the nodes with SIMDs, in the order the runtime numbers them; `domain` and `location_id` give the PCI address, whose
   sysfs entry names the NUMA node.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES given as ordinals are mapped through.
   -1 when anything is missing: the worker then takes an even share of the job's cores, unbound. */
static int gpu_numa_node_sysfs(int gpu)
{ const char *rocr = getenv("ROCR_VISIBLE_DEVICES"), *hipv = getenv("HIP_VISIBLE_DEVICES");
  const char *vis = rocr ? rocr : hipv;
  int want = gpu, n, seen = 0;
  /* the runtime also honours CUDA_VISIBLE_DEVICES and GPU_DEVICE_ORDINAL, and with both a ROCR_ and a HIP_ list the second
     indexes into the first: rather no binding than the threads and the memory policy on another GPU's node */
  if ((getenv("CUDA_VISIBLE_DEVICES") != NULL && getenv("CUDA_VISIBLE_DEVICES")[0] != 0) ||
      (getenv("GPU_DEVICE_ORDINAL") != NULL && getenv("GPU_DEVICE_ORDINAL")[0] != 0) ||
      (rocr != NULL && rocr[0] != 0 && hipv != NULL && hipv[0] != 0))
    return -1;
  if (vis != NULL && vis[0] != 0)
    { const char *c = vis;
      int i;
      for (i = 0; i < gpu && c != NULL; i++)
        c = strchr(c, ',') ? strchr(c, ',') + 1 : NULL;
      if (c == NULL || *c < '0' || *c > '9')
        return -1;                                     /* (a list of UUIDs, or shorter than the ordinal) */
      want = atoi(c);
    }
  for (n = 0; n < 256; n++)
    { char path[128], key[64];
      unsigned long long val, simd = 0, loc = 0, dom = 0;
      FILE *f;
      snprintf(path, sizeof(path), "/sys/class/kfd/kfd/topology/nodes/%d/properties", n);
      f = fopen(path, "r");
      if (f == NULL)
        break;
      while (fscanf(f, "%63s %llu", key, &val) == 2)
        { if (strcmp(key, "simd_count") == 0) simd = val;
          else if (strcmp(key, "location_id") == 0) loc = val;
          else if (strcmp(key, "domain") == 0) dom = val;
        }
      fclose(f);
      if (simd == 0)
        continue;                                      /* a CPU node */
      if (seen++ == want)
        { int node = -1;
          snprintf(path, sizeof(path), "/sys/bus/pci/devices/%04llx:%02llx:%02llx.%llx/numa_node",
                   dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7);
          f = fopen(path, "r");
          if (f != NULL)
            { if (fscanf(f, "%d", &node) != 1)
                node = -1;
              fclose(f);
            }
          return node;
        }
    }
  return -1;
}

static void node_place(int w, int gpu, int nworkers, NodeShared *S)
{ const int node = gpu_numa_node_sysfs(gpu);
  cpu_set_t cur;
  int cores, tails, writers, bound = 0;
  if (node >= 0 && getenv("DAMAR_NODE_NOBIND") == NULL)
    { char path[96], text[4096];
      FILE *f;
      snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
      f = fopen(path, "r");
      if (f != NULL)
        { cpu_set_t set, both;
          if (fgets(text, sizeof(text), f) != NULL && cpulist_to_set(text, &set) > 0 &&
              sched_getaffinity(0, sizeof(cur), &cur) == 0)
            { CPU_AND(&both, &set, &cur);             /* (never outside what the job was given) */
              if (CPU_COUNT(&both) > 0 && sched_setaffinity(0, sizeof(both), &both) == 0)
                bound = 1;
            }
          fclose(f);
        }
      if (node < 1024)
        { unsigned long mask[16];
          memset(mask, 0, sizeof(mask));
          mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
          syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, (unsigned long) (8 * sizeof(mask)));
        }
    }
  cores = (sched_getaffinity(0, sizeof(cur), &cur) == 0) ? CPU_COUNT(&cur) : 1;
  if (!bound && nworkers > 1)                         /* no node to bind to: an even share of what the job has */
    cores = cores / nworkers > 0 ? cores / nworkers : 1;
  /* tail + writer threads within the worker's cores, leaving two for the launching thread and a reader */
  tails = (cores - 2) / 2;    if (tails > 4) tails = 4;      if (tails < 1) tails = 1;
  writers = (cores - 2) / 4;  if (writers > 2) writers = 2;  if (writers < 1) writers = 1;
  { char v[16];
    if (getenv("DAMAR_TAIL_THREADS") == NULL)
      { snprintf(v, sizeof(v), "%d", tails);  setenv("DAMAR_TAIL_THREADS", v, 1); }
    else
      tails = atoi(getenv("DAMAR_TAIL_THREADS"));
    if (getenv("DAMAR_WRITE_THREADS") == NULL)
      { snprintf(v, sizeof(v), "%d", writers);  setenv("DAMAR_WRITE_THREADS", v, 1); }
    else
      writers = atoi(getenv("DAMAR_WRITE_THREADS"));
  }
  S->stat[w].numa = bound ? node : -1;  S->stat[w].cpus = cores;  S->stat[w].tails = tails;  S->stat[w].writers = writers;
}

static int node_worker(int w, int gpu, int nworkers, int sharers, const Opts *o, const char *stem, const Unit *units,
                       NodeShared *S, const char *cwd)
{ Opts   ow = *o;
  int    r, nunits = 0, stolen = 0;
  double t0 = wall_ms();
  ow.gpu = gpu;  ow.plan = NULL;  ow.gpus = NULL;
  if (getenv("DAMAR_PLAN_BLOCKS") != NULL && atoi(getenv("DAMAR_PLAN_BLOCKS")) >= 2)
    PB_max = atoi(getenv("DAMAR_PLAN_BLOCKS"));
  PB_cap = PB_max + LINE_B + 2;
  PB = (PBlock *) calloc((size_t) PB_cap, sizeof(PBlock));
  PB_sharers = sharers;
  damar_gate_wait(gpu);                             /* before anything of the GPU runtime: not into a leaving worker's teardown */
  node_place(w, gpu, nworkers, S);                  /* this worker's threads and pinned buffers next to its GPU (no HIP call) */
  select_device(&ow);
  damar_set_async(1);
  pthread_mutex_lock(&PB_mu);
  PB_dev_ready = 1;
  pthread_mutex_unlock(&PB_mu);
  for (r = 0; r < S->nregions; r++)
    { const int p = (w + r) % S->nregions;
      for (;;)
        { const int i = atomic_fetch_add(&S->cursor[p], 1);
          const Unit *u;
          char  aname[PATH_MAX], bname[NODE_GROUP][PATH_MAX], *bfiles[NODE_GROUP], base[PATH_MAX];
          int   k;
          if (S->first[p] + i >= S->end[p])
            break;
          u = units + S->first[p] + i;
          snprintf(aname, sizeof(aname), "%s.%d", stem, u->a);
          for (k = 0; k < u->nb; k++)
            { snprintf(bname[k], sizeof(bname[k]), "%s.%d", stem, u->b[k]);
              bfiles[k] = bname[k];
            }
          if (u->nparts > 1)                       /* one B-read range of a pair: its files go under the part's directory */
            { PBlock *b = pblock_get(bfiles[0], &ow);
              const int nr = b->blk.nreads;
              part_dir(base, sizeof(base), cwd, u->a, u->b[0], u->part, u->nparts);
              mkdir_p(base);
              OUT_base = base;
              damar_set_bread_range((int) ((long) nr * u->part / u->nparts), (int) ((long) nr * (u->part + 1) / u->nparts));
            }
          plan_line(&ow, aname, bfiles, u->nb);
          if (u->nparts > 1)
            { damar_async_drain();                 /* (the write requests hold pointers into `base`) */
              damar_set_bread_range(0, -1);
              OUT_base = NULL;
            }
          nunits += 1;
          stolen += (r > 0);
        }
    }
  damar_async_drain();
  S->stat[w].units = nunits;  S->stat[w].stolen = stolen;  S->stat[w].builds = PB_builds;
  S->stat[w].wall_ms = wall_ms() - t0;
  { int64  nf = 0, nl = 0, nrec = 0, las[4];               /* what this worker's share cost (as plan_stats_write) */
    double rms = 0, tail = 0, wr = 0;
    int    q;
    damar_async_counts(&nf, &rms, &nl);
    damar_async_totals(&nrec, &tail, &wr);
    S_ms[DAMAR_T_REPORT] += rms;  S_ms[DAMAR_T_TAIL] += tail;  S_ms[DAMAR_T_D2H] += damar_async_d2h_ms();
    damar_las_totals(las);
    for (q = 0; q < DAMAR_T_COUNT; q++)
      S->stat[w].phase_ms[q] = S_ms[q];
    S->stat[w].phase_ms[DAMAR_T_COUNT] = wr;
    S->stat[w].pairs = S_pairs;  S->stat[w].seeds = S_seeds;  S->stat[w].aligns = nf;
    S->stat[w].records = las[2];  S->stat[w].aligned_bp = las[3];  S->stat[w].las_bytes = las[0];
    S->stat[w].loads = __atomic_load_n(&S_loads_, __ATOMIC_RELAXED);
  }
  damar_gate_hold(GATE_gpu);                         /* from here on this worker is only tearing down (damar_gate.h) */
  atomic_store(&S->done[w], 1);
  return 0;
}

static int node_main(const Opts *base, const char *planfile)
{ char ***ltok = NULL, ***mtok = NULL;
  int   *lntok = NULL, *mn = NULL, nl = 0, nm = 0;
  int    gpus[NODE_MAXW], W = 0, i, j;
  char  *stem = NULL;
  size_t stemlen = 0;
  Opts   o0;
  int    first0 = 0, npairs = 0, nself = 0;
  Unit  *units;
  int    nunits = 0;
  NodeShared *S;
  char   cwd[PATH_MAX];
  pid_t  pid[NODE_MAXW];
  int    ok = 1;
  double t0 = wall_ms();

  /* -G n: GPUs 0 .. n-1; -G i,j,...: those.  DAMAR_SHARE_GPU=1 puts every worker on the first (a rehearsal on one GPU) */
  { const char *g = base->gpus;
    if (strchr(g, ',') == NULL)
      { int n = atoi(g);
        for (i = 0; i < n && W < NODE_MAXW; i++)
          gpus[W++] = i;
      }
    else
      { char *copy = strdup(g), *sp = NULL, *t;
        int   ntok = 0;
        for (t = strtok_r(copy, ",", &sp); t != NULL; t = strtok_r(NULL, ",", &sp), ntok++)
          if (W < NODE_MAXW)
            gpus[W++] = atoi(t);
        free(copy);
        if (ntok > NODE_MAXW)
          W = NODE_MAXW + 1;                           /* (refused below; a list of exactly NODE_MAXW ordinals is fine) */
      }
    if (W < 1)
      { fprintf(stderr, "daligner: -G wants a number of GPUs or a list of ordinals\n");
        exit(1);
      }
    if ((strchr(g, ',') == NULL && atoi(g) > NODE_MAXW) || W > NODE_MAXW)
      { fprintf(stderr, "daligner: -G: at most %d workers\n", NODE_MAXW);
        exit(1);
      }
    if (damar_profiler_preloaded())
      { fprintf(stderr, "daligner: -G forks one worker per GPU, which a process with a preloaded profiler must not do: "
                        "profile one worker's share with -P <plan> (it then runs in-process, as with DAMAR_PLAN_TIDY=1)\n");
        exit(1);
      }
    if (getenv("DAMAR_SHARE_GPU") != NULL && atoi(getenv("DAMAR_SHARE_GPU")) > 0)
      for (i = 1; i < W; i++)
        gpus[i] = gpus[0];
  }
  if (getcwd(cwd, sizeof(cwd)) == NULL)
    { fprintf(stderr, "daligner: cannot determine the working directory\n");
      exit(1);
    }
  read_plan(planfile, &ltok, &lntok, &nl, &mtok, &mn, &nm);
  if (nl == 0)
    return 0;

  /* every line: the same options, blocks named <stem>.<number> */
  NB_lo = INT_MAX;  NB_hi = 0;
  for (i = 0; i < nl; i++)
    { Opts o = *base;
      int  first;
      o.plan = NULL;  o.gpus = NULL;
      first = parse_opts(lntok[i], ltok[i], &o);
      if (o.plan != NULL || o.gpus != NULL)
        { fprintf(stderr, "daligner: -P / -G inside a plan\n");
          exit(1);
        }
      if (first + 2 > lntok[i])
        { fprintf(stderr, "[ERROR] - at least one target and one subject block are required\n\n");
          exit(1);
        }
      if (i == 0)
        { o0 = o;  first0 = first; }
      else
        { int same = (first == first0);
          for (j = 1; same && j < first; j++)
            same = (strcmp(ltok[i][j], ltok[0][j]) == 0);
          if (!same)
            { fprintf(stderr, "daligner: -G needs a plan whose lines share their options (line %d differs)\n", i + 1);
              exit(1);
            }
        }
      for (j = first; j < lntok[i]; j++)
        { size_t sl = 0;
          const int n = block_no(ltok[i][j], &sl);
          if (n == 0 || (stem != NULL && (sl != stemlen || strncmp(stem, ltok[i][j], sl) != 0)))
            { fprintf(stderr, "daligner: -G needs blocks of ONE database named <db>.<number> (%s)\n", ltok[i][j]);
              exit(1);
            }
          if (stem == NULL)
            { stem = strndup(ltok[i][j], sl);
              stemlen = sl;
            }
          if (n < NB_lo) NB_lo = n;
          if (n > NB_hi) NB_hi = n;
        }
    }
  if (NB_hi - NB_lo >= 30000)
    { fprintf(stderr, "daligner: -G: block numbers span more than 30000\n");
      exit(1);
    }
  { const size_t n = (size_t) ((NB_hi - NB_lo + 1) & 0x7fff), w = n + 1;
    size_t a, b;
    NP = (unsigned char *) calloc(n * n, 1);
    NS = (long *) calloc(w * w, sizeof(long));
    for (i = 0; i < nl; i++)
      { size_t sl;
        const int a = block_no(ltok[i][first0], &sl);
        for (j = first0 + 1; j < lntok[i]; j++)
          { const int b = block_no(ltok[i][j], &sl);
            if (!NP[NIDX(a, b)])
              { NP[NIDX(a, b)] = 1;
                npairs += 1;
                nself += (a == b);
              }
          }
      }
    for (a = 0; a < n; a++)
      for (b = 0; b < n; b++)
        NS[(a + 1) * w + (b + 1)] = NS[a * w + (b + 1)] + NS[(a + 1) * w + b] - NS[a * w + b] +
                                    (NP[a * n + b] ? ((a == b) ? 1 : 2) : 0);
  }

  S = (NodeShared *) mmap(NULL, sizeof(NodeShared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (S == MAP_FAILED)
    { fprintf(stderr, "daligner: cannot map the shared page\n");
      exit(1);
    }
  memset(S, 0, sizeof(*S));
  units = (Unit *) calloc((size_t) npairs * NODE_MAXW + 1, sizeof(Unit));
  if (npairs >= 2 * W)
    { /* one region per worker, units of up to g subject blocks; g is halved until every worker can expect six units */
      Region regs[NODE_MAXW], all;
      int    g = NODE_GROUP, nreg;
      all.alo = all.blo = NB_lo;  all.ahi = all.bhi = NB_hi;
      nreg = split_region(all, W, regs);
      for (;;)
        { nunits = 0;
          for (i = 0; i < nreg; i++)
            { S->first[i] = nunits;
              nunits += region_units(regs[i], g, units + nunits);
              S->end[i] = nunits;
            }
          if (g == 1 || W == 1 || nunits >= 6 * W)
            break;
          g /= 2;
        }
      for (i = nreg; i < W; i++)                    /* fewer regions than workers: the others only steal */
        S->first[i] = S->end[i] = nunits;
      S->nregions = W;
    }
  else
    { /* too few pairs for the GPUs: single pairs from ONE cursor, cross pairs first, split by B-read range so that every
         worker gets about two pieces (multi.work_units) */
      const int nc_ = (2 * W + npairs - 1) / npairs, ncross = nc_ < NODE_MAXW ? nc_ : NODE_MAXW,     /* (the buffers of the
                                                                                                      split-pair path hold NODE_MAXW parts) */
                nsf = ncross / 2 > 1 ? ncross / 2 : 1;
      int a, b, p;
      for (a = NB_lo; a <= NB_hi; a++)
        for (b = NB_hi; b >= NB_lo; b--)
          if (a != b && NP[NIDX(a, b)])
            for (p = 0; p < ncross; p++)
              { Unit *u = units + nunits++;
                u->a = a;  u->b[0] = b;  u->nb = 1;  u->part = p;  u->nparts = ncross;  u->cost = 2;
              }
      for (a = NB_lo; a <= NB_hi; a++)
        if (NP[NIDX(a, a)])
          for (p = 0; p < nsf; p++)
            { Unit *u = units + nunits++;
              u->a = a;  u->b[0] = a;  u->nb = 1;  u->part = p;  u->nparts = nsf;  u->cost = 1;
            }
      S->nregions = 1;
      S->first[0] = 0;  S->end[0] = nunits;
    }
  if (o0.verbose)
    printf("daligner: %d block pairs in %d units over %d GPU worker(s)\n", npairs, nunits, W);
  if (getenv("DAMAR_NODE_DRYRUN") != NULL)        /* the work list only (tests/test_host.py): region, A block, subject blocks, part */
    { for (i = 0; i < S->nregions; i++)
        for (j = S->first[i]; j < S->end[i]; j++)
          { int k;
            printf("unit region %d a %d b", i, units[j].a);
            for (k = 0; k < units[j].nb; k++)
              printf(" %d", units[j].b[k]);
            printf(" part %d of %d cost %d\n", units[j].part, units[j].nparts, units[j].cost);
          }
      return 0;
    }
  fflush(NULL);

  /* ---- the workers: forked before this process has made a single HIP call ---- */
  for (i = 0; i < W; i++)
    { pid[i] = fork();
      if (pid[i] < 0)
        { fprintf(stderr, "daligner: fork failed\n");
          ok = 0;
          W = i;
          break;
        }
      if (pid[i] == 0)
        { int sh = 0, q;
          for (q = 0; q < W; q++)
            sh += (gpus[q] == gpus[i]);
          const int rc = node_worker(i, gpus[i], W, sh, &o0, stem, units, S, cwd);
          fflush(NULL);
          exit(rc);
        }
    }
  /* a worker is finished when it says so (its files are closed: releasing its HBM and its HIP context goes on behind the
     parent's back) or when it has exited without saying so, which is a failure */
  { int left = W;
    char gone[NODE_MAXW];
    memset(gone, 0, sizeof(gone));
    while (left > 0)
      { int progress = 0;
        for (i = 0; i < W; i++)
          if (!gone[i])
            { int st = 0;
              const pid_t r = waitpid(pid[i], &st, WNOHANG);
              if (atomic_load(&S->done[i]))
                { gone[i] = 1;  left -= 1;  progress = 1; }
              else if (r == pid[i] || r < 0)
                { if (!atomic_load(&S->done[i]))
                    { fprintf(stderr, "daligner: the worker on GPU %d failed (%s %d): its units are missing\n", gpus[i],
                              WIFSIGNALED(st) ? "signal" : "exit code", WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st));
                      ok = 0;
                    }
                  gone[i] = 1;  left -= 1;  progress = 1;
                }
            }
        if (!progress)
          usleep(500);
      }
  }
  if (!ok)
    return 1;

  /* ---- the files of split pairs: every part holds the records of its B-read range, sorted; records of one
          (aread, bread) pair never span parts, so a merge on the record order restores the unsplit pair's files ---- */
  { char  lam[] = "LAmerge", db[PATH_MAX];
    snprintf(db, sizeof(db), "%s", stem);
    for (i = 0; i < nunits; i++)
      if (units[i].nparts > 1 && units[i].part == 0)
        { const Unit *u = units + i;
          const char *sl = strrchr(stem, '/');
          const char *root = sl ? sl + 1 : stem;
          int side;
          for (side = 0; side < (u->a == u->b[0] ? 1 : 2); side++)
            { const int x = side ? u->b[0] : u->a, y = side ? u->a : u->b[0];
              char *d = damar_get_dir(o0.runid, x), rel[PATH_MAX], out[PATH_MAX];
              char *argv[NODE_MAXW + 8];
              char  srcs[NODE_MAXW][PATH_MAX];
              int   na = 0, p;
              snprintf(rel, sizeof(rel), "%s/%s.%d.%s.%d.las", d, root, x, root, y);
              snprintf(out, sizeof(out), "%s/%s", cwd, rel);
              argv[na++] = lam;  argv[na++] = db;  argv[na++] = out;
              for (p = 0; p < u->nparts; p++)
                { char pd[PATH_MAX];
                  struct stat sb;
                  part_dir(pd, sizeof(pd), cwd, u->a, u->b[0], p, u->nparts);
                  snprintf(srcs[p], sizeof(srcs[p]), "%s/%s", pd, rel);
                  if (stat(srcs[p], &sb) == 0)
                    argv[na++] = srcs[p];
                }
              argv[na] = NULL;
              if (na > 3)
                { char dd[PATH_MAX];
                  snprintf(dd, sizeof(dd), "%s/%s", cwd, d);
                  mkdir_p(dd);
                  ok = wait_ok(spawn_merge(argv), "LAmerge of the parts of a split block pair") && ok;
                }
              free(d);
            }
        }
    if (S->nregions == 1 && ok)
      { char pd[PATH_MAX];
        snprintf(pd, sizeof(pd), "%s/_parts", cwd);
        remove_tree(pd);
      }
    /* ---- -L: the plan's own LAmerge lines, the step that follows daligner in every plan, W at a time ---- */
    if (base->lamerge && ok)
      { pid_t run[NODE_MAXW];
        int   nrun = 0;
        for (i = 0; i < nm; i++)
          { mtok[i][0] = lam;
            if (nrun == W)
              { int q;
                for (q = 0; q < nrun; q++)
                  ok = wait_ok(run[q], "LAmerge") && ok;
                nrun = 0;
              }
            run[nrun++] = spawn_merge(mtok[i]);
          }
        for (i = 0; i < nrun; i++)
          ok = wait_ok(run[i], "LAmerge") && ok;
      }
  }
  if (getenv("DAMAR_PLAN_STATS") != NULL)            /* one machine-readable line per run: what every worker did */
    { const char *dst = getenv("DAMAR_PLAN_STATS");
      static const char *nm[DAMAR_T_COUNT + 1] = { "tuples", "ksort", "table", "merge", "ssort", "work", "report", "d2h", "tail", "write" };
      FILE  *f = (strcmp(dst, "-") == 0) ? stderr : fopen(dst, "w");
      double mx = 0, sum = 0;
      int    q;
      for (i = 0; i < W; i++)
        { sum += S->stat[i].wall_ms;
          if (S->stat[i].wall_ms > mx) mx = S->stat[i].wall_ms;
        }
      if (f != NULL)
        { fprintf(f, "{\"tool\": \"daligner -P -G\", \"workers\": %d, \"block_pairs\": %d, \"units\": %d, \"regions\": %d, \"wall_ms\": %.1f, "
                     "\"busy_max_over_mean\": %.4f, \"ok\": %d, \"worker\": [", W, npairs, nunits, S->nregions, wall_ms() - t0,
                  sum > 0 ? mx * W / sum : 0., ok);
          for (i = 0; i < W; i++)
            { fprintf(f, "%s{\"gpu\": %d, \"units\": %d, \"stolen\": %d, \"index_builds\": %d, \"block_loads\": %d, \"busy_ms\": %.1f, "
                         "\"numa\": %d, \"cpus\": %d, \"tail_threads\": %d, \"write_threads\": %d, \"block_pairs\": %lld, "
                         "\"seed_pairs\": %lld, \"local_alignments\": %lld, \"records\": %lld, \"aligned_bp\": %lld, \"las_bytes\": %lld, "
                         "\"phase_ms\": {", i ? ", " : "", gpus[i], S->stat[i].units, S->stat[i].stolen, S->stat[i].builds, S->stat[i].loads,
                      S->stat[i].wall_ms, S->stat[i].numa, S->stat[i].cpus, S->stat[i].tails, S->stat[i].writers, S->stat[i].pairs,
                      S->stat[i].seeds, S->stat[i].aligns, S->stat[i].records, S->stat[i].aligned_bp, S->stat[i].las_bytes);
              for (q = 0; q <= DAMAR_T_COUNT; q++)
                fprintf(f, "%s\"%s\": %.1f", q ? ", " : "", nm[q], S->stat[i].phase_ms[q]);
              fprintf(f, "}}");
            }
          fprintf(f, "]}\n");
          if (f != stderr)
            fclose(f);
        }
    }
  if (o0.verbose || getenv("DAMAR_CLIPROF"))
    { fprintf(stderr, "daligner: %d block pairs, %d units, %d GPU worker(s), %.2f s:", npairs, nunits, W, (wall_ms() - t0) * 1e-3);
      for (i = 0; i < W; i++)
        fprintf(stderr, " [gpu %d: %d units (%d stolen), %d index builds, %.2f s busy, numa %d, %d cpus, %d+%d tail/write threads]",
                gpus[i], S->stat[i].units, S->stat[i].stolen, S->stat[i].builds, S->stat[i].wall_ms * 1e-3, S->stat[i].numa,
                S->stat[i].cpus, S->stat[i].tails, S->stat[i].writers);
      { double mx = 0, sum = 0;
        for (i = 0; i < W; i++)
          { sum += S->stat[i].wall_ms;
            if (S->stat[i].wall_ms > mx) mx = S->stat[i].wall_ms;
          }
        fprintf(stderr, " busy max/mean %.3f", sum > 0 ? mx * W / sum : 0.);
      }
      fprintf(stderr, "\n");
    }
  return ok ? 0 : 1;
}

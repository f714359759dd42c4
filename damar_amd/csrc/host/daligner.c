/* daligner.c -- host driver of the MI355X overlapper: same command line, block-pair
 * loop, output directories and .las files as the reference's dalign/daligner.c:662-1077,
 * calling the three-function filter interface of libdamar_hip.so (damar_filter.h).
 * Host code stays C; every heavy step behind Sort_Kmers / Match_Filter runs on the GPU.
 *
 * The B blocks of the line are read, checked and reverse-complemented one or two ahead on a second
 * thread while the GPU works on the current one.  Rejected explicitly: -D (dynamic mask server).
 * -H is accepted and has no effect, exactly like the reference.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <sys/stat.h>
#include <pthread.h>

#include "damar_filter.h"
#include "damar_hip.h"

static void usage(void)
{ fprintf(stderr, "usage:\n");
  fprintf(stderr, "daligner [-vbAIOT] [-k<int(14)>] [-w<int(6)>] [-h<int(35)>] [-t<int>] [-M<int>] [-m<track>]+\n");
  fprintf(stderr, "         [-e<double(.70)] [-l<int(1000)>] [-s<int(100)>] [-H<int>] [-j<int>]\n");
  fprintf(stderr, "         [-r<int(1)>] [-g<gpu ordinal(0)>] <subject:db> <target:db> ...\n");
}

static void make_subdir(const HITS_DB *block, int run)      /* daligner.c:630-660 */
{ char *d = damar_get_dir(run, block->part);
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

static void check_reads(const HITS_DB *b, const char *name, int kmer)     /* daligner.c:499-504 */
{ int i;
  for (i = 0; i < b->nreads; i++)
    if (b->reads[i].rlen < kmer)
      { fprintf(stderr, "[ERROR] - daligner: Block %s contains reads < %dbp long !  Run DBsplit.\n", name, kmer);
        exit(1);
      }
}

/* DAMAR_CLIPROF=1: wall clock of the driver's phases on stderr at exit */
#include <time.h>
static double P_ms[8];
static const char *P_name[8] = { "read_block(2nd thread)", "Sort_Kmers", "Match_Filter", "complement(2nd thread)", "write_submit", "drain", "wait_for_block", "upload(2nd thread)" };
static double wall_ms(void)
{ struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
#define TIMED(slot, stmt) do { double t0_ = wall_ms(); stmt; P_ms[slot] += wall_ms() - t0_; } while (0)

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
            exit(1);
          if (damar_load_masks(&it->blk, PF.mask, PF.mtop))
            { printf("[ERROR] - Unable to load track!\n");
              exit(1);
            }
          check_reads(&it->blk, PF.names[i], PF.kmer);
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

int main(int argc, char *argv[])
{ HITS_DB ablock;
  char   *afile, *aroot;
  void   *aindex, *bindex;
  int     alen, blen;
  Align_Spec *spec;
  int     kmer = 14, hitmin = 35, binshift = 6, maxreps = 0;
  double  ecorr = .70;
  int     spacing = 100, runid = 1, notrace = 0, nthreads = 4, only_id = 0, gpu = -1;
  int     c, i;
  char   *mask[64];                     /* -m tracks, daligner.c:788-795 */
  int     mtop = 0;

  MINOVER = 1000;
  IDENTITY = 0;
  SYMMETRIC = 1;
  opterr = 0;
  while ((c = getopt(argc, argv, "vbOTAIk:w:h:t:M:e:l:s:H:D:m:r:j:g:")) != -1)
    switch (c)
    { case 'v': VERBOSE = 1; break;
      case 'T': notrace = 1; break;
      case 'I': IDENTITY = 1; break;
      case 'O': IDENTITY = 1; only_id = 1; break;
      case 'A': SYMMETRIC = 0; break;
      case 'k': kmer = atoi(optarg); break;
      case 'w': binshift = atoi(optarg); break;
      case 'h': hitmin = atoi(optarg); break;
      case 't': maxreps = atoi(optarg); break;
      case 'H': break;
      case 'e': ecorr = atof(optarg); break;
      case 'l': MINOVER = atoi(optarg); break;
      case 's': spacing = atoi(optarg); break;
      case 'j': nthreads = atoi(optarg); break;
      case 'r': runid = atoi(optarg); break;
      case 'g': gpu = atoi(optarg); break;
      case 'M':
        { int gb = atoi(optarg);
          if (gb < 0)
            fprintf(stderr, "invalid memory limit of (%d)\n", gb);
          damar_hip_init(gpu < 0 ? 0 : gpu);     /* sets MEM_PHYSICAL before we override the limit */
          MEM_LIMIT = (uint64) gb * 0x40000000ull;
          break;
        }
      case 'm':
        if (mtop >= 64)
          { fprintf(stderr, "daligner: too many -m tracks\n");
            exit(1);
          }
        mask[mtop++] = optarg;
        break;
      case 'b': BIASED = 1; break;
      case 'D':
        fprintf(stderr, "daligner: option -%c is not supported by this build\n", c);
        exit(1);
      default:
        fprintf(stderr, "Unsupported option: %s\n", argv[optind - 1]);
        usage();
        exit(1);
    }
  if (kmer < 0 || binshift < 0 || hitmin < 0 || maxreps < 0 || MINOVER < 0 || spacing < 0 || runid < 0)
    { fprintf(stderr, "daligner: negative option value\n");
      exit(1);
    }
  if (ecorr < .5 || ecorr >= 1.)
    { fprintf(stderr, "Average correlation must be in [.5,1.) (%g)\n", ecorr);
      exit(1);
    }
  if (optind + 2 > argc)
    { fprintf(stderr, "[ERROR] - at least one target and one subject block are required\n\n");
      usage();
      exit(1);
    }
  MINOVER *= 2;
  if (Set_Filter_Params(kmer, binshift, maxreps, hitmin, nthreads))
    { fprintf(stderr, "Illegal combination of filter parameters\n");
      exit(1);
    }
  if (gpu >= 0)
    damar_hip_init(gpu);

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

  if (SYMMETRIC)                                   /* daligner.c:911-946 */
    for (i = optind; i < argc; i++)
      if (strcmp(afile, argv[i]) != 0)
        { char *broot = damar_root(argv[i], ".db");
          char *ad = strrchr(aroot, '.'), *bd = strrchr(broot, '.');
          size_t la = ad ? (size_t) (ad - aroot + 1) : strlen(aroot);
          size_t lb = bd ? (size_t) (bd - broot + 1) : strlen(broot);
          if (strncmp(aroot, broot, la > lb ? la : lb) != 0)
            { if (VERBOSE)
                printf("[WARNING] - Daligner is performed on different databases (%s - %s). SYMMETRIC option is disabled!\n",
                       aroot, broot);
              SYMMETRIC = 0;
            }
          free(broot);
          if (!SYMMETRIC)
            break;
        }

  make_subdir(&ablock, runid);
  spec = New_Align_Spec(ecorr, spacing, ablock.freq, nthreads, SYMMETRIC, only_id, notrace, 1);

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
    if (gpu < 0 && getenv("DAMAR_DEVICE") != NULL)
      gpu = atoi(getenv("DAMAR_DEVICE"));
    damar_hip_init(gpu < 0 ? 0 : gpu);            /* before the second thread makes its first HIP call */
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
      for (i = 0; i < 8; i++)
        fprintf(stderr, " %s=%.1f", P_name[i], P_ms[i]);
      fprintf(stderr, "\n");
    }
  return 0;
}

/* datander.c -- host driver of the MI355X self-tandem finder: same command line and
 * tan/<block>.<block>.las output as the reference's scrub/datander.c:121-263, calling
 * Match_Self of libdamar_hip.so for every block named.  Host code stays C.
 *
 * One command over several blocks (scrub/datander.c:226-258 loops over them too) is a small pipeline: a reader thread
 * reads and unpacks block i + 1 while block i is on the GPU, and -- as `daligner -P` does -- the work runs in a child
 * forked before the first HIP call, so that the command returns when the last tan .las file is closed and the child's
 * teardown (unmapping its HBM, the HIP context) finishes behind the caller.  DAMAR_PLAN_TIDY=1, or a preloaded profiler,
 * keeps everything in one process. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/wait.h>

#include "damar_filter.h"
#include "damar_hip.h"
#include "damar_gate.h"

typedef struct
{ char   **name;
  int      n, kmer;
  HITS_DB *blk;
  damar_packed *pk;             /* pk[i].raw != NULL: block i is kept as its stretch of the .bps file, the GPU unpacks it */
  int     *ready;               /* 1: read, -1: failed */
  int      taken;               /* blocks the main thread has finished with: the reader stays two ahead */
  pthread_mutex_t mu;
  pthread_cond_t  cv;
} Reader;

static void *read_ahead(void *arg)
{ Reader *R = (Reader *) arg;
  int i;
  for (i = 0; i < R->n; i++)
    { int ok, r;
      pthread_mutex_lock(&R->mu);
      while (i >= R->taken + 2)
        pthread_cond_wait(&R->cv, &R->mu);
      pthread_mutex_unlock(&R->mu);
      if (getenv("DAMAR_DB_UNPACKED") != NULL)                  /* (test hook: unpacked on the host as until round 4) */
        ok = (damar_read_block(R->name[i], R->blk + i) == 0);
      else
        ok = (damar_read_block_packed(R->name[i], R->blk + i, R->pk + i) >= 0);
      if (ok)
        for (r = 0; r < R->blk[i].nreads; r++)
          if (R->blk[i].reads[r].rlen < R->kmer)
            { fprintf(stderr, "[ERROR] - datander: Block %s contains reads < %dbp long !  Run DBsplit.\n", R->name[i], R->kmer);
              ok = 0;
              break;
            }
      pthread_mutex_lock(&R->mu);
      R->ready[i] = ok ? 1 : -1;
      pthread_cond_broadcast(&R->cv);
      pthread_mutex_unlock(&R->mu);
      if (!ok)
        break;
    }
  return NULL;
}


int main(int argc, char *argv[])
{ int    kmer = 12, hitmin = 35, binshift = 4, spacing = 100, nthreads = 4, c, i, gpu = -1;
  double ecorr = .70;
  char  *outdir = "tan";
  struct stat st;
  int    done_fd = -1;
  Reader R;
  pthread_t th;

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

  /* the worker: a child forked before any HIP call; the command returns when it says that every file is closed */
  if (getenv("DAMAR_PLAN_TIDY") == NULL && !damar_profiler_preloaded())
    { int   pfd[2];
      pid_t pid;
      fflush(NULL);
      if (pipe(pfd) != 0 || (pid = fork()) < 0)
        { fprintf(stderr, "datander: cannot fork the worker\n");
          exit(1);
        }
      if (pid > 0)
        { char b = 0;
          ssize_t got;
          close(pfd[1]);
          do
            got = read(pfd[0], &b, 1);
          while (got < 0 && errno == EINTR);
          if (got == 1)
            return 0;
          { int ws = 0;
            waitpid(pid, &ws, 0);
            if (WIFEXITED(ws) && WEXITSTATUS(ws) != 0)
              return WEXITSTATUS(ws);
            fprintf(stderr, "datander: the worker ended before the blocks were done\n");
            return 1;
          }
        }
      close(pfd[0]);
      done_fd = pfd[1];
    }

  /* a self-comparison has few alignments in flight: two report wavefronts per SIMD need a quarter of the scratch a
     block-against-block launch maps (and a cold process pays for every GB it maps) */
  setenv("DAMAR_REPORT_WPS", "2", 0);

  R.name = argv + optind;  R.n = argc - optind;  R.kmer = kmer;
  R.blk = (HITS_DB *) calloc((size_t) R.n, sizeof(HITS_DB));
  R.pk = (damar_packed *) calloc((size_t) R.n, sizeof(damar_packed));
  R.ready = (int *) calloc((size_t) R.n, sizeof(int));
  R.taken = 0;
  pthread_mutex_init(&R.mu, NULL);
  pthread_cond_init(&R.cv, NULL);
  if (pthread_create(&th, NULL, read_ahead, &R) != 0)
    { fprintf(stderr, "datander: cannot start the reader thread\n");
      exit(1);
    }
  gpu = gpu >= 0 ? gpu : (getenv("DAMAR_DEVICE") ? atoi(getenv("DAMAR_DEVICE")) : 0);
  damar_gate_wait(gpu);                              /* not into the teardown of the command before this one (damar_gate.h) */
  damar_hip_init(gpu);                               /* beside the first read */

  for (i = 0; i < R.n; i++)
    { HITS_DB *blk = R.blk + i;
      char    *root;
      Align_Spec *spec;
      pthread_mutex_lock(&R.mu);
      while (R.ready[i] == 0)
        pthread_cond_wait(&R.cv, &R.mu);
      pthread_mutex_unlock(&R.mu);
      if (R.ready[i] < 0)
        exit(1);
      root = damar_root(R.name[i], ".db");
      spec = New_Align_Spec(ecorr, spacing, blk->freq, nthreads, 1, 0, 0, 0);
      if (R.pk[i].raw != NULL)                       /* Match_Self (scrub/tandem.c:1182) on a block that stays packed on the host */
        { damar_dev_block *dev = damar_block_upload_packed(blk, R.pk + i, 0);
          if (VERBOSE)
            printf("\nIndexing %s\n\nComparing %s to itself\n", root, root);
          damar_match_self(blk, dev, spec, NULL);
          damar_block_free(dev);
          damar_packed_forget(blk);
          damar_free_packed(R.pk + i);
        }
      else
        Match_Self(root, blk, spec);
      Write_Overlap_Buffer(spec, outdir, outdir, root, root, blk->ufirst + blk->nreads - 1);
      Reset_Overlap_Buffer(spec);
      Free_Align_Spec(spec);
      free(root);
      damar_close_block(blk);
      pthread_mutex_lock(&R.mu);
      R.taken = i + 1;
      pthread_cond_broadcast(&R.cv);
      pthread_mutex_unlock(&R.mu);
    }
  pthread_join(th, NULL);
  fflush(NULL);
  if (done_fd >= 0)
    { char b = 1;
      damar_gate_hold(gpu);                          /* from here on this process is only tearing down */
      if (write(done_fd, &b, 1) != 1)
        _exit(1);
      close(done_fd);
      _exit(0);                 /* every file is closed: what is left is teardown */
    }
  return 0;
}

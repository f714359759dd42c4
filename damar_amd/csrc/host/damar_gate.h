/* damar_gate.h -- one file lock per GPU that says "a worker is tearing its GPU context down".
 *
 * `daligner -P` and `datander` return when their forked worker has closed the last output file; the worker then still
 * unmaps its HBM and the HIP context (a quarter of a second in the driver).  A command that brings the GPU up in exactly
 * that window was measured to take 1.3 - 1.6 s instead of 0.55 (scripts/b2b.py: context creation and unmapping contend in
 * the driver).  So a worker takes the lock exclusively just BEFORE it reports "done" and keeps it until the process is
 * gone, and a worker that starts waits at the gate -- after it has started reading its input, before its first HIP
 * call -- for as long as somebody holds it: the next command then pays at most what is left of the teardown, never
 * more.  Workers that are still computing do not hold the lock: commands that share a GPU on purpose are not serialised.
 * No lock file (read-only /tmp ...) or a holder that does not go away within two seconds: the gate is open. */
#ifndef DAMAR_GATE_H
#define DAMAR_GATE_H
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <time.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <string.h>

/* Where the lock files live: DAMAR_GATE_DIR if set ("" = no gate at all), else the user's runtime directory
 * ($XDG_RUNTIME_DIR, a per-user 0700 directory where nobody else can plant anything), else /tmp with the user id in the
 * file name.  The gate is per USER: the commands of one user wait for each other's teardown; another user's command
 * neither waits at this gate nor can hold it (a world-writable lock file in /tmp would let any local user do both, and
 * would be refused by fs.protected_regular on most hosts anyway -- ADVICE r4). */
static int damar_gate_open(int gpu)
{ char path[512];
  const char *dir = getenv("DAMAR_GATE_DIR");
  struct stat st;
  int fd;
  if (dir != NULL && dir[0] == '\0')
    return -1;                                       /* DAMAR_GATE_DIR="" : no gate */
  if (dir == NULL)
    { dir = getenv("XDG_RUNTIME_DIR");
      if (dir != NULL && (dir[0] != '/' || stat(dir, &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid()))
        dir = NULL;
    }
  if (dir != NULL)
    snprintf(path, sizeof(path), "%s/damar_gpu%d.teardown", dir, gpu < 0 ? 0 : gpu);
  else
    snprintf(path, sizeof(path), "/tmp/damar_u%u_gpu%d.teardown", (unsigned) geteuid(), gpu < 0 ? 0 : gpu);
  fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW | O_NONBLOCK, 0600);
  if (fd < 0)
    return -1;
  /* only a plain file of our own with a single name is a lock file: not a symlink's target (O_NOFOLLOW), not a hard
     link to something else, not a FIFO or device somebody left under that name */
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_nlink != 1 || st.st_uid != geteuid())
    { close(fd);
      return -1;
    }
  return fd;
}

/* before the first HIP call of a worker */
static void damar_gate_wait(int gpu)
{ int fd = damar_gate_open(gpu), i;
  if (fd < 0)
    return;
  for (i = 0; i < 400; i++)                          /* 5 ms steps, two seconds at most */
    { struct timespec ts = { 0, 5000000 };
      if (flock(fd, LOCK_SH | LOCK_NB) == 0)
        { flock(fd, LOCK_UN);
          break;
        }
      nanosleep(&ts, NULL);
    }
  close(fd);
}

/* just before a worker tells its caller that every file is closed; the lock goes with the process */
static void damar_gate_hold(int gpu)
{ int fd = damar_gate_open(gpu), i;
  if (fd < 0)
    return;
  for (i = 0; i < 10; i++)                           /* another worker of the same GPU may be leaving just now: it holds the
                                                        gate then, and this one does not queue up behind it for long */
    { struct timespec ts = { 0, 5000000 };
      if (flock(fd, LOCK_EX | LOCK_NB) == 0)
        return;                                      /* (kept open: held until the process is gone) */
      nanosleep(&ts, NULL);
    }
  close(fd);
}

/* A process into which a profiler has been preloaded (rocprofv3 and the like) has the GPU runtime up before main(): its
   children must not use the GPU (a fork after HIP initialisation, or an exec from such a process, takes the machine
   down on some hosts).  The drivers then run in-process instead of forking a worker. */
static int damar_profiler_preloaded(void)
{ const char *t = getenv("ROCP_TOOL_LIBRARIES"), *p = getenv("LD_PRELOAD"), *h = getenv("HSA_TOOLS_LIB");
  return (t != NULL && t[0] != 0) || (h != NULL && h[0] != 0) ||
         (p != NULL && (strstr(p, "rocprof") != NULL || strstr(p, "roctracer") != NULL || strstr(p, "rocprofiler") != NULL));
}
#endif

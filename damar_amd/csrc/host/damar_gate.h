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

static int damar_gate_open(int gpu)
{ char path[512];
  const char *dir = getenv("DAMAR_GATE_DIR");
  int fd;
  if (dir != NULL && dir[0] == '\0')
    return -1;                                       /* DAMAR_GATE_DIR="" : no gate */
  snprintf(path, sizeof(path), "%s/damar_gpu%d.teardown", dir ? dir : "/tmp", gpu < 0 ? 0 : gpu);
  fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
  if (fd >= 0)
    (void) fchmod(fd, 0666);                         /* (another user's command waits at the same gate) */
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
#endif

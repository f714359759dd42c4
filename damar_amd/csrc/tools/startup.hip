/* startup.hip -- where the time of bringing the GPU runtime up goes in a cold process (the contract command of SURVEY 8(d)
 * is 0.25 s of kernels in a 0.49 s command).  Prints milliseconds since process start after every step:
 *   startup [mb]      mb = size of the "block" allocation in MB (default 512)
 * Steps: first runtime call (hipGetDeviceCount), hipSetDevice, device properties, 4 streams, events, a small and a large
 * allocation, pinned host memory, the first launch of a trivial kernel (code object load), a 128 MB upload, free, and
 * what is left until the process is gone is printed by the caller (scripts/gpu_startup.sh). */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <string.h>

static double t0;
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void lap(const char *what) { printf("%8.1f ms  %s\n", now() - t0, what); fflush(stdout); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void touch(unsigned *p) { p[threadIdx.x] = threadIdx.x; }

int main(int argc, char **argv)
{ const size_t mb = argc > 1 ? (size_t) atoi(argv[1]) : 512;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;   /* 0: free everything, return; 1: return with everything allocated; 2: _exit(0) with everything allocated */
  int n = 0;
  hipDeviceProp_t prop;
  hipStream_t st[4];
  hipEvent_t ev[32];
  void *small, *big, *pin;
  t0 = now();
  lap("main entered (after dynamic linking, static constructors, fat binary registration)");
  CK(hipGetDeviceCount(&n));                      lap("hipGetDeviceCount (runtime + driver up)");
  CK(hipSetDevice(0));                            lap("hipSetDevice");
  CK(hipGetDeviceProperties(&prop, 0));           lap("hipGetDeviceProperties");
  for (int i = 0; i < 4; i++) { CK(hipStreamCreate(&st[i])); }
  lap("4 x hipStreamCreate");
  for (int i = 0; i < 32; i++) CK(hipEventCreate(&ev[i]));
  lap("32 x hipEventCreate");
  CK(hipMalloc(&small, 1 << 20));                 lap("hipMalloc 1 MB (first allocation)");
  CK(hipMalloc(&big, mb << 20));                  lap("hipMalloc block");
  CK(hipHostMalloc(&pin, 128 << 20));             lap("hipHostMalloc 128 MB");
  hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st[0], (unsigned *) small);
  CK(hipStreamSynchronize(st[0]));                lap("first kernel launch + sync (code object load)");
  hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st[0], (unsigned *) small);
  CK(hipStreamSynchronize(st[0]));                lap("second launch + sync");
  CK(hipMemcpyAsync(big, pin, 128 << 20, hipMemcpyHostToDevice, st[1]));
  CK(hipStreamSynchronize(st[1]));                lap("128 MB upload from pinned memory");
  CK(hipMemsetAsync(big, 0, mb << 20, st[0]));
  CK(hipStreamSynchronize(st[0]));                lap("memset of the block");
  { /* a 34 MB host array (the .bps stretch of a 135 Mbp block) to the device, three ways */
    const size_t nb = (size_t) 34 << 20;
    void *h1 = NULL, *h2 = NULL, *h3 = NULL, *bounce = NULL;
    if (posix_memalign(&h1, 2 << 20, nb) || posix_memalign(&h2, 2 << 20, nb) || posix_memalign(&h3, 2 << 20, nb)) return 1;
    memset(h1, 1, nb);  memset(h2, 2, nb);  memset(h3, 3, nb);
    lap("three 34 MB host arrays touched");
    CK(hipMemcpyAsync(big, h1, nb, hipMemcpyHostToDevice, st[1]));  CK(hipStreamSynchronize(st[1]));
    lap("34 MB pageable -> device");
    CK(hipMemcpyAsync(big, h1, nb, hipMemcpyHostToDevice, st[1]));  CK(hipStreamSynchronize(st[1]));
    lap("34 MB pageable -> device, again");
    CK(hipHostRegister(h2, nb, hipHostRegisterDefault));            lap("hipHostRegister 34 MB");
    CK(hipMemcpyAsync(big, h2, nb, hipMemcpyHostToDevice, st[1]));  CK(hipStreamSynchronize(st[1]));
    lap("34 MB registered -> device");
    CK(hipHostUnregister(h2));                                      lap("hipHostUnregister");
    CK(hipHostMalloc(&bounce, nb));                                 lap("hipHostMalloc 34 MB (bounce buffer)");
    memcpy(bounce, h3, nb);                                         lap("memcpy into the bounce buffer");
    CK(hipMemcpyAsync(big, bounce, nb, hipMemcpyHostToDevice, st[1]));  CK(hipStreamSynchronize(st[1]));
    lap("34 MB bounce -> device");
  }
  { void *more[32];                                  /* what a plan holds at its end: tens of GB in a few dozen buffers */
    for (int i = 0; i < 32; i++) { CK(hipMalloc(&more[i], (size_t) 1 << 30)); CK(hipMemsetAsync(more[i], 1, (size_t) 1 << 30, st[0])); }
    CK(hipStreamSynchronize(st[0]));              lap("32 x 1 GB allocated and touched");
    if (mode == 0)
      { for (int i = 0; i < 32; i++) CK(hipFree(more[i]));
        lap("32 x hipFree");
      }
  }
  if (mode == 0)
    { CK(hipFree(big));  CK(hipFree(small));        lap("hipFree x 2");
      CK(hipHostFree(pin));                         lap("hipHostFree");
    }
  lap("leaving main");
  if (mode == 2)
    _exit(0);
  return 0;
}

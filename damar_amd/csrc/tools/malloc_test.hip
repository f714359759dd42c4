#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{ hipSetDevice(0);
  void *w; hipMalloc(&w, 1 << 20);
  hipStream_t st; hipStreamCreate(&st);
  for (int rep = 0; rep < 3; rep++)
    for (size_t gb = 1; gb <= 4; gb *= 2)
      { void *p = NULL;
        double t0 = now();
        hipError_t e = hipMalloc(&p, gb << 30);
        double t1 = now();
        hipFree(p);
        double t2 = now();
        void *q = NULL;
        e = hipMallocAsync(&q, gb << 30, st);
        hipStreamSynchronize(st);
        double t3 = now();
        hipFreeAsync(q, st);
        hipStreamSynchronize(st);
        double t4 = now();
        printf("rep %d %zu GB: hipMalloc %.1f ms, hipFree %.1f ms, hipMallocAsync+sync %.1f ms (%s), hipFreeAsync %.1f ms\n", rep, gb, t1 - t0, t2 - t1, t3 - t2, hipGetErrorName(e), t4 - t3);
      }
  /* many allocations growing the footprint */
  void *ps[24]; double t0 = now();
  for (int i = 0; i < 24; i++) hipMalloc(&ps[i], (size_t) 2 << 30);
  printf("24 x 2 GB hipMalloc: %.1f ms\n", now() - t0);
  t0 = now();
  for (int i = 0; i < 24; i++) hipFree(ps[i]);
  printf("24 x hipFree: %.1f ms\n", now() - t0);
  return 0;
}

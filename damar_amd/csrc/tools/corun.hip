/* corun.hip -- do a persistent register-heavy kernel (the shape of the report kernel: one-wavefront workgroups, ~96
 * VGPRs, resident for tens of ms at 4 or 5 wavefronts per SIMD) and short memory-bound kernels (the shape of the seed
 * stage) launched on ANOTHER stream share the machine, and at what price?  Decides whether the seed stage of the next
 * comparisons can run under the report launch of the current ones.
 *   hipcc -O3 --offload-arch=gfx950 -o corun corun.hip && ./corun
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

/* ~96 VGPRs live, dependent integer chains, no memory traffic: `trips` x 46 x 2 VALU per lane */
__global__ __launch_bounds__(64) void heavy(unsigned *out, int trips)
{ unsigned acc[46];
  for (int i = 0; i < 46; i++) acc[i] = threadIdx.x * 2654435761u + i;
  for (int t = 0; t < trips; t++)
#pragma unroll
    for (int i = 0; i < 46; i++)
      acc[i] = acc[i] * 5u + acc[(i + 7) % 46];
  unsigned s = 0;
  for (int i = 0; i < 46; i++) s ^= acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

/* streaming copy, 256 threads, few registers */
__global__ __launch_bounds__(256) void stream_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{ for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
    { uint4 v = src[i];  v.x += 1;  dst[i] = v; }
}

/* the same with ~120 VGPRs (the shape of radix_scatter: 16 items in registers per thread) */
__global__ __launch_bounds__(256) void stream_copy_fat(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{ uint4 v[24];
  for (size_t base = (size_t) blockIdx.x * 256 * 24; base < n; base += (size_t) gridDim.x * 256 * 24)
    {
#pragma unroll
      for (int r = 0; r < 24; r++) { size_t i = base + (size_t) r * 256 + threadIdx.x; v[r] = i < n ? src[i] : uint4{0, 0, 0, 0}; }
#pragma unroll
      for (int r = 0; r < 24; r++) { size_t i = base + (size_t) r * 256 + threadIdx.x; v[r].x += 1; if (i < n) dst[i] = v[r]; }
    }
}

static float run(bool do_heavy, int hwaves, int trips, int which, int reps, size_t n, const uint4 *src, uint4 *dst, unsigned *out,
                 hipStream_t sh, hipStream_t sm, float *heavy_ms)
{ hipDeviceProp_t p;  CK(hipGetDeviceProperties(&p, 0));
  hipEvent_t h0, h1, m0, m1;
  CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1)); CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1));
  CK(hipDeviceSynchronize());
  if (do_heavy)
    { CK(hipEventRecord(h0, sh));
      hipLaunchKernelGGL(heavy, dim3(p.multiProcessorCount * 4 * hwaves), dim3(64), 0, sh, out, trips);
      CK(hipEventRecord(h1, sh));
    }
  float mem_ms = 0;
  if (which >= 0)
    { CK(hipEventRecord(m0, sm));
      for (int r = 0; r < reps; r++)
        if (which == 0) hipLaunchKernelGGL(stream_copy, dim3(p.multiProcessorCount * 8), dim3(256), 0, sm, src, dst, n);
        else            hipLaunchKernelGGL(stream_copy_fat, dim3(p.multiProcessorCount * 4), dim3(256), 0, sm, src, dst, n);
      CK(hipEventRecord(m1, sm));
    }
  CK(hipDeviceSynchronize());
  if (which >= 0) CK(hipEventElapsedTime(&mem_ms, m0, m1));
  *heavy_ms = 0;
  if (do_heavy) CK(hipEventElapsedTime(heavy_ms, h0, h1));
  return mem_ms;
}

int main()
{ setvbuf(stdout, NULL, _IOLBF, 0);
  const size_t bytes = (size_t) 1 << 30, n = bytes / 16;
  uint4 *src, *dst;  unsigned *out;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, 4u << 20));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
  hipStream_t sh, sm;  CK(hipStreamCreate(&sh)); CK(hipStreamCreate(&sm));
  float hm;
  const int trips = 100000, reps = 100;
  run(true, 4, 100, 0, 2, n, src, dst, out, sh, sm, &hm);            /* warm up */
  for (int which = 0; which < 2; which++)
    { float m_alone = run(false, 0, 0, which, reps, n, src, dst, out, sh, sm, &hm);
      printf("%s alone: %d x 2 GiB in %.2f ms = %.2f TB/s\n", which ? "fat copy (120 VGPRs)" : "copy", reps, m_alone, reps * 2.0 * bytes / m_alone / 1e9);
      for (int hw = 3; hw <= 5; hw++)
        { float h_alone;  run(true, hw, trips, -1, 0, n, src, dst, out, sh, sm, &h_alone);
          float h_both;   float m_both = run(true, hw, trips, which, reps, n, src, dst, out, sh, sm, &h_both);
          printf("  heavy at %d wavefronts/SIMD: alone %.2f ms | together: heavy %.2f ms (x%.2f), %s %.2f ms (x%.2f) | one after the other %.2f ms, together %.2f ms\n",
                 hw, h_alone, h_both, h_both / h_alone, which ? "fat copy" : "copy", m_both, m_both / m_alone, h_alone + m_alone,
                 h_both > m_both ? h_both : m_both);
        }
    }
  return 0;
}

/* sortbench.hip -- the radix sort of kernels/radix_sort.hip on its own: checks every entry point against
 * std::stable_sort on the host (small n, ragged sizes, every digit count) and times the two sorts of the
 * path at their config-2 sizes (k-mer index: 135 M packed u64 on bits 32..60; seed pairs: 61 M u64 on 43 bits).
 * Built per variant: hipcc -DOS_THREADS=.. -DOS_ITEMS=.. -DOS_MINW=.. (scripts/gpu_sortbench.sh).
 *
 *   sortbench check          correctness sweep, exit 1 on the first difference
 *   sortbench time [reps]    timings; prints GB/s against the ALGORITHMIC bytes of SURVEY 8(d)
 *                            (16-byte records, one read + one write per 8-bit digit) and the bytes moved here
 */
#include "../kernels/radix_sort.hip"
#include <vector>
#include <algorithm>
#include <numeric>
#include <string.h>

static u64 rng_state = 88172645463325252ull;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <typename T> static T *dev(size_t n) { void *p; HIP_CHECK(hipMalloc(&p, sizeof(T) * (n ? n : 1))); return (T *) p; }

static int check_err(void *ws)
{ u32 e = 0;
  HIP_CHECK(hipMemcpy(&e, damar_sort_error_word(ws), 4, hipMemcpyDeviceToHost));
  return e != 0;
}

/* kind 0: u32 key + u32 val; 1: u32 keys; 2: u64 key + u32 val; 3: u64 keys on [lo,hi); 4: split */
static int check_one(int kind, u64 n, int lo, int hi, int skew)
{ std::vector<u64> hk(n);
  std::vector<u32> hv(n);
  for (u64 i = 0; i < n; i++)
    { u64 x = rnd();
      if (skew == 1) x &= 0x0303030303030303ull;            /* few distinct digits: long runs */
      if (skew == 2) x = (x & 0xff) * 0x0101010101010101ull;
      if (kind < 2) x &= 0xffffffffull;
      hk[i] = x;
      hv[i] = (u32) i;
    }
  std::vector<u32> ord(n);
  std::iota(ord.begin(), ord.end(), 0u);
  const u64 m = (hi - lo >= 64) ? ~0ull : (((1ull << (hi - lo)) - 1) << lo);
  std::stable_sort(ord.begin(), ord.end(), [&](u32 a, u32 b) { return (hk[a] & m) < (hk[b] & m); });
  void *ws = dev<char>(damar_sort_workspace_bytes(n));
  int bad = 0;
  if (kind < 2)
    { std::vector<u32> h32(n);
      for (u64 i = 0; i < n; i++) h32[i] = (u32) hk[i];
      u32 *k0 = dev<u32>(n), *k1 = dev<u32>(n), *v0 = dev<u32>(n), *v1 = dev<u32>(n);
      HIP_CHECK(hipMemcpy(k0, h32.data(), 4 * n, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(v0, hv.data(), 4 * n, hipMemcpyHostToDevice));
      int side = kind == 0 ? damar_radix_sort_u32(k0, v0, k1, v1, n, hi, ws, 0) : damar_radix_sort_keys_u32(k0, k1, n, hi, ws, 0);
      HIP_CHECK(hipDeviceSynchronize());
      std::vector<u32> ok(n), ov(n);
      HIP_CHECK(hipMemcpy(ok.data(), side ? k1 : k0, 4 * n, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(ov.data(), side ? v1 : v0, 4 * n, hipMemcpyDeviceToHost));
      for (u64 i = 0; i < n && !bad; i++)
        if (ok[i] != h32[ord[i]] || (kind == 0 && ov[i] != ord[i]))
          { fprintf(stderr, "kind %d n %llu bits %d: item %llu differs\n", kind, (unsigned long long) n, hi, (unsigned long long) i); bad = 1; }
      hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1);
    }
  else
    { u64 *k0 = dev<u64>(n), *k1 = dev<u64>(n);
      u32 *v0 = dev<u32>(n), *v1 = dev<u32>(n), *oh = dev<u32>(n), *ol = dev<u32>(n);
      HIP_CHECK(hipMemcpy(k0, hk.data(), 8 * n, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(v0, hv.data(), 4 * n, hipMemcpyHostToDevice));
      int side = 0;
      if (kind == 2) side = damar_radix_sort_u64(k0, v0, k1, v1, n, hi, ws, 0);
      else if (kind == 3) side = damar_radix_sort_keys_u64(k0, k1, n, lo, hi, ws, 0);
      else damar_radix_sort_split_u64(k0, k1, n, lo, hi, oh, ol, ws, 0);
      HIP_CHECK(hipDeviceSynchronize());
      std::vector<u64> ok(n);
      std::vector<u32> ov(n), h1(n), h2(n);
      HIP_CHECK(hipMemcpy(ok.data(), side ? k1 : k0, 8 * n, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(ov.data(), side ? v1 : v0, 4 * n, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(h1.data(), oh, 4 * n, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(h2.data(), ol, 4 * n, hipMemcpyDeviceToHost));
      for (u64 i = 0; i < n && !bad; i++)
        { const u64 want = hk[ord[i]];
          bool good = (kind == 4) ? (h1[i] == (u32) (want >> 32) && h2[i] == (u32) want)
                                  : (ok[i] == want && (kind != 2 || ov[i] == ord[i]));
          if (!good)
            { fprintf(stderr, "kind %d n %llu bits [%d,%d): item %llu differs\n", kind, (unsigned long long) n, lo, hi, (unsigned long long) i); bad = 1; }
        }
      hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(oh); hipFree(ol);
    }
  if (check_err(ws))
    { fprintf(stderr, "kind %d n %llu: look-back timeout flagged\n", kind, (unsigned long long) n); bad = 1; }
  hipFree(ws);
  return bad;
}

static int do_check(void)
{ const u64 sizes[] = { 1, 2, 63, 64, 65, 255, 4095, 4096, 4097, 8191, 8193, 100000, 1000003, 5000011 };
  int nbad = 0, ncase = 0;
  for (u64 n : sizes)
    for (int skew = 0; skew < 3; skew++)
      { if (n > 200000 && skew == 2) continue;
        nbad += check_one(0, n, 0, 28, skew);  ncase++;
        nbad += check_one(0, n, 0, 13, skew);  ncase++;
        nbad += check_one(1, n, 0, 32, skew);  ncase++;
        nbad += check_one(2, n, 0, 43, skew);  ncase++;
        nbad += check_one(2, n, 0, 64, skew);  ncase++;
        nbad += check_one(3, n, 15, 58, skew); ncase++;
        nbad += check_one(3, n, 3, 8, skew);   ncase++;
        nbad += check_one(4, n, 32, 60, skew); ncase++;
        nbad += check_one(4, n, 32, 40, skew); ncase++;
        if (nbad) { printf("FAILED after %d cases\n", ncase); return 1; }
      }
  printf("check ok: %d cases (threads %d, items %d)\n", ncase, sort_threads(), sort_threads() == 1024 ? 8 : 16);
  return 0;
}

/* time one sort shape; returns ms per sort */
static double time_sort(int kind, u64 n, int lo, int hi, int reps)
{ std::vector<u64> hk(n);
  for (u64 i = 0; i < n; i++) hk[i] = rnd();
  void *ws = dev<char>(damar_sort_workspace_bytes(n));
  u64 *src = dev<u64>(n), *k0 = dev<u64>(n), *k1 = dev<u64>(n);
  u32 *v0 = dev<u32>(n), *v1 = dev<u32>(n);
  HIP_CHECK(hipMemcpy(src, hk.data(), 8 * n, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double tot = 0;
  for (int r = 0; r < reps + 1; r++)
    { HIP_CHECK(hipMemcpyAsync(k0, src, (kind == 0 ? 4 : 8) * n, hipMemcpyDeviceToDevice, 0));
      hipEventRecord(e0, 0);
      if (kind == 0) damar_radix_sort_u32((u32 *) k0, v0, (u32 *) k1, v1, n, hi, ws, 0);
      else if (kind == 2) damar_radix_sort_u64(k0, v0, k1, v1, n, hi, ws, 0);
      else if (kind == 3) damar_radix_sort_keys_u64(k0, k1, n, lo, hi, ws, 0);
      else damar_radix_sort_split_u64(k0, k1, n, lo, hi, v0, v1, ws, 0);
      hipEventRecord(e1, 0);
      HIP_CHECK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r > 0) tot += ms;
    }
  if (check_err(ws)) { fprintf(stderr, "look-back timeout flagged\n"); exit(1); }
#ifdef OS_STATS
  { u32 e[4];
    HIP_CHECK(hipMemcpy(e, damar_sort_error_word(ws), 16, hipMemcpyDeviceToHost));
    printf("   look-back of digit 0, last sort: %.2f steps and %.2f empty polls per tile (%u tiles)\n", e[1] / (double) e[3], e[2] / (double) e[3], e[3]);
  }
#endif
  hipFree(ws); hipFree(src); hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1);
  return tot / reps;
}

static void report(const char *name, int kind, u64 n, int lo, int hi, int reps)
{ const int P = (hi - lo + 7) / 8;
  const double ms = time_sort(kind, n, lo, hi, reps);
  const double item = (kind == 0) ? 8 : (kind == 2 ? 12 : 8);                   /* bytes per item as laid out here */
  const double keyb = (kind == 0) ? 4 : 8;
  const double moved = (double) n * (keyb + 2 * item * P);
  const double algo  = (double) n * 32.0 * P;                                    /* 16 B read + 16 B written per digit */
  printf("%-34s n=%9llu P=%d  %7.3f ms   algorithmic %6.0f GB/s   moved %6.0f GB/s\n", name, (unsigned long long) n, P, ms,
         algo / ms * 1e-6, moved / ms * 1e-6);
}

int main(int argc, char **argv)
{ if (argc > 1 && strcmp(argv[1], "check") == 0)
    return do_check();
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  printf("variant: shape %d (1024: threads x 8 keys, 512 / 256: threads x 16 keys), minw %d\n", sort_threads(), OS_MINW);
  report("kmer index, packed u64 split", 4, 135000000ull, 32, 60, reps);
  report("kmer index, u32 + u32",        0, 135000000ull, 0, 28, reps);
  report("seed pairs, packed u64",       3, 61000000ull, 15, 58, reps);
  report("seed pairs, u64 + u32",        2, 61000000ull, 0, 43, reps);
  report("seed pairs, packed u64 (c4)",  3, 6000000ull, 16, 58, reps);
  return 0;
}

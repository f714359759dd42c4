/* lanewave.hip -- a MAPPING EXPERIMENT, not a product path and not bit-exact with anything: how many band cells per
 * second does the O(nd) furthest-reaching wave reach on gfx950 when every LANE owns one alignment (64 read pairs per
 * wavefront, each lane sweeping its own band serially, band state in LDS) instead of a band lying across the lanes of
 * half a wavefront (kernels/report_packed.h: 95 G cells/s at 38 % of the lanes)?  On paper that mapping costs 1.5-2
 * wavefront-instructions per cell; this tool measures its core (result, round 4: 3.5 instructions per cell, 45 G cells/s
 * at 26 % of the vector pipes -- DESIGN.md section 10): the forward wave of align.c:667-999
 * with the reference's predecessor rule, the packed 16-bases-per-step snake, the 64-bit match history with its popcount,
 * best / pruning (MAX_WAVE_LAG = 30) and a stop at the end of either sequence -- WITHOUT pebbles, trim points and
 * clipping (per-cell work of the same kind; they would add to the figure measured here).
 *
 *   lanewave [pairs] [length] [error %]      default 262144 pairs of 6000 bases at 15 %
 *
 * Checks itself against the same recurrence on the host for 256 pairs, then prints cells, time, cells/s and the lanes
 * whose band outgrew the 32-diagonal ring (they stop early and are not counted).
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#define RING   32
#define LAG    30
#define NEG    (-(1 << 28))

struct Out { int x, y, d, fail, msum; unsigned long long cells; };

/* 16 bases starting at base i of a packed sequence (2 bits per base, base j in bits 2 (j % 16) of word j / 16) */
__host__ __device__ static inline u32 window(const u32 *pk, int i)
{ const u32 lo = pk[i >> 4], hi = pk[(i >> 4) + 1];
  const int sh = 2 * (i & 15);
  return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
}

template <typename ST>
__host__ __device__ static inline void wave(const u32 *a, const u32 *b, int alen, int blen, ST &S, Out &o)
{ int low = 0, hgh = 0, best = 0, d = 0, done = 0, fail = 0, ex = 0, ey = 0, msum = 0;
  unsigned long long cells = 0;
  /* wave 0: the slide from (0, 0) */
  { int x = 0;
    for (;;)
      { int lim = alen - x < blen - x ? alen - x : blen - x;
        if (lim <= 0) { done = 1; break; }
        const u32 df = window(a, x) ^ window(b, x);
        int s = df ? (__builtin_ctz(df) >> 1) : 16;
        if (s > lim) s = lim;
        x += s;
        if (s < 16) break;
      }
    S.V(0) = 2 * x;  S.T(0) = ~0ull;
    best = 2 * x;  ex = ey = x;
  }
  while (!done)
    { d += 1;
      if (hgh - low + 3 > RING - 2) { fail = 1; break; }
      S.V(low - 1) = NEG;  S.V(low - 2) = NEG;  S.V(hgh + 1) = NEG;      /* (low - 2: what the lowest diagonal reads as its lower neighbour) */
      low -= 1;  hgh += 1;
      cells += (unsigned long long) (hgh - low + 1);
      int ap = NEG;                          /* old V[k + 1] */
      u64 tp = 0;                            /* old T[k + 1] */
      int ac = S.V(hgh);
      u64 tc = S.T(hgh);
      int nbest = best;
      for (int k = hgh; k >= low; k--)
        { const int am = S.V(k - 1);         /* still the old value: the sweep runs downwards */
          int c;  u64 t;
          if (ac < am)
            { if (am < ap) { c = ap + 1;  t = tp; }
              else         { c = am + 1;  t = S.T(k - 1); }
            }
          else
            { if (ac < ap) { c = ap + 1;  t = tp; }
              else         { c = ac + 2;  t = tc; }
            }
          t <<= 1;                           /* the edit */
          int y = (c - k) >> 1, x = y + k;
          if (c > NEG / 2 && x >= 0 && y >= 0)
            { for (;;)
                { int lim = alen - x < blen - y ? alen - x : blen - y;
                  if (lim <= 0) { done = 1;  ex = x;  ey = y;  break; }
                  const u32 df = window(a, x) ^ window(b, y);
                  int s = df ? (__builtin_ctz(df) >> 1) : 16;
                  if (s > lim) s = lim;
                  x += s;  y += s;  c += 2 * s;
                  t = (t << s) | ((1ull << s) - 1);
                  if (s < 16) break;
                }
            }
          else
            c = NEG;
          msum += __builtin_popcountll(t & 0x1fffffffffffffffull);          /* (the popcount every cell pays for the trim test) */
          if (c > nbest)
            nbest = c;
          ap = ac;  tp = tc;                 /* old values of k become "k + 1" of the next diagonal */
          ac = am;  tc = S.T(k - 1);
          S.V(k) = c;  S.T(k) = t;
        }
      best = nbest;
      while (low <= hgh && S.V(low) < best - LAG) low += 1;
      while (hgh >= low && S.V(hgh) < best - LAG) hgh -= 1;
      if (hgh < low || d > alen + blen) { fail = 2; break; }
    }
  o.x = ex;  o.y = ey;  o.d = d;  o.fail = fail;  o.cells = cells;  o.msum = msum;
}

/* band state of one lane in LDS: slot (k & 31) of a ring, lanes interleaved (bank = lane % 32 whatever k is) */
struct LdsState
{ int *v;  u32 *tlo, *thi;  int lane;
  struct RefV { int *p; __device__ operator int() const { return *p; } __device__ RefV &operator=(int x) { *p = x; return *this; } };
  struct RefT { u32 *lo, *hi;
                __device__ operator u64() const { return ((u64) *hi << 32) | *lo; }
                __device__ RefT &operator=(u64 x) { *lo = (u32) x;  *hi = (u32) (x >> 32);  return *this; } };
  __device__ RefV V(int k) { return RefV{ v + ((k & (RING - 1)) << 6) + lane }; }
  __device__ RefT T(int k) { const int i = ((k & (RING - 1)) << 6) + lane;  return RefT{ tlo + i, thi + i }; }
};

struct HostState
{ int v[RING];  u64 t[RING];
  int &V(int k) { return v[k & (RING - 1)]; }
  u64 &T(int k) { return t[k & (RING - 1)]; }
};

__global__ __launch_bounds__(64)
void lanewave_kernel(const u32 *__restrict__ apk, const u32 *__restrict__ bpk, const u32 *__restrict__ aoff,
                     const u32 *__restrict__ boff, const int *__restrict__ alen, const int *__restrict__ blen,
                     int npairs, Out *__restrict__ out)
{ __shared__ int sv[RING * 64];
  __shared__ u32 stl[RING * 64], sth[RING * 64];
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= npairs)
    return;
  LdsState S = { sv, stl, sth, (int) threadIdx.x };
  Out o;
  wave(apk + aoff[p], bpk + boff[p], alen[p], blen[p], S, o);
  out[p] = o;
}

static u64 rng = 88172645463325252ull;
static inline u32 rnd() { rng ^= rng << 13;  rng ^= rng >> 7;  rng ^= rng << 17;  return (u32) (rng >> 16); }

int main(int argc, char **argv)
{ const int npairs = argc > 1 ? atoi(argv[1]) : 262144;
  const int len    = argc > 2 ? atoi(argv[2]) : 6000;
  const double err = (argc > 3 ? atof(argv[3]) : 15.) / 100.;
  const int words  = (len + len / 4 + 64) / 16 + 3;                 /* room for insertions, and the window's second word */
  std::vector<u32> ha((size_t) npairs * words, 0), hb((size_t) npairs * words, 0), hao(npairs), hbo(npairs);
  std::vector<int> hal(npairs), hbl(npairs);
  std::vector<unsigned char> sa(len + len / 4 + 64), sb(len + len / 4 + 64);
  for (int p = 0; p < npairs; p++)
    { int na = len, nb = 0;
      for (int i = 0; i < na; i++) sa[i] = rnd() & 3;
      for (int i = 0; i < na && nb < (int) sb.size() - 2; i++)     /* B = A with substitutions, insertions, deletions */
        { const double r = (rnd() & 0xffffff) / 16777216.;
          if (r < err / 3)          sb[nb++] = (sa[i] + 1 + rnd() % 3) & 3;
          else if (r < 2 * err / 3) { sb[nb++] = rnd() & 3;  sb[nb++] = sa[i]; }
          else if (r < err)         ;
          else                      sb[nb++] = sa[i];
        }
      hao[p] = hbo[p] = (u32) ((size_t) p * words);
      hal[p] = na;  hbl[p] = nb;
      for (int i = 0; i < na; i++) ha[(size_t) p * words + (i >> 4)] |= (u32) sa[i] << (2 * (i & 15));
      for (int i = 0; i < nb; i++) hb[(size_t) p * words + (i >> 4)] |= (u32) sb[i] << (2 * (i & 15));
    }
  u32 *da, *db, *dao, *dbo;  int *dal, *dbl;  Out *dout;
  HIP_CHECK(hipMalloc(&da, ha.size() * 4));  HIP_CHECK(hipMalloc(&db, hb.size() * 4));
  HIP_CHECK(hipMalloc(&dao, npairs * 4));    HIP_CHECK(hipMalloc(&dbo, npairs * 4));
  HIP_CHECK(hipMalloc(&dal, npairs * 4));    HIP_CHECK(hipMalloc(&dbl, npairs * 4));
  HIP_CHECK(hipMalloc(&dout, sizeof(Out) * npairs));
  HIP_CHECK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(dao, hao.data(), npairs * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(dbo, hbo.data(), npairs * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(dal, hal.data(), npairs * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(dbl, hbl.data(), npairs * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));  HIP_CHECK(hipEventCreate(&e1));
  float best_ms = 1e30f;
  for (int rep = 0; rep < 3; rep++)
    { HIP_CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(lanewave_kernel, dim3((npairs + 63) / 64), dim3(64), 0, 0, da, db, dao, dbo, dal, dbl, npairs, dout);
      HIP_CHECK(hipEventRecord(e1, 0));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms;  HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best_ms) best_ms = ms;
    }
  std::vector<Out> ho(npairs);
  HIP_CHECK(hipMemcpy(ho.data(), dout, sizeof(Out) * npairs, hipMemcpyDeviceToHost));
  int bad = 0, nfail = 0;
  unsigned long long cells = 0, steps = 0;
  for (int p = 0; p < npairs; p++)
    { if (ho[p].fail) nfail += 1; else { cells += ho[p].cells;  steps += (unsigned long long) ho[p].d; }
      if (p < 256)
        { HostState S;  Out o;
          wave(ha.data() + hao[p], hb.data() + hbo[p], hal[p], hbl[p], S, o);
          if (o.x != ho[p].x || o.y != ho[p].y || o.d != ho[p].d || o.cells != ho[p].cells || o.fail != ho[p].fail || o.msum != ho[p].msum)
            { if (bad < 5) fprintf(stderr, "pair %d: device (%d,%d) d %d cells %llu fail %d, host (%d,%d) d %d cells %llu fail %d\n", p,
                                   ho[p].x, ho[p].y, ho[p].d, ho[p].cells, ho[p].fail, o.x, o.y, o.d, o.cells, o.fail);
              bad += 1;
            }
        }
    }
  printf("lanewave: %d pairs of %d bases at %.0f %% error: %llu band cells in %llu wave steps (%.1f cells per step), %.3f ms -> %.1f G cells/s; "
         "%d lanes left the ring; host check of 256 pairs: %s\n", npairs, len, 100 * err, cells, steps, steps ? (double) cells / steps : 0.,
         best_ms, cells / (best_ms * 1e6), nfail, bad ? "FAILED" : "ok");
  return bad != 0;
}

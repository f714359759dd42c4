/* shim.hip -- the C-ABI of libdamar_hip.so: the reference's three-function filter
 * interface (dalign/filter.h:64-70) implemented on one MI355X.
 *
 *   Set_Filter_Params  filter.c:171-201
 *   Sort_Kmers         filter.c:753-994   -> damar_block_upload + damar_index_build
 *   Match_Filter       filter.c:2519-2929 -> damar_match (+ ownership of btable)
 *
 * Everything compute-heavy is a HIP kernel (kernels/ *.hip); this file owns device
 * memory, orders the launches on one stream, copies the alignment records back and
 * runs the per-read-pair host tail (host/redundancy.c -> host/las.c).  There is no CPU
 * fallback: without a HIP device every entry point fails loudly.
 */
#include <algorithm>
#include <vector>
#include <deque>
#include <string>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <string.h>
#include <unistd.h>

#include "kernels/dev_common.h"
#include <ctype.h>
#include "kernels/kernels.h"

extern "C" {
#include "damar_filter.h"
#include "damar_hip.h"
#include "host/damar_host.h"
}

#define MAXGRAM 10000          /* filter.c:71 */

/* Fatal errors end the process like the reference's exit(1) (db/DB.h:84-86), but without
 * running atexit handlers: tearing the HIP runtime down from inside a failed call can hang
 * when the library is embedded (ctypes). */
static void die(void)
{ fflush(NULL);
  _exit(1);
}

/***** globals shared with the caller (filter.h:54-62 / daligner.c:131-140) *****************/

extern "C" {
int    BIASED    = 0;
int    VERBOSE   = 0;
int    MINOVER   = 2000;       /* 2 * (-l 1000), daligner.c:695, 861 */
int    HGAP_MIN  = 0;
int    SYMMETRIC = 1;
int    IDENTITY  = 0;
uint64 MEM_LIMIT    = ~0ull;   /* replaced by the physical memory size at first use */
uint64 MEM_PHYSICAL = ~0ull;
}

static int P_kmer = 14, P_hitmin = 35, P_binshift = 6, P_suppress = 0, P_nshift = 2;
static u32 P_bread_lo = 0, P_bread_hi = 0xffffffffu;      /* damar_set_bread_range */

/* Restrict the following damar_match / Match_Filter calls to the read pairs whose B read (block-local
 * index) lies in [lo, hi): the index merge and the seed sort still see the whole block pair, so the seed
 * list, the reference's thread slices (filter.c:2804-2816) and with them every record of the range are
 * exactly those of the unrestricted call.  A multi-GPU scheduler splits one block pair over several GPUs
 * this way; the parts' files merge into the unsplit files.  hi < 0 lifts the restriction. */
extern "C" void damar_set_bread_range(int lo, int hi)
{ P_bread_lo = lo < 0 ? 0u : (u32) lo;
  P_bread_hi = hi < 0 ? 0xffffffffu : (u32) hi;
}

extern "C" int Set_Filter_Params(int kmer, int binshift, int suppress, int hitmin, int nthreads)
{ if (kmer <= 1)
    return 1;
  P_kmer = kmer;  P_binshift = binshift;  P_suppress = suppress;  P_hitmin = hitmin;
  P_nshift = 0;
  while ((2 << P_nshift) <= nthreads)
    P_nshift += 1;
  return 0;
}

/***** device state ***************************************************************************/

static int          G_ready = 0;
static hipStream_t  G_st;
static hipStream_t   G_copy;                /* record downloads of the asynchronous mode */
static hipStream_t   G_rep;                 /* report launches of the asynchronous mode: beside the next comparisons' seed stages */
static hipStream_t   G_ctl;                 /* small downloads (a launch's counters) that must not queue behind anything */
static std::thread  *G_late = NULL;         /* creates G_copy, G_rep, G_ctl behind damar_hip_init (see there) */
static std::mutex    G_late_mu;
static std::atomic<int> G_late_pending(0);
static hipStream_t preload_stream(void);
static void late_streams(void)              /* before the first use of G_copy / G_rep / G_ctl */
{ if (G_late_pending.load(std::memory_order_acquire) == 0)
    return;
  std::lock_guard<std::mutex> lk(G_late_mu);
  if (G_late != NULL)
    { G_late->join();
      delete G_late;
      G_late = NULL;
      G_late_pending.store(0, std::memory_order_release);
    }
}
/* The helper threads of this library (the start-up thread above, the allocation of the second landing buffer) are inside
   HIP calls; a process that leaves through exit() -- an error path of the caller, a plan with nothing to compare -- would
   run the runtime's exit-time teardown beside them.  Registered with atexit by damar_hip_init and called by
   damar_set_async(0): they are joined first. */
static void helpers_join(void);
static hipEvent_t    G_front_done;          /* the seed stages a report launch reads from are complete */
static hipEvent_t    G_rep_done;            /* the report launch in flight is complete (DAMAR_OVERLAP=2) */
static hipEvent_t    G_report_done;
static hipEvent_t    G_set_d2h[2];
static hipEvent_t    G_last_d2h[2] = { NULL, NULL };   /* per set of record buffers: the download the next kernel that writes
                                                          into the set must not overtake */
static hipDeviceProp_t G_prop;
static hipEvent_t   G_ev[24];
static double       G_ms[DAMAR_T_COUNT];
static int          G_limit = 0;          /* the mutual-count cap the last Match_Filter used */
static double       H_ms[8];              /* host wall clock per phase (DAMAR_HOSTPROF=1 prints them at drain) */
static const char  *H_name[8] = { "index_build", "match:front", "match:order", "match:report", "match:d2h",
                                  "match:submit", "match:total", "final_drain" };
static double now_ms(void);
static int    ilog2_ceil(u64 n);

/* The filter parameters and option globals a comparison was set up under (filter.h:54-62).  A report launch may be made --
   or made again after an overflow -- during a LATER call, after the caller has moved on to other options: launches and
   host tails use this snapshot, taken when the comparison's seed stage ran, never the globals of the moment. */
struct JobParams
{ int kmer, hitmin, binshift, symmetric, minover, hgap_min; };

static JobParams params_now(void)
{ JobParams p;
  p.kmer = P_kmer;  p.hitmin = P_hitmin;  p.binshift = P_binshift;
  p.symmetric = SYMMETRIC;  p.minover = MINOVER;  p.hgap_min = HGAP_MIN;
  return p;
}


static int64        G_cnt[8];

struct Arena { char *base; size_t cap, top; };
static Arena G_work = { NULL, 0, 0 };       /* per-call temporaries, grow-only */
static Arena G_hitsJ[DAMAR_MAX_JOBS];       /* sorted seed pairs of the comparisons of the current report launch, one arena each
                                               (12 B per seed pair: what the report kernel reads) */
static Arena G_ordJ[DAMAR_MAX_JOBS];        /* their work lists and processing orders   */
static Arena G_tmp2 = { NULL, 0, 0 };       /* the same for what is left after the early cut */
static Arena G_tmp  = { NULL, 0, 0 };       /* sort ping-pong partner, flags, scan space of the seed stage: shared by the
                                               comparisons (the stream orders them), 20 B per seed pair */

static void *dmalloc(size_t n)
{ void *p = NULL;
  static int prof = -1;
  if (prof < 0)
    prof = getenv("DAMAR_HOSTPROF") != NULL;
  const double t0 = prof ? now_ms() : 0.;
  HIP_CHECK(hipMalloc(&p, n ? n : 16));
  if (prof && now_ms() - t0 > 50.)            /* the driver's occasional multi-second allocations */
    fprintf(stderr, "damar: hipMalloc of %.3f GB took %.0f ms\n", n / 1073741824., now_ms() - t0);
  return p;
}

/* The k-mer index arrays (8 B per k-mer) come and go with every index:
 * hipMalloc / hipFree of such sizes costs tens of milliseconds each and synchronises the device,
 * so released buffers are parked here and handed out again (best fit within 25 % slack).  The
 * pool is bounded; what does not fit is really freed. */
struct PoolBuf { void *p; size_t n; };
static std::vector<PoolBuf> &DP_free = *new std::vector<PoolBuf>();
static size_t DP_bytes = 0;
static const size_t DP_LIMIT = (size_t) 48 << 30;

static void *dpool_get(size_t n, size_t *got)
{ int best = -1;
  for (size_t i = 0; i < DP_free.size(); i++)
    if (DP_free[i].n >= n && DP_free[i].n <= n + (n >> 2) + (1 << 20) && (best < 0 || DP_free[i].n < DP_free[best].n))
      best = (int) i;
  if (best >= 0)
    { PoolBuf b = DP_free[best];
      DP_free.erase(DP_free.begin() + best);
      DP_bytes -= b.n;
      *got = b.n;
      return b.p;
    }
  *got = n;
  return dmalloc(n);
}

static void dpool_put(void *p, size_t n)
{ if (p == NULL)
    return;
  if (DP_bytes + n > DP_LIMIT || DP_free.size() >= 64)
    { HIP_CHECK(hipFree(p));
      return;
    }
  PoolBuf b = { p, n };
  DP_free.push_back(b);
  DP_bytes += n;
}

static void arena_reserve(Arena *a, size_t need)
{ if (need <= a->cap)
    { a->top = 0;
      return;
    }
  if (a->base)
    { HIP_CHECK(hipStreamSynchronize(G_st));
      HIP_CHECK(hipFree(a->base));
    }
  a->cap  = need + (need >> 3) + (1u << 20);
  a->base = (char *) dmalloc(a->cap);
  a->top  = 0;
}

static void *arena_take(Arena *a, size_t n)
{ size_t at = (a->top + 255) & ~(size_t) 255;
  if (at + n > a->cap)
    { fprintf(stderr, "damar: internal error, device arena overflow (%zu + %zu > %zu)\n", at, n, a->cap);
      die();
    }
  a->top = at + n;
  return a->base + at;
}

static size_t pad256(size_t n) { return (n + 511) & ~(size_t) 255; }
/* one bit per seed in whole tiles of DAMAR_SCAN_TILE (pair_heads_mark / pair_work_mark write all 64 words of a tile) */
static size_t bit_words_bytes(u64 n) { return (size_t) ((n + DAMAR_SCAN_TILE - 1) / DAMAR_SCAN_TILE) * (DAMAR_SCAN_TILE / 8); }

static int G_device = 0;      /* the device of this process; threads that touch HIP select it first */

/* NUMA node of the host memory next to GPU `device` (its PCI function's numa_node in sysfs), or -1 when the host says
   nothing.  Makes the HIP runtime start but allocates nothing: a node worker calls it BEFORE damar_hip_init, binds its
   threads and its memory policy to that node, and only then lets the library create streams and pinned landing buffers. */
extern "C" int damar_hip_numa_node(int device)
{ int ndev = 0;
  char bus[64] = "", path[160];
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return -1;
  if (hipDeviceGetPCIBusId(bus, (int) sizeof(bus), device) != hipSuccess || bus[0] == 0)
    return -1;
  for (char *c = bus; *c; c++)
    *c = (char) tolower(*c);
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  FILE *f = fopen(path, "r");
  int node = -1;
  if (f != NULL)
    { if (fscanf(f, "%d", &node) != 1)
        node = -1;
      fclose(f);
    }
  return node;
}

extern "C" int damar_hip_init(int device)
{ int ndev = 0;
  const bool iprof = getenv("DAMAR_INITPROF") != NULL;      /* where the bring-up time goes, on stderr */
  const double ip0 = now_ms();
#define INIT_MARK(what) do { if (iprof) fprintf(stderr, "init: +%.1f ms %s\n", now_ms() - ip0, what); } while (0)
  hipError_t e = hipGetDeviceCount(&ndev);
  INIT_MARK("hipGetDeviceCount");
  if (e != hipSuccess || ndev <= 0)
    { fprintf(stderr, "damar: FATAL: no HIP device visible (%s); libdamar_hip has no CPU fallback\n",
              hipGetErrorString(e));
      die();
    }
  if (device < 0 || device >= ndev)
    { fprintf(stderr, "damar: FATAL: device %d requested, %d present\n", device, ndev);
      die();
    }
  if (G_ready && device != G_device)
    { /* the streams, events and every pooled buffer of this process live on the first device: a second one
         would launch on streams of the first with memory of the second */
      fprintf(stderr, "damar: FATAL: device %d requested after the library was initialised on device %d "
                      "(one GPU per process)\n", device, G_device);
      die();
    }
  HIP_CHECK(hipSetDevice(device));
  G_device = device;
  HIP_CHECK(hipGetDeviceProperties(&G_prop, device));
  INIT_MARK("hipSetDevice + hipGetDeviceProperties");
  if (!G_ready)
    { if (strncmp(G_prop.gcnArchName, "gfx950", 6) != 0 && getenv("DAMAR_ANY_ARCH") == NULL)
        { fprintf(stderr, "damar: FATAL: device %d is %s; the kernels of this library are built for gfx950 (MI355X)\n",
                  device, G_prop.gcnArchName);
          die();
        }
      /* A stream costs 10 - 20 ms to create (a hardware queue each), a file's code object 4 - 10 ms to load, the first
         host-to-device copy of a process another 10 - 15 ms (tools/startup.hip, scripts/gpu_hiptrace.sh,
         profiles/r05_startup.txt) -- and a cold command is 0.25 s of kernels.  A start-up thread does, while this one
         creates the main stream and returns to its caller: the preload stream the block readers are waiting for and a
         first small copy on it, the seed side's code objects in the order a job needs them, the copy / report /
         control streams (first used when the first report launch is due: late_streams() joins the thread), the
         report kernel's code object. */
      G_late_pending.store(1, std::memory_order_release);
      G_late = new std::thread([device]()
        { const double t0 = now_ms();
          HIP_CHECK(hipSetDevice(device));
          { hipStream_t ps = preload_stream();
            void *d = NULL;
            static char warm[4096];
            HIP_CHECK(hipMalloc(&d, sizeof(warm)));
            HIP_CHECK(hipMemcpyAsync(d, warm, sizeof(warm), hipMemcpyHostToDevice, ps));
            HIP_CHECK(hipStreamSynchronize(ps));
            HIP_CHECK(hipFree(d));
          }
          const double t1 = now_ms();
          damar_preload_index();
          damar_preload_sort();
          damar_preload_merge();
          damar_preload_scan();
          const double t2 = now_ms();
          HIP_CHECK(hipStreamCreate(&G_copy));
          { /* EXPERIMENT (profiles/r06_sweeps.txt): the report stream on DAMAR_REPORT_CUS of the 256 CUs, so
               that the one-wavefront-per-tile kernels of the seed stream do not queue for wave slots behind a report launch.
               DAMAR_REPORT_CUPAT=1 spreads the CUs left out over the mask instead of taking them off its end. */
            const char *e = getenv("DAMAR_REPORT_CUS");
            const int ncu = G_prop.multiProcessorCount, want = e ? atoi(e) : 0;
            if (want > 0 && want < ncu)
              { const int pat = getenv("DAMAR_REPORT_CUPAT") ? atoi(getenv("DAMAR_REPORT_CUPAT")) : 0;
                std::vector<uint32_t> mask((size_t) (ncu + 31) / 32, 0u);
                for (int i = 0; i < ncu; i++)
                  { const bool on = pat ? ((long long) (i + 1) * want / ncu != (long long) i * want / ncu) : (i < want);
                    if (on)
                      mask[(size_t) i / 32] |= 1u << (i % 32);
                  }
                HIP_CHECK(hipExtStreamCreateWithCUMask(&G_rep, (uint32_t) mask.size(), mask.data()));
              }
            else
              HIP_CHECK(hipStreamCreate(&G_rep));
          }
          HIP_CHECK(hipStreamCreate(&G_ctl));
          const double t3 = now_ms();
          damar_preload_report();
          if (getenv("DAMAR_INITPROF") != NULL)
            fprintf(stderr, "init: start-up thread: preload stream + first copy %.1f ms, code objects of the seed side %.1f ms, three "
                            "streams %.1f ms, report code object %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, now_ms() - t3);
        });
      HIP_CHECK(hipStreamCreate(&G_st));
      INIT_MARK("first stream");
      { /* which seed-side kernels raise their wave priority (kernels/kernels.h SEED_PRIO): 1 = sorts, 2 = merge and
           work list, 4 = k-mer tuples.  Config-2 step on three boxes (profiles/r03_sweeps.txt): none 398 / 415 / 434 ms,
           sorts only 403 / 405, all 409 / 413 / 408: the sorts (a chain of tiles through the look-back) are the kernels
           whose time beside a report launch varies threefold from box to box, and with priority it does not. */
        const char *e = getenv("DAMAR_SEED_PRIO");
        const int mask = e ? atoi(e) : 1;
        damar_sort_set_prio(mask & 1);
        damar_merge_set_prio((mask >> 1) & 1);
        damar_index_set_prio((mask >> 2) & 1);
      }
      HIP_CHECK(hipEventCreate(&G_front_done));
      HIP_CHECK(hipEventCreateWithFlags(&G_rep_done, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&G_set_d2h[0], hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&G_set_d2h[1], hipEventDisableTiming));
      HIP_CHECK(hipEventCreate(&G_report_done));
      for (int i = 0; i < 24; i++)
        HIP_CHECK(hipEventCreate(&G_ev[i]));
      if (MEM_PHYSICAL == ~0ull)
        { uint64 phys = (uint64) sysconf(_SC_PHYS_PAGES) * (uint64) sysconf(_SC_PAGESIZE);
          if (MEM_LIMIT == ~0ull)
            MEM_LIMIT = phys;
          MEM_PHYSICAL = phys;
        }
      G_ready = 1;
      atexit(helpers_join);
      INIT_MARK("events, done");
    }
#undef INIT_MARK
  return ndev;
}

static double Q_ms[4];      /* DAMAR_HOSTPROF: scratch_prepare, scratch_outputs, outside the library between two
                               damar_match_batch calls, from a report launch to the next comparison's first kernel */
static double Q_exit = 0, Q_flush = 0, Q_seg[4], Q_entry = 0;

static void hostprof_at_exit(void)
{ fprintf(stderr, "damar host wall ms (at exit):");
  for (int i = 0; i < 8; i++)
    fprintf(stderr, " %s=%.1f", H_name[i], H_ms[i]);
  fprintf(stderr, " | scratch_prepare=%.1f scratch_outputs=%.1f\n", Q_ms[0], Q_ms[1]);
}

static void ensure_init(void)
{ if (!G_ready)
    { const char *d = getenv("DAMAR_DEVICE");
      damar_hip_init(d ? atoi(d) : 0);
      if (getenv("DAMAR_HOSTPROF"))
        atexit(hostprof_at_exit);
    }
}

static void finish_all(void);

extern "C" void damar_hip_sync(void)
{ ensure_init();
  finish_all();
  HIP_CHECK(hipDeviceSynchronize());
}

extern "C" const char *damar_hip_device_name(void)
{ static char name[400];
  ensure_init();
  snprintf(name, sizeof(name), "%s (%s, %d CUs)", G_prop.name[0] ? G_prop.name : "AMD GPU", G_prop.gcnArchName,
           G_prop.multiProcessorCount);
  return name;
}

/* A cold process pays ~25 ms per GB the first time its HBM footprint grows (the driver maps and clears new
 * memory; a buffer that is freed and allocated again is handed back at once: build/malloc_test.hip).  A command
 * that knows it will need tens of GB calls this early on a thread of its own, next to reading its input: the
 * footprint is grown and released there, and the allocations of the real work find it ready. */
extern "C" void damar_prewarm(int gigabytes)
{ ensure_init();
  HIP_CHECK(hipSetDevice(G_device));
  std::vector<void *> held;
  for (int g = 0; g < gigabytes; g += 2)
    { void *p = NULL;
      if (hipMalloc(&p, (size_t) 2 << 30) != hipSuccess)
        { (void) hipGetLastError();
          break;
        }
      held.push_back(p);
    }
  for (void *p : held)
    (void) hipFree(p);
}

static int G_debug = -1;
/* DAMAR_DEBUG=1: synchronise after every stage and name it on stderr, so that a device
 * fault can be attributed to the kernel that caused it. */
static void stage(const char *name)
{ if (G_debug < 0)
    G_debug = (getenv("DAMAR_DEBUG") != NULL);
  if (G_debug)
    { hipError_t e = hipStreamSynchronize(G_st);
      fprintf(stderr, "[damar] stage %-14s %s\n", name, hipGetErrorName(e));
      fflush(stderr);
    }
}
/* The host waits for a count from the device three times per comparison (seed pairs, work items, ...) with the stream's
   next kernels depending on it: hipStreamSynchronize sleeps until an interrupt wakes it, 30 - 45 us after the copy has
   landed (gaps on the seed stream in the kernel trace, config 4's first 24 blocks: 97 ms of 1 160).  DAMAR_SYNC_SPIN=1
   polls the stream instead: measured, config-4 lead leg 1.178 -> 1.153 s, config 2 and 3 unchanged (profiles/r06_sweeps.txt);
   not the default -- a polling thread per worker is a core less for the tails and writers of a node job. */
static void stream_wait(hipStream_t st)
{ static int spin = -1;
  if (spin < 0)
    { const char *e = getenv("DAMAR_SYNC_SPIN");
      spin = e ? atoi(e) : 0;
    }
  if (spin)
    { hipError_t q;
      while ((q = hipStreamQuery(st)) == hipErrorNotReady)
        __builtin_ia32_pause();
      HIP_CHECK(q);
      return;
    }
  HIP_CHECK(hipStreamSynchronize(st));
}

static void  tick(int i)            { HIP_CHECK(hipEventRecord(G_ev[i], G_st)); }
static void  tick_on(int i, hipStream_t st) { HIP_CHECK(hipEventRecord(G_ev[i], st)); }
static float lap(int i, int j)      { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, G_ev[i], G_ev[j])); return ms; }

extern "C" void damar_last_timings(double *ms)  { memcpy(ms, G_ms, sizeof(G_ms)); }
extern "C" void damar_last_counters(int64 *c)   { memcpy(c, G_cnt, sizeof(G_cnt)); }
extern "C" int  damar_last_limit(void)          { return G_limit; }

/***** blocks ************************************************************************************/

struct damar_dev_block
{ DevBlock d;
  float freq[4];           /* base frequencies of the block (-b) */
  int   minlen;            /* shortest read */
  u32 *pk_alloc;
  u32 *moff;
  int *mdat;
  u8  *bases_alloc;        /* d.bases = bases_alloc + 64; d.bases[-1] is the leading terminator */
  u32 *boff, *coarse;
  int  nreads;
};

static damar_dev_block *block_upload_on(const HITS_DB *block, hipStream_t st, const damar_packed *pk = NULL, int comp = 0);

extern "C" damar_dev_block *damar_block_upload(const HITS_DB *block)
{ ensure_init();
  return block_upload_on(block, G_st);
}

/* Blocks uploaded ahead of their Sort_Kmers call (the command-line driver prepares the next B block on a
   second thread): on their own stream, so the copy runs beside the kernels of the current block pair.
   Sort_Kmers takes a preloaded block by the address of its bases. */
static std::mutex PL_mu;
static std::vector<std::pair<const void *, damar_dev_block *>> PL_ready;
static hipStream_t PL_st = NULL;
static std::atomic<int> PL_state(0);         /* 0: nobody has asked for it, 1: being created (by the start-up thread of damar_hip_init or
                                                by the first caller), 2: there */
static hipStream_t preload_stream(void)
{ int s = PL_state.load(std::memory_order_acquire);
  if (s == 2)
    return PL_st;
  int zero = 0;
  if (s == 0 && PL_state.compare_exchange_strong(zero, 1))
    { HIP_CHECK(hipStreamCreateWithFlags(&PL_st, hipStreamNonBlocking));
      PL_state.store(2, std::memory_order_release);
      return PL_st;
    }
  while (PL_state.load(std::memory_order_acquire) != 2)
    usleep(50);
  return PL_st;
}

extern "C" void damar_block_preload(const HITS_DB *block)
{ ensure_init();
  HIP_CHECK(hipSetDevice(G_device));          /* the current device is a per-thread setting */
  (void) preload_stream();
  damar_dev_block *b = block_upload_on(block, PL_st);
  std::lock_guard<std::mutex> lk(PL_mu);
  PL_ready.push_back(std::make_pair((const void *) block->bases, b));
}

/* damar_block_upload on the preload stream, for a second host thread that prepares blocks ahead of the one that
 * launches the kernels (hipStreamSynchronize inside: the block is complete when this returns) */
extern "C" damar_dev_block *damar_block_upload_bg(const HITS_DB *block)
{ ensure_init();
  HIP_CHECK(hipSetDevice(G_device));          /* the current device is a per-thread setting */
  (void) preload_stream();
  return block_upload_on(block, PL_st);
}

/* A block that the host keeps packed (damar_read_block_packed: block->bases == NULL): the forward strand (comp 0) or the
 * reverse complement (comp 1) is unpacked on the device, on the preload stream like damar_block_upload_bg.  The host
 * tail needs the bases of a read pair only where two local alignments have to be bridged by a realignment (filter.c:1950
 * -2059, a few pairs in a thousand): it unpacks those reads out of the registered stretch (host_read). */
static std::mutex PK_mu;
static std::vector<std::pair<const HITS_READ *, const damar_packed *>> PK_reg;      /* by the read table: the tail works on COPIES of
                                                                                       the block records, and a block shares its
                                                                                       read table with its complement */

extern "C" damar_dev_block *damar_block_upload_packed(const HITS_DB *block, const damar_packed *pk, int comp)
{ ensure_init();
  HIP_CHECK(hipSetDevice(G_device));          /* the current device is a per-thread setting */
  (void) preload_stream();
  { std::lock_guard<std::mutex> lk(PK_mu);
    bool have = false;
    for (auto &e : PK_reg)
      if (e.first == block->reads)
        { e.second = pk;  have = true; }
    if (!have)
      PK_reg.push_back(std::make_pair((const HITS_READ *) block->reads, pk));
  }
  return block_upload_on(block, PL_st, pk, comp);
}

/* the block is about to be closed: the tail must have drained (damar_async_drain) */
extern "C" void damar_packed_forget(const HITS_DB *block)
{ std::lock_guard<std::mutex> lk(PK_mu);
  for (size_t i = 0; i < PK_reg.size(); i++)
    if (PK_reg[i].first == block->reads)
      { PK_reg.erase(PK_reg.begin() + i);
        break;
      }
}

/* read r of a host block, one byte per base with a 4 on either side: straight out of an unpacked block, or unpacked
   into buf out of a packed one (comp: the record is the block's reverse complement) */
static const char *host_read(const HITS_DB *block, int r, int comp, std::vector<char> &buf)
{ if (block->bases != NULL)
    return (const char *) block->bases + block->reads[r].boff;
  const damar_packed *pk = NULL;
  { std::lock_guard<std::mutex> lk(PK_mu);
    for (auto &e : PK_reg)
      if (e.first == block->reads)
        pk = e.second;
  }
  if (pk == NULL)
    { fprintf(stderr, "damar: internal error, a block without bases that was never uploaded packed\n");
      die();
    }
  buf.resize((size_t) block->reads[r].rlen + 2);
  damar_unpack_read(pk, block, r, comp, buf.data() + 1);
  return buf.data() + 1;
}

static damar_dev_block *take_preloaded(const HITS_DB *block)
{ std::lock_guard<std::mutex> lk(PL_mu);
  for (size_t i = 0; i < PL_ready.size(); i++)
    if (PL_ready[i].first == (const void *) block->bases)
      { damar_dev_block *b = PL_ready[i].second;
        PL_ready.erase(PL_ready.begin() + i);
        return b;
      }
  return NULL;
}

struct PackStage { u8 *raw; size_t nraw; u32 *foff; size_t nfoff; int64 holds; };
static thread_local PackStage TL_stage = { NULL, 0, NULL, 0, 0 };

/* pk != NULL: the block comes as its stretch of the .bps file and is unpacked (comp: into its reverse complement) on the
   device (kernels/kmer_index.hip unpack_bps); block->bases is not looked at then */
static damar_dev_block *block_upload_on(const HITS_DB *block, hipStream_t st, const damar_packed *pk, int comp)
{ damar_dev_block *b = (damar_dev_block *) calloc(1, sizeof(damar_dev_block));
  int    n = block->nreads;
  int64  total = block->reads[n].boff;
  if (total > 0x7fffffffll)
    { fprintf(stderr, "damar: Fatal error, DB blocks are greater than 2Gbp!\n");   /* filter.c:794-798 */
      die();
    }
  std::vector<u32> boff((size_t) n + 1);
  for (int i = 0; i <= n; i++)
    boff[i] = (u32) block->reads[i].boff;
  int minlen = 0x7fffffff;
  for (int i = 0; i < n; i++)
    minlen = std::min(minlen, (int) (boff[i + 1] - boff[i]) - 1);
  size_t nq = ((size_t) total >> COARSE_SHIFT) + 2;
  std::vector<u32> coarse(nq);
  { u32 r = 0;
    for (size_t q = 0; q < nq; q++)
      { u64 p = (u64) q << COARSE_SHIFT;
        while (r + 1 < (u32) n && (u64) boff[r + 1] <= p)
          r += 1;
        coarse[q] = r;
      }
  }
  b->bases_alloc = (u8 *) dmalloc((size_t) total + 192);      /* 64 B of padding on both sides */
  HIP_CHECK(hipMemsetAsync(b->bases_alloc, 4, (size_t) total + 192, st));
  b->boff   = (u32 *) dmalloc(sizeof(u32) * ((size_t) n + 1));
  b->coarse = (u32 *) dmalloc(sizeof(u32) * nq);
  if (pk == NULL)
    HIP_CHECK(hipMemcpyAsync(b->bases_alloc + 63, ((const char *) block->bases) - 1, (size_t) total + 1,
                             hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(b->boff, boff.data(), sizeof(u32) * ((size_t) n + 1), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(b->coarse, coarse.data(), sizeof(u32) * nq, hipMemcpyHostToDevice, st));
  if (pk != NULL)
    { /* the stretch goes up once per host thread and block: a block's two strands are unpacked out of the same copy
         (TL_stage remembers what it holds).  The staging buffers belong to the thread and are never freed: a hipFree
         waits for every stream of the device, i.e. for the kernels of the thread that computes */
      DevBlock tmp;
      memset(&tmp, 0, sizeof(tmp));
      tmp.boff = b->boff;  tmp.coarse = b->coarse;  tmp.total = (u32) total;  tmp.nreads = (u32) n;
      if (TL_stage.holds != pk->serial)
        { if (TL_stage.nraw < (size_t) pk->nraw + 64)
            { if (TL_stage.raw) HIP_CHECK(hipFree(TL_stage.raw));
              TL_stage.nraw = (size_t) pk->nraw + ((size_t) pk->nraw >> 3) + 4096;
              TL_stage.raw  = (u8 *) dmalloc(TL_stage.nraw);
            }
          if (TL_stage.nfoff < (size_t) n + 1)
            { if (TL_stage.foff) HIP_CHECK(hipFree(TL_stage.foff));
              TL_stage.nfoff = (size_t) n + ((size_t) n >> 3) + 1024;
              TL_stage.foff  = (u32 *) dmalloc(sizeof(u32) * TL_stage.nfoff);
            }
          HIP_CHECK(hipMemcpyAsync(TL_stage.raw, pk->raw, (size_t) pk->nraw, hipMemcpyHostToDevice, st));
          HIP_CHECK(hipMemcpyAsync(TL_stage.foff, pk->foff, sizeof(u32) * (size_t) n, hipMemcpyHostToDevice, st));
          TL_stage.holds = pk->serial;
        }
      damar_launch_unpack_bps(TL_stage.raw, TL_stage.foff, &tmp, comp, b->bases_alloc + 64, st);
    }
  HIP_CHECK(hipStreamSynchronize(st));
  b->d.bases  = b->bases_alloc + 64;
  b->pk_alloc = (u32 *) dmalloc(sizeof(u32) * 2 * (size_t) damar_pack_words((u32) total));      /* forward, then reversed */
  b->d.pk     = b->pk_alloc + PK_PAD;
  b->d.rbias  = (u32) (16 * damar_pack_words((u32) total) + 16 * PK_PAD);
  damar_launch_pack_bases(b->d.bases, (u32) total, b->pk_alloc + PK_PAD, st);
  HIP_CHECK(hipStreamSynchronize(st));
  b->d.boff   = b->boff;
  b->d.coarse = b->coarse;
  for (int i = 0; i < 4; i++)
    b->freq[i] = block->freq[i];
  b->minlen = minlen;
  b->d.nreads = (u32) n;
  b->d.total  = (u32) total;
  b->d.maxlen = block->maxlen;
  /* position words of this block's k-mer indexes: read << rpbits | offset in the read when both fit 32 bits (kernels.h) */
  { const int pb = std::max(1, ilog2_ceil((u64) block->maxlen + 1)), ab = std::max(1, ilog2_ceil((u64) n));
    static int rp_on = -1;
    if (rp_on < 0)
      { const char *e = getenv("DAMAR_PACK_POS");
        rp_on = e ? atoi(e) : 1;
      }
    b->d.rpbits = (rp_on && pb + ab <= 32 && pb < 32) ? pb : 0;
  }
  if (block->tracks != NULL)                  /* the merged mask track of daligner.c:442-497 */
    { const int64 *anno = (const int64 *) block->tracks->anno;
      const int   *data = (const int *) block->tracks->data;
      std::vector<u32> moff((size_t) n + 1);
      for (int i = 0; i <= n; i++)
        moff[i] = (u32) anno[i];
      b->moff = (u32 *) dmalloc(sizeof(u32) * ((size_t) n + 1));
      b->mdat = (int *) dmalloc(sizeof(int) * ((size_t) anno[n] + 2));
      HIP_CHECK(hipMemcpyAsync(b->moff, moff.data(), sizeof(u32) * ((size_t) n + 1), hipMemcpyHostToDevice, st));
      if (anno[n] > 0)
        HIP_CHECK(hipMemcpyAsync(b->mdat, data, sizeof(int) * (size_t) anno[n], hipMemcpyHostToDevice, st));
      HIP_CHECK(hipStreamSynchronize(st));
      b->d.moff = b->moff;
      b->d.mdat = b->mdat;
    }
  b->nreads   = n;
  return b;
}


extern "C" void damar_block_free(damar_dev_block *b)
{ if (b == NULL)
    return;
  finish_all();                                /* a report launch in flight (or held back) may still read the bases */
  HIP_CHECK(hipStreamSynchronize(G_st));
  HIP_CHECK(hipFree(b->bases_alloc));
  HIP_CHECK(hipFree(b->pk_alloc));
  if (b->moff) HIP_CHECK(hipFree(b->moff));
  if (b->mdat) HIP_CHECK(hipFree(b->mdat));
  HIP_CHECK(hipFree(b->boff));
  HIP_CHECK(hipFree(b->coarse));
  free(b);
}

/***** index **************************************************************************************/

struct damar_dev_index
{ size_t codes_bytes, pos_bytes;                   /* what the pool gave (dpool_get) */
  damar_dev_block *blk;
  int   own_block;
  void *codes;                 /* u32 per k-mer, u64 when wide (k > 16) */
  u32  *pos;
  u32   n;
  int   kbits, wide;
};

/* The radix sort's look-back is bounded; a timeout (never seen) raises a device word.  sort_check queues its copy
   behind the sort, sort_verify looks at the copies once the stream has been synchronised. */
static u32 *H_serr = NULL;
static int  H_nserr = 0;
static void sort_verify(void)
{ for (int i = 0; i < H_nserr; i++)
    if (H_serr[i] != 0)
      { fprintf(stderr, "damar: FATAL: the radix sort's look-back timed out\n");
        die();
      }
  H_nserr = 0;
}
static void sort_check(const void *sw)
{ if (H_serr == NULL)
    HIP_CHECK(hipHostMalloc((void **) &H_serr, 64 * sizeof(u32), hipHostMallocDefault));
  if (H_nserr == 64)
    { HIP_CHECK(hipStreamSynchronize(G_st));
      sort_verify();
    }
  HIP_CHECK(hipMemcpyAsync(&H_serr[H_nserr++], damar_sort_error_word(sw), sizeof(u32), hipMemcpyDeviceToHost, G_st));
}

/* Tile shape of the radix sort (radix_sort.hip): 1024 threads x 8 keys -- the 8192-key tile of the 512 x 16 shape in half
   the registers, so that two workgroups of 16 wavefronts fill a CU (measured alone and beside a resident report launch:
   profiles/r04_sweeps.txt; the 256-thread shape of round 3 is the slowest of the three now that a report wavefront
   holds 64 registers).  DAMAR_SORT_THREADS overrides. */
static bool overlap_on(void);
static bool corun_on(void);
static void pick_sort_shape(void)
{ static int forced = -1;
  if (forced < 0)
    { const char *e = getenv("DAMAR_SORT_THREADS");
      forced = e ? atoi(e) : 0;
    }
  damar_sort_set_threads(forced ? forced : 1024);
}

static int ilog2_ceil(u64 n)
{ int b = 0;
  while ((1ull << b) < n)
    b += 1;
  return b;
}

static int  B_have = 0, B_log[4];        /* -b log weights, fixed by the first biased Sort_Kmers of the process */

/* A new job in the same process (the in-process driver, tests): forget the -b weights, as a
 * fresh daligner process would. */
extern "C" void damar_bias_reset(void) { B_have = 0; }

static damar_dev_index *index_build_k(damar_dev_block *blk, int own_block, int *len, int K, int suppress, int use_bias)
{ ensure_init();
  pick_sort_shape();
  if (K > 32)
    { fprintf(stderr, "damar: FATAL: -k%d: a k-mer code holds at most 32 bases\n", K);
      die();
    }
  if (blk->nreads > 0 && blk->minlen < K)      /* daligner.c:499-504: the k-mer slots assume rlen >= k */
    { fprintf(stderr, "[ERROR] - daligner: Block contains reads < %dbp long !  Run DBsplit.\n", K);
      die();
    }
  int64 nk64 = (int64) blk->d.total - (int64) K * blk->nreads;
  if (nk64 <= 0)
    { *len = 0;
      if (own_block)
        damar_block_free(blk);
      return NULL;
    }
  u32 nk = (u32) nk64;
  const int kbits = 2 * K;
  const int wide = K > 16;                  /* codes as u64 (filter.c's KmerPos.code is 64 bits wide) */
  const size_t cs = wide ? sizeof(u64) : sizeof(u32);
  const bool masked = blk->d.moff != NULL;
  const bool biased = use_bias != 0;
  /* -b (filter.c:774-789): the log weights are set by the FIRST block a process sorts and then
     kept (the reference tests a static pointer), also for complemented and other blocks */
  if (biased && !B_have)
    { const double scale = -10000. / log(4.);
      for (int i = 0; i < 4; i++)
        B_log[i] = (int) ceil(scale * log((double) blk->freq[i]));
      B_have = 1;
    }
  const u32 cap = biased ? blk->d.total : nk;          /* -b can leave one k-mer per base */
  const int npass = (kbits + 7) / 8;
  damar_dev_index *ix = (damar_dev_index *) calloc(1, sizeof(damar_dev_index));
  ix->blk = blk;  ix->own_block = own_block;  ix->kbits = kbits;  ix->wide = wide;
  /* k <= 16: the sort runs on ONE u64 per k-mer, code << 32 | pos, on the code bits only (radix_sort.hip), and its
     last pass writes the two halves apart: codes[] and pos[] share one buffer of 8 bytes per k-mer.
     k > 16: u64 codes and u32 positions as key and payload. */
  if (wide)
    { ix->codes = dpool_get(cs * (size_t) cap, &ix->codes_bytes);
      ix->pos   = (u32 *) dpool_get(sizeof(u32) * (size_t) cap, &ix->pos_bytes);
    }
  else
    { ix->codes = dpool_get(2 * sizeof(u32) * (size_t) cap, &ix->codes_bytes);
      ix->pos   = (u32 *) ix->codes + cap;
    }

  size_t swb = damar_sort_workspace_bytes(cap);
  arena_reserve(&G_work, pad256(sizeof(u64) * (size_t) cap) + 3 * pad256(sizeof(u32) * (size_t) cap) + pad256(swb) +
                         pad256(damar_scan_workspace_bytes(cap)) + (1 << 16));
  void *tk = arena_take(&G_work, sizeof(u64) * (size_t) cap);
  u32 *tv = (u32 *) arena_take(&G_work, sizeof(u32) * (size_t) cap);
  void *sw = arena_take(&G_work, swb);
  u32 *keep = NULL, *off = NULL;
  void *scw = NULL;
  u64 *tot = NULL;
  if (masked || biased || suppress > 0)
    { keep = (u32 *) arena_take(&G_work, sizeof(u32) * (size_t) cap);
      off  = (u32 *) arena_take(&G_work, sizeof(u32) * (size_t) cap);
      scw  = arena_take(&G_work, damar_scan_workspace_bytes(cap));
      tot  = (u64 *) arena_take(&G_work, 64);
    }

  /* the sort ping-pongs: start on the side that makes it end in the index's own arrays (wide), resp. that makes
     its last pass READ the temporary, so that it can write the index's buffer (packed) */
  void *k0, *k1;
  u32  *v0, *v1;
  if (wide)
    { k0 = (npass & 1) ? tk : ix->codes;  k1 = (npass & 1) ? ix->codes : tk;
      v0 = (npass & 1) ? tv : ix->pos;    v1 = (npass & 1) ? ix->pos : tv;
    }
  else
    { k0 = (npass & 1) ? tk : ix->codes;  k1 = (npass & 1) ? ix->codes : tk;
      v0 = v1 = NULL;
    }

  tick(0);
  if (biased || masked)
    { /* filter.c:474-526 / 549-688 + the filler squeeze of :855-888: only k-mers inside one
         unmasked stretch (resp. the windows the -b walk yields) enter the index; dropping the rest
         before the sort leaves the same sorted list.  The candidates sit in the sort's other buffer. */
      void *k9 = k1;
      u32  *v9 = wide ? ((v0 == tv) ? ix->pos : tv) : (u32 *) k1 + cap;
      u64   kept = 0;
      u32   nin = nk;
      if (biased)
        { nin = cap;
          HIP_CHECK(hipMemsetAsync(keep, 0, sizeof(u32) * (size_t) cap, G_st));
          damar_launch_biased_tuples(&blk->d, K, B_log, k9, wide, v9, keep, G_st);
        }
      else
        { damar_launch_kmer_tuples(&blk->d, K, nk, k9, wide, v9, G_st);
          damar_launch_mask_flags(&blk->d, K, v9, nk, keep, G_st);
        }
      damar_exclusive_scan_u32(keep, off, nin, scw, tot, G_st);
      damar_launch_compact_pairs(k9, wide, v9, keep, off, nin, k0, v0, G_st);      /* v0 == NULL: packed into k0 */
      HIP_CHECK(hipMemcpyAsync(&kept, tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
      HIP_CHECK(hipStreamSynchronize(G_st));
      nk = (u32) kept;
      if (nk == 0)
        { damar_index_free(ix);
          *len = 0;
          return NULL;
        }
      if (VERBOSE && biased)
        printf("\n   Revised kmer count = %u\n", nk);
    }
  else
    damar_launch_kmer_tuples(&blk->d, K, nk, k0, wide, v0, G_st);                   /* v0 == NULL: packed */
  tick(1);
  if (wide)
    { int side = damar_radix_sort_u64((u64 *) k0, v0, (u64 *) k1, v1, nk, kbits, sw, G_st);
      if ((side ? k1 : k0) != ix->codes)
        { fprintf(stderr, "damar: internal error, sort ended on the wrong side\n");
          die();
        }
    }
  else
    damar_radix_sort_split_u64((u64 *) k0, (u64 *) k1, nk, 32, 32 + kbits, (u32 *) ix->codes, ix->pos, sw, G_st);
  sort_check(sw);
  tick(2);
  u32 n = nk;
  if (suppress > 0)                         /* filter.c:890-939 */
    { u64  kept = 0;
      damar_launch_suppress_flags(ix->codes, wide, n, suppress, keep, G_st);
      damar_exclusive_scan_u32(keep, off, n, scw, tot, G_st);
      damar_launch_compact_pairs(ix->codes, wide, ix->pos, keep, off, n, tk, tv, G_st);
      HIP_CHECK(hipMemcpyAsync(&kept, tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
      HIP_CHECK(hipStreamSynchronize(G_st));
      n = (u32) kept;
      HIP_CHECK(hipMemcpyAsync(ix->codes, tk, cs * (size_t) n, hipMemcpyDeviceToDevice, G_st));
      HIP_CHECK(hipMemcpyAsync(ix->pos, tv, sizeof(u32) * (size_t) n, hipMemcpyDeviceToDevice, G_st));
    }
  tick(3);
  HIP_CHECK(hipStreamSynchronize(G_st));
  sort_verify();
  G_ms[DAMAR_T_TUPLES] = lap(0, 1);
  G_ms[DAMAR_T_KSORT]  = lap(1, 2);
  G_ms[DAMAR_T_TABLE]  = lap(2, 3);
  ix->n = n;
  if (VERBOSE)
    { printf("\n   Kmer count = %u\n   Index occupies %.2fGb of HBM\n", n, ((cs + 4.) * n) / 1073741824.);
      fflush(stdout);
    }
  if (n == 0)
    { damar_index_free(ix);
      *len = 0;
      return NULL;
    }
  *len = (int) n;
  return ix;
}

extern "C" damar_dev_index *damar_index_build(damar_dev_block *blk, int own_block, int *len)
{ double t0 = now_ms();
  damar_dev_index *ix = index_build_k(blk, own_block, len, P_kmer, P_suppress, BIASED);
  H_ms[0] += now_ms() - t0;
  return ix;
}

extern "C" void damar_index_free(damar_dev_index *ix)
{ if (ix == NULL)
    return;
  HIP_CHECK(hipStreamSynchronize(G_st));
  dpool_put(ix->codes, ix->codes_bytes);       /* (the stream was synchronised above: nothing reads them any more) */
  if (ix->pos_bytes)                           /* (k <= 16: the positions live in the codes' buffer) */
    dpool_put(ix->pos, ix->pos_bytes);
  if (ix->own_block)
    damar_block_free(ix->blk);
  free(ix);
}

/* Residency figures for a scheduler that keeps blocks and indexes in HBM (daligner -P, driver.Plan): free and total bytes
   of the device, the bytes one resident block holds, and a way to hand the pool of parked index buffers back. */
extern "C" void damar_hbm_info(uint64_t *free_bytes, uint64_t *total_bytes)
{ ensure_init();
  size_t f = 0, t = 0;
  HIP_CHECK(hipMemGetInfo(&f, &t));
  if (free_bytes)  *free_bytes = (uint64_t) f + (uint64_t) DP_bytes;       /* (parked buffers are as good as free) */
  if (total_bytes) *total_bytes = (uint64_t) t;
}

extern "C" uint64_t damar_block_bytes(const damar_dev_block *b)
{ if (b == NULL)
    return 0;
  const uint64_t tot = b->d.total;
  return tot + 128 + (tot >> 2) + 64 + sizeof(u32) * ((uint64_t) b->nreads + 1) + sizeof(u32) * ((tot >> COARSE_SHIFT) + 2);
}

extern "C" void damar_pool_trim(void)
{ finish_all();
  HIP_CHECK(hipStreamSynchronize(G_st));
  for (size_t i = 0; i < DP_free.size(); i++)
    HIP_CHECK(hipFree(DP_free[i].p));
  DP_free.clear();
  DP_bytes = 0;
}

/* HBM held by one index (what its buffers took from the pool): lets a scheduler bound residency */
extern "C" uint64_t damar_index_bytes(const damar_dev_index *ix)
{ return ix == NULL ? 0 : (uint64_t) (ix->codes_bytes + ix->pos_bytes); }

extern "C" void damar_index_download(const damar_dev_index *ix, void *out)
{ struct KP { uint64 code; int rpos; int read; } *kp = (KP *) out;
  std::vector<u32> pos(ix->n), boff((size_t) ix->blk->nreads + 1);
  std::vector<u64> codes(ix->n);
  if (ix->wide)
    HIP_CHECK(hipMemcpy(codes.data(), ix->codes, sizeof(u64) * (size_t) ix->n, hipMemcpyDeviceToHost));
  else
    { std::vector<u32> c32(ix->n);
      HIP_CHECK(hipMemcpy(c32.data(), ix->codes, sizeof(u32) * (size_t) ix->n, hipMemcpyDeviceToHost));
      for (u32 i = 0; i < ix->n; i++)
        codes[i] = c32[i];
    }
  HIP_CHECK(hipMemcpy(pos.data(), ix->pos, sizeof(u32) * (size_t) ix->n, hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(boff.data(), ix->blk->boff, sizeof(u32) * boff.size(), hipMemcpyDeviceToHost));
  const int rp = ix->blk->d.rpbits;
  for (u32 i = 0; i < ix->n; i++)
    { kp[i].code = codes[i];
      if (rp)                                   /* packed position word: read << rpbits | offset in the read */
        { kp[i].rpos = (int) (pos[i] & ((1u << rp) - 1u));
          kp[i].read = (int) (pos[i] >> rp);
        }
      else
        { u32 r = (u32) (std::upper_bound(boff.begin(), boff.end(), pos[i]) - boff.begin()) - 1;
          kp[i].rpos = (int) (pos[i] - boff[r]);
          kp[i].read = (int) r;
        }
    }
}

extern "C" void *Sort_Kmers(HITS_DB *block, int *len)
{ /* block->tracks, if any, is the merged mask of daligner.c:442-497: it travels with the block */
  damar_dev_block *b = take_preloaded(block);
  if (b == NULL)
    b = damar_block_upload(block);
  return (void *) damar_index_build(b, 1, len);
}

/***** report scratch ******************************************************************************/

struct ReportScratch
{ int nslots_wanted; int   nslots, span, bwidth;
  u32   cell_cap;
  u32   ttmp_stride;
  void *state;  int *marks;  void *cells;  int *buckets;  u16 *ttmp;
  u64   state_stride, marks_stride, bucket_stride;
  short *tables;             /* SCORE then TABLE */
  const void *tables_of;     /* host spec they were copied from */
  u32  *counters;
  LaRecord *recs;  u32 rec_cap;      /* the record / trace buffers and the counters of set `cur` */
  u16  *tpool;     u32 tpool_cap;
  u32  *ctr;
  /* two sets of record / trace buffers and counters: a launch writes into one set while the records of the launch before
     it are still on their way to the host out of the other (measured: a launch waited 1.4 - 2.2 ms for that download),
     and the next launch can be queued behind a running one */
  LaRecord *recs_set[2];  u16 *tpool_set[2];  u32 rec_cap_set[2], tpool_cap_set[2];  int cur;
  u32  *widemap[2];  u32 widemap_cap[2];          /* per output set: one bit per work item of the launch's jobs (ReportArgs.widemap) */
  void *wcells;  u32 wcell_cap;  int wslots;      /* the wide kernel's 16-byte pebbles (allocated when a launch first needs them) */
};
static ReportScratch RS = {};   /* (nslots_wanted: the slot count asked for when nslots was last sized) */

static bool overlap_on(void);
static bool corun_on(void);

static int default_slots(void)
{ const char *e = getenv("DAMAR_SLOTS");
  if (e && atoi(e) > 0)
    return atoi(e);
  /* (rounds 2-3: a report launch that shares the machine with the next comparisons' seed stages left them a fifth of the
     register file, 4 of the 5 wavefronts per SIMD the round-3 kernel was compiled for.  report_packed.h's kernel now needs 64
     VGPRs and a launch now takes every wave slot: measured per config-2 step 298 / 300 / 293 / 298 / 286 ms at 4 / 5 / 6 /
     7 / 8 report wavefronts per SIMD, profiles/r04_sweeps.txt -- the seed kernels then run in the gaps the report
     wavefronts leave as they retire, and nothing is gained by reserving registers for them) */
  /* (DAMAR_REPORT_WPS: report wavefronts per SIMD of a launch, for sweeps) */
  if (const char *w = getenv("DAMAR_REPORT_WPS"))
    if (atoi(w) > 0)
      return G_prop.multiProcessorCount * 4 * damar_report2_slots_per_wave() * std::min(atoi(w), damar_report2_waves_per_simd());
  if (corun_on())
    return G_prop.multiProcessorCount * 4 * damar_report2_slots_per_wave() * (damar_report2_waves_per_simd() >= 8 ? 8 : std::min(4, damar_report2_waves_per_simd()));
  /* every wave slot of the chip: one scratch slot per wavefront of the one-pair kernel, two per wavefront of the packed one */
  return G_prop.multiProcessorCount * 4 * std::max(damar_report_waves_per_simd(), damar_report2_slots_per_wave() * damar_report2_waves_per_simd());
}

static int G_ring = 0;
/* pebbles per slot to start with: an alignment drops about 200 per kb and direction; a pool that overflows is
   quadrupled and the launch repeated.  Kept small because the driver clears what it hands out: an 8 GB pool
   (65536 cells x 8192 slots) cost every process 0.25 - 0.8 s in hipMalloc, and the 6.6 GB of 12288 slots with
   16384 cells and rings of 4096 diagonals 0.2 - 0.7 s. */
#define DEFAULT_CELLS (1u << 13)

/* the pebble pool of a slot after an overflow: a chain head holds 18 bits of pebble index (report.hip PK_HBITS) */
static u32 max_cells(void)                  /* (DAMAR_TEST_MAX_CELLS: a test hook that sends ordinary read pairs through the wide kernel) */
{ static u32 m = 0;
  if (m == 0)
    { const char *e = getenv("DAMAR_TEST_MAX_CELLS");
      m = (e && atoi(e) >= 16) ? std::min<u32>((u32) atoi(e), DAMAR_MAX_CELLS) : DAMAR_MAX_CELLS;
    }
  return m;
}

static u32 grow_cells(u32 cell_cap)
{ if (cell_cap >= max_cells())
    { fprintf(stderr, "damar: FATAL: an alignment needs more than %u trace pebbles; use a larger trace spacing (-s)\n", max_cells());
      die();
    }
  return std::min(cell_cap * 4, max_cells());
}

/* one read pair per wavefront with 8-byte pebbles (DAMAR_PACKED=0): no wide kernel runs behind that path, its limit stays loud */
static void marks_must_fit(int amax, int bmax, int tspace)
{ if (tspace > 0 && std::max(amax, bmax) / tspace + 8 > DAMAR_MAX_MARKS)
    { fprintf(stderr, "damar: FATAL: reads of %d bases need a trace spacing (-s) of at least %d here\n", std::max(amax, bmax),
              std::max(amax, bmax) / (DAMAR_MAX_MARKS - 8) + 1);
      die();
    }
}

static void scratch_prepare(int amax, int bmax, int binshift, int tspace, u32 cell_cap, hipStream_t st)
{ if (tspace <= 0)
    { fprintf(stderr, "damar: FATAL: trace spacing %d\n", tspace);
      die();
    }
  /* (reads of more than DAMAR_MAX_MARKS trace spacings -- 14 bits of trace-grid index in a packed chain head -- are the wide
     kernel's: report_launch, kernels/report.hip report_wide_kernel) */
  if (tspace > DAMAR_MAX_TSPACE)                  /* pebbles carry diagonal and wave number modulo 2^16 (kernels/report.hip: struct Cell) */
    { fprintf(stderr, "damar: FATAL: a trace spacing (-s) above %d is not supported by this build\n", DAMAR_MAX_TSPACE);
      die();
    }
  if (G_last_d2h[RS.cur] != NULL)  /* whatever is launched next overwrites the record buffers of this set */
    { HIP_CHECK(hipStreamWaitEvent(st, G_last_d2h[RS.cur], 0));
      G_last_d2h[RS.cur] = NULL;
    }
  /* The band state of a slot (used only while a band is wider than the 64 lanes) is a ring of G_ring
     diagonals, not one entry per diagonal of the pair: 1024 instead of alen + blen keeps the scratch of
     12288 slots at 2.5 GB (the driver stalls for seconds whenever a process first grows past ~24 GB).
     A band that outgrows the ring raises DAMAR_ERR_WIDE and the launch is repeated with a larger one. */
  if (G_ring == 0)
    { const char *e = getenv("DAMAR_RING");
      G_ring = e ? atoi(e) : 1024;
      if (G_ring < 128) G_ring = 128;
      while (G_ring & (G_ring - 1)) G_ring += G_ring & -G_ring;      /* next power of two */
    }
  int span = 128;
  while (span < amax + bmax + 72) span *= 2;
  if (span > G_ring) span = G_ring;
  int bwidth = (amax >> binshift) - ((-bmax) >> binshift) + 1;
  int mtp    = 2 * (std::max(amax, bmax) / tspace + 2) + 8;
  u32 tstr   = (u32) (4 * mtp + 32);
  /* the longest reads differ a little from block to block: sizes are taken with a quarter of headroom so
     that a plan line does not rebuild the scratch for every B block */
  if (RS.bwidth < bwidth)      bwidth = ((bwidth + bwidth / 4) + 255) & ~255;
  if (RS.ttmp_stride < tstr)   tstr   = ((tstr + tstr / 4) + 255u) & ~255u;
  int nslots = default_slots();
  bool grow = RS.span < span || RS.bwidth < bwidth || RS.cell_cap < cell_cap || RS.ttmp_stride < tstr;
  if (grow || (RS.nslots != nslots && RS.nslots_wanted != nslots))
    { const double g0 = now_ms();
      HIP_CHECK(hipStreamSynchronize(G_st));
      late_streams();
      HIP_CHECK(hipStreamSynchronize(G_rep));
      if (RS.state)   { HIP_CHECK(hipFree(RS.state)); HIP_CHECK(hipFree(RS.marks)); HIP_CHECK(hipFree(RS.cells));
                        HIP_CHECK(hipFree(RS.buckets)); HIP_CHECK(hipFree(RS.ttmp)); RS.state = NULL; }
      /* the scratch of all wave slots must fit what is left of HBM (very long reads: the
         per-slot band buffers grow with alen + blen): fewer resident alignments then, never a
         failed allocation in the middle of a run */
      RS.nslots_wanted = nslots;
      { size_t freeb = 0, totb = 0;
        const u64 nspan = (u64) std::max(RS.span, span), ncell = std::max(RS.cell_cap, cell_cap);
        const u64 per = damar_report_state_stride((int) nspan) + 4ull * 2 * nspan + 8ull * ncell +
                        4ull * (3ull * std::max(RS.bwidth, bwidth) + 16) + 2ull * std::max(RS.ttmp_stride, tstr);
        const double g1 = now_ms();
        HIP_CHECK(hipMemGetInfo(&freeb, &totb));
        if (getenv("DAMAR_HOSTPROF")) fprintf(stderr, "damar: scratch grow: sync+free %.1f ms, hipMemGetInfo %.1f ms\n", g1 - g0, now_ms() - g1);
        const u64 budget = (u64) (freeb * 0.6);
        if ((u64) nslots * per > budget)
          { int fit = (int) (budget / per);
            if (fit < 64)
              { fprintf(stderr, "damar: FATAL: not enough device memory for the alignment scratch (%.1f MB per wave slot, %.1f GB free)\n",
                        per / 1048576., freeb / 1073741824.);
                die();
              }
            if (VERBOSE)
              fprintf(stderr, "damar: %d instead of %d resident alignments (%.1f MB of scratch each)\n", fit, nslots, per / 1048576.);
            nslots = fit - fit % damar_report2_slots_per_wave();      /* (a wavefront of the packed kernel holds that many slots) */
          }
      }
      RS.nslots = nslots;
      RS.span   = std::max(RS.span, span);
      RS.bwidth = std::max(RS.bwidth, bwidth);
      RS.cell_cap = std::max(RS.cell_cap, cell_cap);
      RS.ttmp_stride = std::max(RS.ttmp_stride, tstr);
      RS.state_stride  = damar_report_state_stride(RS.span);
      RS.marks_stride  = (u64) 2 * RS.span;
      RS.bucket_stride = (u64) 3 * RS.bwidth + 16;
      const double g2 = now_ms();
      RS.state   = dmalloc((size_t) RS.state_stride * nslots);
      RS.marks   = (int *) dmalloc(sizeof(int) * (size_t) RS.marks_stride * nslots);
      RS.cells   = dmalloc((size_t) 8 * RS.cell_cap * nslots);                /* 8-byte pebbles (kernels/report.hip: struct Cell) */
      RS.buckets = (int *) dmalloc(sizeof(int) * (size_t) RS.bucket_stride * nslots);
      RS.ttmp    = (u16 *) dmalloc(sizeof(u16) * (size_t) RS.ttmp_stride * nslots);
      HIP_CHECK(hipMemsetAsync(RS.state, 0, (size_t) RS.state_stride * nslots, st));
      HIP_CHECK(hipMemsetAsync(RS.marks, 0, sizeof(int) * (size_t) RS.marks_stride * nslots, st));
      if (getenv("DAMAR_HOSTPROF"))
        fprintf(stderr, "damar: scratch grow: 5 allocations of %.2f GB in all %.1f ms (span %d bwidth %d cells %u)\n",
                ((double) RS.state_stride + 4. * RS.marks_stride + 8. * RS.cell_cap + 4. * RS.bucket_stride + 2. * RS.ttmp_stride) * nslots / 1073741824.,
                now_ms() - g2, RS.span, RS.bwidth, RS.cell_cap);
    }
  if (RS.counters == NULL)
    { RS.counters = (u32 *) dmalloc(sizeof(u32) * DAMAR_COUNTER_WORDS * 2);
      RS.tables   = (short *) dmalloc(sizeof(short) * 65536 * DAMAR_MAX_JOBS);
    }
  HIP_CHECK(hipMemsetAsync(RS.buckets, 0, sizeof(int) * (size_t) RS.bucket_stride * RS.nslots, st));
}

/* the record / trace buffers of set RS.cur hold at least this much (nothing is in flight on that set: its last launch has
   been completed; its download may still run) */
static void scratch_outputs(u32 rec_cap, u32 tpool_cap)
{ const int c = RS.cur;
  if (RS.rec_cap_set[c] < rec_cap || RS.tpool_cap_set[c] < tpool_cap)
    { late_streams();
      HIP_CHECK(hipStreamSynchronize(G_copy));
      G_last_d2h[c] = NULL;
      if (RS.rec_cap_set[c] < rec_cap)
        { RS.rec_cap_set[c] = rec_cap + (rec_cap >> 2);
          if (RS.recs_set[c]) HIP_CHECK(hipFree(RS.recs_set[c]));
          RS.recs_set[c] = (LaRecord *) dmalloc(sizeof(LaRecord) * (size_t) RS.rec_cap_set[c]);
        }
      if (RS.tpool_cap_set[c] < tpool_cap)
        { RS.tpool_cap_set[c] = tpool_cap + (tpool_cap >> 2);
          if (RS.tpool_set[c]) HIP_CHECK(hipFree(RS.tpool_set[c]));
          RS.tpool_set[c] = (u16 *) dmalloc(sizeof(u16) * (size_t) RS.tpool_cap_set[c]);
        }
    }
  RS.recs  = RS.recs_set[c];   RS.rec_cap   = RS.rec_cap_set[c];
  RS.tpool = RS.tpool_set[c];  RS.tpool_cap = RS.tpool_cap_set[c];
  RS.ctr   = RS.counters + (size_t) c * DAMAR_COUNTER_WORDS;
}

static void fill_report_args(ReportArgs *ra, const damar_dev_block *ab, const damar_dev_block *bb,
                             int comp, int self, Align_Spec *spec, hipStream_t st, int job, int tslot, const JobParams &jp)
{ memset(ra, 0, sizeof(*ra));
  ra->job = job;
  ra->ablk = ab->d;  ra->bblk = bb->d;
  ra->kmer = jp.kmer;  ra->hitmin = jp.hitmin;  ra->binshift = jp.binshift;
  ra->minhit = (jp.hitmin - 1) / jp.kmer + 1;
  ra->comp = comp;  ra->self = self;  ra->symmetric = jp.symmetric;
  ra->minover = jp.minover;  ra->hgap_min = jp.hgap_min;
  ra->tspace = Trace_Spacing(spec);
  ra->ave_path = damar_spec_ave_path(spec);
  ra->reach = Overlap_If_Possible(spec);
  /* SCORE/TABLE of this Align_Spec, every time (128 KB): remembering "the tables of spec X are
     already up" by X's address went wrong when a freed spec's address came back for a new one
     with another -e */
  { /* ... but only when the slot does not hold these very values already (a host copy of what is up there is compared):
       the copy comes out of pageable memory, and behind the download of the previous launch's records it held the
       launch back by 1.4 - 7 ms (scripts/gpu_gaps_api.sh) */
    static short *shadow = NULL;
    static bool   valid[DAMAR_MAX_JOBS];
    if (shadow == NULL)
      shadow = (short *) malloc(sizeof(short) * 65536 * DAMAR_MAX_JOBS);
    short *mine = shadow + (size_t) tslot * 65536;
    if (shadow == NULL || !valid[tslot] || memcmp(mine, damar_spec_score_table(spec), sizeof(short) * 65536) != 0)
      { HIP_CHECK(hipMemcpyAsync(RS.tables + (size_t) tslot * 65536, damar_spec_score_table(spec), sizeof(short) * 65536,
                                 hipMemcpyHostToDevice, st));
        if (shadow != NULL)
          { memcpy(mine, damar_spec_score_table(spec), sizeof(short) * 65536);
            valid[tslot] = true;
            HIP_CHECK(hipStreamSynchronize(st));        /* (a later launch on another stream may find the slot "up") */
          }
      }
  }
  ra->score = RS.tables + (size_t) tslot * 65536;
  ra->table = ra->score + 32768;
  { const int16 *sc = damar_spec_score_table(spec);          /* SCORE[x] = matches * mscore - (15 - matches) * dscore */
    ra->mscore = sc[32767] / 15;
    ra->dscore = -sc[0] / 15;
  }
  ra->state = RS.state;  ra->state_stride = RS.state_stride;  ra->span = RS.span;
  ra->marks = RS.marks;  ra->marks_stride = RS.marks_stride;
  ra->cells = RS.cells;  ra->cell_cap = RS.cell_cap;
  ra->buckets = RS.buckets;  ra->bucket_stride = RS.bucket_stride;  ra->bwidth = RS.bwidth;
  ra->bucket_bits = ilog2_ceil((u64) RS.bwidth + 2);
  ra->ttmp = RS.ttmp;  ra->ttmp_stride = RS.ttmp_stride;
  ra->recs = RS.recs;  ra->rec_cap = RS.rec_cap;
  ra->tpool = RS.tpool;  ra->tpool_cap = RS.tpool_cap;
  ra->t8 = 0;                                   /* (the pipeline's launches set it: report_launch) */
  ra->widemap = NULL;  ra->wcells = NULL;  ra->wcell_cap = 0;  ra->cell_max = DAMAR_MAX_CELLS;
  { static int lim = -1;                        /* test hook: a lower limit makes the 16-bit re-launch happen (tests/test_gpu_configs.py) */
    if (lim < 0)
      lim = getenv("DAMAR_TEST_T8_LIMIT") ? atoi(getenv("DAMAR_TEST_T8_LIMIT")) : 255;
    ra->t8max = lim;
  }
  ra->counters = RS.ctr;
  ra->cursor = RS.ctr + DAMAR_CNT_CURSOR + job;
  ra->nfilt  = RS.ctr + DAMAR_CNT_NFILT + job;
}


/* Two read pairs per wavefront (kernels/report_packed.h) unless DAMAR_PACKED=0.  Measured on config 2 (profiles/r02_*):
 * identical output, 356 ms of report kernel per step against 366 ms for one pair per wavefront (357 ms at the end of
 * round 1).  Both kernels are bound by how long ONE wavefront takes for a wave step (~4500-6000 cycles of serial issue;
 * the scalar unit is 86 % busy in the one-pair kernel), not by HBM; the packed one advances two alignments per such
 * step but fits 4 instead of 8 wavefronts per SIMD (its wave loop needs ~100 VGPRs). */
static bool use_packed(const ReportArgs *ra, int amax, int bmax)
{ static int want = -1;
  if (want < 0)
    { const char *e = getenv("DAMAR_PACKED");
      want = e ? atoi(e) : 1;
    }
  if (!want || (RS.nslots % damar_report2_slots_per_wave()) || ra->tspace <= 0)
    return false;
  if (ra->mscore * 8 > 32000 || ra->dscore * 8 > 32000)
    return false;
  return true;
}

/***** host tail (filter.c:2442-2483 per read pair) and its optional worker thread *********************/

struct RecOrder
{ bool operator()(const LaRecord &x, const LaRecord &y) const
  { return (x.item != y.item) ? (x.item < y.item) : (x.seq < y.seq); }
};

static double now_ms(void)
{ struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

/* Page-locked landing buffers for the records and trace points of one report launch, recycled
 * through a small pool: the download runs at PCIe speed and nothing is allocated or touched
 * for the first time inside the timed loop. */
static bool   A_on = false;                   /* the asynchronous host pipeline is running (damar_set_async) */
struct HostBuf
{ LaRecord *recs;   size_t rec_cap;
  u16      *tpool;  size_t tp_cap;
  size_t    nrec, ntp;
  hipEvent_t e0, e1;               /* around the download; e1 is what the tail thread waits for */
  bool      pending;
  int       t8;                    /* the trace values are bytes (the launch compressed them: ReportArgs.t8) */
  u32      *wmap;  size_t wmap_cap;  /* nwide > 0: the jobs' bit maps of the read pairs the wide kernel took over (job j from wm_off[j]) */
  u32       nwide, wm_off[DAMAR_MAX_JOBS + 1];
  int       users;                 /* comparisons of the launch whose tails have not run yet */
};
static std::mutex             &HB_mu   = *new std::mutex();
static std::vector<HostBuf *> &HB_free = *new std::vector<HostBuf *>();

/* Page-locking costs a millisecond per 4 - 5 MB (hipHostMalloc of 70 MB: 13 ms, a hipHostFree 5 ms: scripts/gpu_hiptrace.sh)
   and the thread that asks is the one that launches the kernels: in a cold four-line plan the two buffers, each
   allocated at its first launch and grown at its second (cross comparisons hold twice the records of the self
   comparison a plan starts with), were 60 ms of that thread's 490.  So a buffer is sized for 2.5 x what its first launch
   needs, and the first one brings a twin along, allocated by a thread of its own while the tail of the first launch
   runs. */
static void hostbuf_size(HostBuf *h, size_t nrec, size_t ntp, bool generous)
{ if (h->rec_cap < nrec)
    { if (h->recs) HIP_CHECK(hipHostFree(h->recs));
      h->rec_cap = generous ? nrec + nrec + (nrec >> 1) + 4096 : nrec + (nrec >> 2) + 4096;
      HIP_CHECK(hipHostMalloc((void **) &h->recs, sizeof(LaRecord) * h->rec_cap, hipHostMallocDefault));
    }
  if (h->tp_cap < ntp)
    { if (h->tpool) HIP_CHECK(hipHostFree(h->tpool));
      h->tp_cap = generous ? ntp + ntp + (ntp >> 1) + 65536 : ntp + (ntp >> 2) + 65536;
      HIP_CHECK(hipHostMalloc((void **) &h->tpool, sizeof(u16) * h->tp_cap, hipHostMallocDefault));
    }
}

static std::thread *HB_twin = NULL;           /* allocates the second landing buffer (hostbuf_get); joined by helpers_join() */
static HostBuf *hostbuf_new(void)
{ HostBuf *h = new HostBuf();
  memset(h, 0, sizeof(*h));
  HIP_CHECK(hipEventCreate(&h->e0));
  HIP_CHECK(hipEventCreate(&h->e1));
  return h;
}

static void tail_pool_join(void);
static void helpers_join(void)
{ late_streams();
  tail_pool_join();
  std::thread *t = NULL;
  { std::lock_guard<std::mutex> lk(HB_mu);
    t = HB_twin;  HB_twin = NULL;
  }
  if (t != NULL)
    { t->join();
      delete t;
    }
}

static HostBuf *hostbuf_get(size_t nrec, size_t ntp)
{ HostBuf *h = NULL;
  static std::atomic<int> made(0);
  { std::lock_guard<std::mutex> lk(HB_mu);
    if (!HB_free.empty())
      { h = HB_free.back();
        HB_free.pop_back();
      }
  }
  if (h == NULL)
    h = hostbuf_new();
  h->pending = false;
  const bool first = (h->rec_cap == 0 && h->tp_cap == 0);
  hostbuf_size(h, nrec, ntp, first);
  if (first && made.fetch_add(1) == 0 && A_on)
    { /* the twin of the process's first buffer (asynchronous mode: a second launch's records land while the first
         one's are with the tail) */
      const size_t rc = h->rec_cap, tc = h->tp_cap;
      const int dev = G_device;
      HB_twin = new std::thread([rc, tc, dev]()
        { HIP_CHECK(hipSetDevice(dev));
          HostBuf *t = hostbuf_new();
          t->rec_cap = rc;  t->tp_cap = tc;
          HIP_CHECK(hipHostMalloc((void **) &t->recs, sizeof(LaRecord) * rc, hipHostMallocDefault));
          HIP_CHECK(hipHostMalloc((void **) &t->tpool, sizeof(u16) * tc, hipHostMallocDefault));
          std::lock_guard<std::mutex> lk(HB_mu);
          HB_free.push_back(t);
        });
    }
  h->nrec = nrec;  h->ntp = ntp;
  return h;
}

static void hostbuf_put(HostBuf *h)
{ std::lock_guard<std::mutex> lk(HB_mu);
  HB_free.push_back(h);
}

/* byte trace values of the device (ReportArgs.t8) into the 16-bit pool the redundancy handling works on */
static int64 tpool_push8(damar_tpool *tp, const u8 *src, int n)
{ static thread_local std::vector<uint16> wide;
  wide.resize((size_t) n);
  for (int i = 0; i < n; i++)
    wide[i] = src[i];
  return damar_tpool_push(tp, wide.data(), n);
}

/* One contiguous range [lo, hi) of the ordered records (whole read pairs): filter.c:2442-2483
 * per pair, results appended to obuf. */
static int64 tail_range(const LaRecord *recs, const u32 *ord, const u64 *okey, size_t lo, size_t hi, const u16 *tpool, int t8,
                        const HITS_DB *ablock, const HITS_DB *bblock, int self, int comp, int ts,
                        Overlap_IO_Buffer *obuf, int symmetric, int hgap_min)
{ int64 ncheck = 0;
  std::vector<damar_path> am, bm;
  damar_tpool tp = { NULL, 0, 0 };
  std::vector<char> aread_buf, bread_buf;                       /* (packed host blocks: the reads of a pair that is bridged) */
  size_t i = lo;
  while (i < hi)
    { size_t j = i;
      /* the records arrive in completion order and are visited in item order: every record and every trace is a
         cache miss in 300 MB of landing buffer unless it is asked for ahead (records 16 ahead, their traces 8 ahead) */
      if (i + 16 < hi)
        __builtin_prefetch(&recs[ord[i + 16]]);
      if (i + 8 < hi)
        { const LaRecord &pr = recs[ord[i + 8]];
          const size_t vb = t8 ? sizeof(u8) : sizeof(u16);
          const char *pt = (const char *) tpool + vb * (size_t) pr.toff;
          const size_t nb = vb * (size_t) (pr.atlen + pr.btlen);
          for (size_t o = 0; o < nb; o += 64)
            __builtin_prefetch(pt + o);
        }
      while (j < hi && (okey[j] >> 32) == (okey[i] >> 32))          /* (the item sits in the key: no record is touched for this) */
        j += 1;
      const int ar = recs[ord[i]].aread, br = recs[ord[i]].bread;
      const int al = ablock->reads[ar].rlen, bl = bblock->reads[br].rlen;
      const int doA = (al >= hgap_min);
      const int doB = (symmetric && bl >= hgap_min && (ar != br || !self || !comp));   /* filter.c:2300-2301 */
      if (j - i == 1)
        { /* one alignment for the pair -- the common case: nothing for Handle_Redundancies to look at, so the records
             (filter.c:2470-2483: the A view, then the B view) are written straight from the downloaded traces, which
             Compress_TraceTo8 may shorten in place (the landing buffer is this comparison's own) */
          const LaRecord &r = recs[ord[i]];
          Overlap ovl;
          memset(&ovl, 0, sizeof(ovl));
          ovl.flags = (uint32) comp;
          if (doA)
            { ovl.aread = ar + ablock->ufirst;  ovl.bread = br + bblock->ufirst;
              ovl.path.tlen = r.atlen;  ovl.path.diffs = r.diffs;
              ovl.path.abpos = r.abpos;  ovl.path.bbpos = r.bbpos;  ovl.path.aepos = r.aepos;  ovl.path.bepos = r.bepos;
              /* (t8: the device has compressed the values already -- and raised DAMAR_ERR_T8 had one not fitted) */
              ovl.path.trace = t8 ? (void *) ((const u8 *) tpool + r.toff) : (void *) (tpool + r.toff);
              if (ts <= TRACE_XOVR && !t8)
                Compress_TraceTo8(&ovl, 1);
              AddOverlapToBuffer(obuf, &ovl, (ts <= TRACE_XOVR) ? 1 : 2);
              ncheck += 1;
            }
          if (doB)
            { ovl.aread = br + bblock->ufirst;  ovl.bread = ar + ablock->ufirst;
              ovl.path.tlen = r.btlen;  ovl.path.diffs = r.diffs;
              if (comp)                                          /* align.c:2039-2042 */
                { ovl.path.abpos = bl - r.bepos;  ovl.path.bbpos = al - r.aepos;
                  ovl.path.aepos = bl - r.bbpos;  ovl.path.bepos = al - r.abpos;
                }
              else                                               /* align.c:2059-2062 */
                { ovl.path.abpos = r.bbpos;  ovl.path.bbpos = r.abpos;  ovl.path.aepos = r.bepos;  ovl.path.bepos = r.aepos; }
              ovl.path.trace = t8 ? (void *) ((const u8 *) tpool + r.toff + r.atlen) : (void *) (tpool + r.toff + r.atlen);
              if (ts <= TRACE_XOVR && !t8)
                Compress_TraceTo8(&ovl, 1);
              AddOverlapToBuffer(obuf, &ovl, (ts <= TRACE_XOVR) ? 1 : 2);
              ncheck += 1;
            }
          i = j;
          continue;
        }
      am.clear();  bm.clear();  tp.top = 0;
      for (size_t q = i; q < j; q++)
        { const LaRecord &r = recs[ord[q]];
          damar_path p;
          if (doA)
            { p.tlen = r.atlen;  p.diffs = r.diffs;
              p.abpos = r.abpos;  p.bbpos = r.bbpos;  p.aepos = r.aepos;  p.bepos = r.bepos;
              p.toff = t8 ? tpool_push8(&tp, (const u8 *) tpool + r.toff, r.atlen) : damar_tpool_push(&tp, tpool + r.toff, r.atlen);
              am.push_back(p);
            }
          if (doB)
            { p.tlen = r.btlen;  p.diffs = r.diffs;
              if (comp)                                          /* align.c:2039-2042 */
                { p.abpos = bl - r.bepos;  p.bbpos = al - r.aepos;
                  p.aepos = bl - r.bbpos;  p.bepos = al - r.abpos;
                }
              else                                               /* align.c:2059-2062 */
                { p.abpos = r.bbpos;  p.bbpos = r.abpos;  p.aepos = r.bepos;  p.bepos = r.aepos; }
              p.toff = t8 ? tpool_push8(&tp, (const u8 *) tpool + r.toff + r.atlen, r.btlen)
                          : damar_tpool_push(&tp, tpool + r.toff + r.atlen, r.btlen);
              bm.push_back(p);
            }
        }
      damar_bridge_ctx bctx;
      if (am.size() > 1 || bm.size() > 1)      /* (only Handle_Redundancies looks at the sequences) */
        { bctx.aseq = host_read(ablock, ar, 0, aread_buf);  bctx.bseq = host_read(bblock, br, comp, bread_buf); }
      else
        { bctx.aseq = NULL;  bctx.bseq = NULL; }
      bctx.alen = al;  bctx.blen = bl;
      damar_emit_pair(am.data(), (int) am.size(), bm.data(), (int) bm.size(), &tp, comp, ts,
                      ar + ablock->ufirst, br + bblock->ufirst, &bctx, obuf, &ncheck);
      i = j;
    }
  free(tp.val);
  damar_bridge_release();
  return ncheck;
}

static int tail_threads(void)
{ static int n = 0;
  if (n == 0)
    { const char *e = getenv("DAMAR_TAIL_THREADS");
      n = e ? atoi(e) : 4;
      if (n < 1) n = 1;
      if (n > 64) n = 64;
    }
  return n;
}

/* The threads of the tail are kept between calls: a comparison's tail is 0.8 ms of work and four threads created and joined
   for it were 0.15 ms of that (config 3: 306 tails).  tail_pool_run(n, fn) runs fn(0) .. fn(n - 1), fn(0) on the caller. */
static struct TailPool
{ std::mutex              mu, run_mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> th;
  const std::function<void(int)> *fn = NULL;
  int      n = 0, left = 0;
  uint64_t gen = 0;
  bool     stop = false;
} TP;

static void tail_pool_worker(int w)
{ uint64_t seen = 0;
  for (;;)
    { const std::function<void(int)> *fn;
      { std::unique_lock<std::mutex> lk(TP.mu);
        TP.cv_work.wait(lk, [&] { return TP.stop || TP.gen != seen; });
        if (TP.stop)
          return;
        seen = TP.gen;
        if (w >= TP.n)
          continue;
        fn = TP.fn;
      }
      (*fn)(w);
      { std::lock_guard<std::mutex> lk(TP.mu);
        if (--TP.left == 0)
          TP.cv_done.notify_one();
      }
    }
}

static void tail_pool_run(int n, const std::function<void(int)> &fn)
{ std::lock_guard<std::mutex> one(TP.run_mu);
  { std::lock_guard<std::mutex> lk(TP.mu);
    while ((int) TP.th.size() < n - 1)
      { const int w = (int) TP.th.size() + 1;
        TP.th.emplace_back(tail_pool_worker, w);
      }
    TP.fn = &fn;  TP.n = n;  TP.left = n - 1;  TP.gen += 1;
  }
  TP.cv_work.notify_all();
  fn(0);
  std::unique_lock<std::mutex> lk(TP.mu);
  TP.cv_done.wait(lk, [&] { return TP.left == 0; });
  TP.fn = NULL;  TP.n = 0;
}

static void tail_pool_join(void)
{ { std::lock_guard<std::mutex> lk(TP.mu);
    TP.stop = true;
  }
  TP.cv_work.notify_all();
  for (auto &t : TP.th)
    t.join();
  TP.th.clear();
  TP.stop = false;
}

static int64 run_tail(LaRecord *recs, size_t nrecs_all, const u16 *tpool, int t8, const u32 *wmap,
                      const HITS_DB *ablock, const HITS_DB *bblock, int self, int comp, Align_Spec *spec,
                      const JobParams &jp, int jobid = 0, int njobs = 1)
{ const int ts = Trace_Spacing(spec);
  /* (work item, sequence) order = the reference's order of read pairs and of the alignments inside one.  ONE pass over
     the records (of a launch over several comparisons: those of this one, top byte of seq) collects 8-byte keys
     item << 32 | seq; everything after that -- a counting sort on the item (a wave emits the records of its item in
     sequence order; the insertion pass only guards that), the pair boundaries, the thread cuts -- works on the keys,
     which stay in cache, instead of on 48-byte records scattered over the landing buffer */
  const bool tprof = getenv("DAMAR_HOSTPROF") != NULL;
  const double tp0 = tprof ? now_ms() : 0.;
  std::vector<u64> key;
  std::vector<u32> idx;
  key.reserve(nrecs_all);  idx.reserve(nrecs_all);
  u32 maxitem = 0;
  for (size_t q = 0; q < nrecs_all; q++)
    if (njobs <= 1 || (int) (recs[q].seq >> DAMAR_SEQ_BITS) == jobid)
      { u32 it = recs[q].item;
        if (it & DAMAR_ITEM_WIDE)                    /* the wide kernel's record */
          { it &= ~DAMAR_ITEM_WIDE;
            recs[q].item = it;
          }
        else if (wmap != NULL && ((wmap[it >> 5] >> (it & 31)) & 1u))
          continue;                                   /* what the two-pair kernel wrote for a pair before it gave it up */
        key.push_back(((u64) it << 32) | (u64) recs[q].seq);
        idx.push_back((u32) q);
        if (it > maxitem) maxitem = it;
      }
  const size_t nrecs = key.size();
  const double tp1 = tprof ? now_ms() : 0.;
  std::vector<u32> ord(nrecs);
  std::vector<u64> okey(nrecs);
  { std::vector<u32> first((size_t) maxitem + 2, 0);
    for (size_t q = 0; q < nrecs; q++)
      first[(size_t) (key[q] >> 32) + 1] += 1;
    for (size_t q = 1; q < first.size(); q++)
      first[q] += first[q - 1];
    for (size_t q = 0; q < nrecs; q++)
      { const u32 at = first[(size_t) (key[q] >> 32)]++;
        ord[at] = idx[q];  okey[at] = key[q];
      }
    for (size_t q = 1; q < nrecs; q++)
      { const u64 x = okey[q];
        const u32 xi = ord[q];
        size_t r = q;
        while (r > 0 && okey[r - 1] > x && (okey[r - 1] >> 32) == (x >> 32))
          { okey[r] = okey[r - 1];  ord[r] = ord[r - 1];  r -= 1; }
        okey[r] = x;  ord[r] = xi;
      }
  }
  const double tp2 = tprof ? now_ms() : 0.;
  Overlap_IO_Buffer *obuf = OVL_IO_Buffer(spec);
  static size_t tmin = 0;                  /* below this many records one thread does it (DAMAR_TAIL_MIN) */
  if (tmin == 0)
    { const char *e = getenv("DAMAR_TAIL_MIN");
      tmin = (e && atol(e) > 0) ? (size_t) atol(e) : 8192;
    }
  const int nthr = (nrecs >= tmin) ? tail_threads() : 1;
  if (nthr == 1)
    return tail_range(recs, ord.data(), okey.data(), 0, nrecs, tpool, t8, ablock, bblock, self, comp, ts, obuf, jp.symmetric, jp.hgap_min);

  /* Read pairs are independent: cut the ordered records into nthr ranges at pair boundaries,
     let each thread fill a private buffer, append the buffers in order. */
  std::vector<size_t> cut(nthr + 1, nrecs);
  cut[0] = 0;
  for (int t = 1; t < nthr; t++)
    { size_t c = std::max(cut[t - 1], nrecs * (size_t) t / nthr);
      while (c < nrecs && c > 0 && (okey[c] >> 32) == (okey[c - 1] >> 32))
        c += 1;
      cut[t] = c;
    }
  /* The Align_Spec holds one buffer per -j thread and the writer gathers them all and sorts (align.c:6166-6200), so tail
     thread t appends to buffer t: no private buffer to copy over afterwards (a read pair's records stay together in one
     buffer, which is all the sort's tie-break needs).  With fewer -j buffers than tail threads: private buffers, appended
     in order. */
  const bool direct = Num_Threads(spec) >= nthr;
  std::vector<Overlap_IO_Buffer *> part(nthr, (Overlap_IO_Buffer *) NULL);
  std::vector<int64> got(nthr, 0);
  for (int t = 0; t < nthr; t++)
    { if (direct)
        part[t] = obuf + t;
      else
        { part[t] = CreateOverlapBuffer(4 * nthr, obuf->tbytes ? obuf->tbytes : 1, obuf->no_trace);
          if (part[t] == NULL)
            die();
        }
    }
  tail_pool_run(nthr, [&](int t) { got[t] = tail_range(recs, ord.data(), okey.data(), cut[t], cut[t + 1], tpool, t8, ablock, bblock,
                                                       self, comp, ts, part[t], jp.symmetric, jp.hgap_min); });
  int64 ncheck = 0;
  for (int t = 0; t < nthr; t++)
    { if (!direct)
        { if (damar_append_overlap_buffer(obuf, part[t]))
            { fprintf(stderr, "damar: FATAL: out of memory appending overlaps\n");
              die();
            }
          free(part[t]->ovls);
          free(part[t]->trace);
          free(part[t]);
        }
      ncheck += got[t];
    }
  if (tprof)
    fprintf(stderr, "damar: tail of %zu records: keys %.2f ms, order %.2f ms, %d threads %.2f ms\n", nrecs, tp1 - tp0, tp2 - tp1, nthr, now_ms() - tp2);
  return ncheck;
}

/* Asynchronous mode (damar_set_async(1)): the host side runs as a two-stage pipeline behind
 * the GPU.  Stage 1 (one thread) does the tail of each Match_Filter in submission order and,
 * for a write request, detaches the filled overlap buffers from the Align_Spec; stage 2 (a
 * second thread) sorts and writes the detached buffers.  The overlap buffers of an Align_Spec
 * are touched by stage 1 only; damar_async_drain() must be called before the blocks or the
 * Align_Spec involved are released, and before the counters are read. */
static bool G_tail_reports = false;          /* a tail job writes its number of confirmed hits into the caller's job struct (damar_match with -v) */
struct TailJob
{ int kind;                                  /* 0 = tail of one Match_Filter, 1 = write + reset */
  HostBuf *hb;
  int  jobid, njobs;                         /* which comparison of the launch whose records hb holds */
  HITS_DB ablock, bblock;                    /* copies of the block records: the caller may reuse its structs
                                                (the read tables and bases they point to must stay alive) */
  int  self, comp;
  JobParams jp;                              /* the options the comparison ran under */
  Align_Spec *spec;
  std::string d1, d2, a, b;
  bool has1, has2;
  int  last;
  damar_write_params wp;                     /* stage 2 */
  Overlap_IO_Buffer *bufs;
  int64 *got;                                /* kind 0, -v only: where the caller wants the number of confirmed hits (it drains before it reads) */
};

/* Heap objects that are never destroyed: at process exit a worker may still be parked in
 * wait(), and destroying a condition variable with a waiter (static destructors) hangs. */
struct Stage
{ std::mutex              mu;
  std::condition_variable cv, idle;
  std::deque<TailJob *>   queue;
  int                     busy;              /* jobs taken off the queue and not finished yet */
  bool                    quit;
  std::vector<std::thread *> threads;
  Stage() : busy(0), quit(false) {}
};
static Stage &A_s1 = *new Stage();
static Stage &A_s2 = *new Stage();
static std::mutex &A_mu = *new std::mutex();            /* the totals below */
static int64  A_ncheck = 0;
static double A_tail_ms = 0, A_write_ms = 0, A_d2h_ms = 0;

static void stage_submit(Stage &st, TailJob *job)
{ { std::lock_guard<std::mutex> lk(st.mu);
    st.queue.push_back(job);
  }
  st.cv.notify_one();
}

static TailJob *stage_next(Stage &st)
{ std::unique_lock<std::mutex> lk(st.mu);
  st.cv.wait(lk, [&st] { return st.quit || !st.queue.empty(); });
  if (st.queue.empty())
    return NULL;
  TailJob *job = st.queue.front();
  st.queue.pop_front();
  st.busy += 1;
  return job;
}

static void stage_done(Stage &st)
{ { std::lock_guard<std::mutex> lk(st.mu);
    st.busy -= 1;
  }
  st.idle.notify_all();
}

static void stage_drain(Stage &st)
{ std::unique_lock<std::mutex> lk(st.mu);
  st.idle.wait(lk, [&st] { return st.queue.empty() && st.busy == 0; });
}

static void tail_worker(void)
{ for (;;)
    { TailJob *job = stage_next(A_s1);
      if (job == NULL)
        return;
      double t0 = now_ms();
      if (job->kind == 0)
        { if (job->hb->pending)
            { float ms = 0;
              HIP_CHECK(hipSetDevice(G_device));          /* the current device is a per-thread setting */
              HIP_CHECK(hipEventSynchronize(job->hb->e1));
              HIP_CHECK(hipEventElapsedTime(&ms, job->hb->e0, job->hb->e1));
              job->hb->pending = false;                   /* (the other comparisons of the launch share the buffer) */
              std::lock_guard<std::mutex> lk(A_mu);
              A_d2h_ms += ms;
              t0 = now_ms();
            }
          int64 n = run_tail(job->hb->recs, job->hb->nrec, job->hb->tpool, job->hb->t8,
                               job->hb->nwide ? job->hb->wmap + job->hb->wm_off[job->jobid] : (const u32 *) NULL, &job->ablock, &job->bblock,
                             job->self, job->comp, job->spec, job->jp, job->jobid, job->njobs);
          if (job->got != NULL)
            *job->got = n;
          if (--job->hb->users == 0)
            hostbuf_put(job->hb);
          delete job;
          std::lock_guard<std::mutex> lk(A_mu);
          A_ncheck += n;
          A_tail_ms += now_ms() - t0;
        }
      else
        { job->bufs = damar_detach_overlap_buffers(job->spec, &job->wp);
          stage_submit(A_s2, job);
          std::lock_guard<std::mutex> lk(A_mu);
          A_tail_ms += now_ms() - t0;
        }
      stage_done(A_s1);
    }
}

static void write_worker(void)
{ for (;;)
    { TailJob *job = stage_next(A_s2);
      if (job == NULL)
        return;
      double t0 = now_ms();
      damar_write_detached(&job->wp, job->bufs, job->has1 ? job->d1.c_str() : NULL,
                           job->has2 ? job->d2.c_str() : NULL, job->a.c_str(), job->b.c_str(), job->last);
      delete job;
      { std::lock_guard<std::mutex> lk(A_mu);
        A_write_ms += now_ms() - t0;
      }
      stage_done(A_s2);
    }
}

static void async_submit(TailJob *job)
{ stage_submit(A_s1, job);
}

extern "C" void damar_async_drain(void)
{ if (!A_on)
    return;
  finish_all();
  stage_drain(A_s1);           /* stage 1 feeds stage 2: drain in pipeline order */
  stage_drain(A_s2);
}

extern "C" void damar_set_async(int on)
{ if (on && !A_on)
    { A_s1.quit = A_s2.quit = false;
      A_s1.threads.push_back(new std::thread(tail_worker));      /* one: the tails append to the overlap buffers in order */
      /* the sort + write of a block pair's files is independent of every other pair's: two writers by default
         (DAMAR_WRITE_THREADS), so that the files of the last pairs are not written one after the other at the drain */
      { const char *e = getenv("DAMAR_WRITE_THREADS");
        int n = e ? atoi(e) : 2;
        if (n < 1) n = 1;
        if (n > 16) n = 16;
        for (int i = 0; i < n; i++)
          A_s2.threads.push_back(new std::thread(write_worker));
      }
      A_on = true;
    }
  else if (!on && A_on)
    { damar_async_drain();
      Stage *st[2] = { &A_s1, &A_s2 };
      for (int i = 0; i < 2; i++)
        { { std::lock_guard<std::mutex> lk(st[i]->mu);
            st[i]->quit = true;
          }
          st[i]->cv.notify_all();
          for (std::thread *th : st[i]->threads)
            { th->join();
              delete th;
            }
          st[i]->threads.clear();
        }
      A_on = false;
      helpers_join();
    }
}

/* totals since the last call: confirmed records, tail ms, write ms (drains first) */
extern "C" void damar_async_totals(int64 *ncheck, double *tail_ms, double *write_ms)
{ const double d0 = now_ms();
  damar_async_drain();
  H_ms[7] += now_ms() - d0;
  if (getenv("DAMAR_HOSTPROF"))
    { fprintf(stderr, "damar host wall ms:");
      for (int i = 0; i < 8; i++)
        { fprintf(stderr, " %s=%.1f", H_name[i], H_ms[i]);
          H_ms[i] = 0;
        }
      fprintf(stderr, " | between calls=%.1f launch->next front=%.1f flush->exit=%.1f entry->front=%.1f front->tick0=%.1f\n",
              Q_ms[2], Q_ms[3], Q_seg[0], Q_seg[2], Q_seg[3]);
      Q_ms[2] = Q_ms[3] = 0;  Q_exit = 0;  Q_seg[0] = Q_seg[2] = Q_seg[3] = 0;
    }
  std::lock_guard<std::mutex> lk(A_mu);
  if (ncheck)   *ncheck = A_ncheck;
  if (tail_ms)  *tail_ms = A_tail_ms;
  if (write_ms) *write_ms = A_write_ms;
  A_ncheck = 0;  A_tail_ms = 0;  A_write_ms = 0;
}

/* milliseconds the downloads of the asynchronous mode took since the last call (drains first) */
extern "C" double damar_async_d2h_ms(void)
{ damar_async_drain();
  std::lock_guard<std::mutex> lk(A_mu);
  double v = A_d2h_ms;
  A_d2h_ms = 0;
  return v;
}

/***** Match_Filter **********************************************************************************/

static std::vector<u64> G_seed_keys;
static std::vector<u32> G_seed_vals;
static int  G_seed_pbits = 0, G_seed_abits = 0, G_seed_dbits = 0;
static int  G_keep_seeds = 0;

static int64 sizeof_db(const HITS_DB *db)      /* db/DB.c:726 sizeof_DB without tracks */
{ return (int64) sizeof(HITS_DB) + (int64) sizeof(HITS_READ) * (db->nreads + 2) + db->totlen + db->nreads + 4 +
         (db->path ? (int64) strlen(db->path) + 1 : 0);
}


/* What the seed stage of one comparison leaves on the device for the report launch. */
struct Front
{ const u64 *keys;  const u32 *vals;  u64 total;
  const u32 *work;  u32 nwork;
  const u32 *order;
  int pbits, abits, dbits;
  JobParams jp;                 /* parameters and options of the moment the seed stage ran */
  size_t bytes;                 /* of the two arenas that hold the above */
  double t_entry;
};

/* Seed stage of one comparison (filter.c:2603-2760): merge-count, scan, emit, seed sort, work list and its
   processing order, into the arenas of job slot `slot`.  Returns false when there is nothing to report. */
static bool match_front(damar_match_job *job, int slot, Front *f)
{ const HITS_DB *ablock = job->ablock, *bblock = job->bblock;
  damar_dev_index *aidx = job->aidx, *bidx = job->bidx;
  const int self = job->self, comp = job->comp;
  Arena &G_hits = G_hitsJ[slot], &G_ord = G_ordJ[slot];
  int64 nhits = 0;
  const double t_entry = now_ms();
  pick_sort_shape();
  memset(f, 0, sizeof(*f));
  f->t_entry = t_entry;
  f->jp = params_now();
  job->counts[0] = job->counts[1] = job->counts[2] = 0;
  if (aidx == NULL || bidx == NULL || aidx->n == 0 || bidx->n == 0)
    return false;
  if (aidx->kbits != bidx->kbits)
    { fprintf(stderr, "damar: internal error, index parameters differ\n");
      die();
    }
  const u32 alen = aidx->n, blen = bidx->n;
  MergeArgs m;
  memset(&m, 0, sizeof(m));
  m.acode = aidx->codes;  m.apos = aidx->pos;  m.alen = alen;
  m.bcode = bidx->codes;  m.bpos = bidx->pos;  m.blen = blen;
  m.wide = aidx->wide;
  m.kbits = aidx->kbits;
  m.self = self;  m.comp = comp;  m.identity = IDENTITY;
  m.limit = (MEM_LIMIT > 0) ? MAXGRAM : 0x7fffffffu;      /* filter.c:2700-2702 */
  m.ablk = aidx->blk->d;  m.bblk = bidx->blk->d;
  m.pbits = std::max(1, ilog2_ceil((u64) ablock->maxlen + 1));
  m.abits = std::max(1, ilog2_ceil((u64) ablock->nreads));
  int bbits = std::max(1, ilog2_ceil((u64) bblock->nreads));
  if (m.pbits + m.abits + bbits > 64)
    { fprintf(stderr, "damar: FATAL: seed key needs %d bits (> 64)\n", m.pbits + m.abits + bbits);
      die();
    }
  /* Packed seeds: when position-in-B fits beside the sort key, a seed is ONE u64 (pair | apos | bpos) and the sort moves
     8 bytes per seed instead of 12; otherwise the diagonal travels in a second array (DAMAR_PACK_SEEDS=0 forces that) */
  { static int pack_on = -1;
    if (pack_on < 0)
      { const char *e = getenv("DAMAR_PACK_SEEDS");
        pack_on = e ? atoi(e) : 1;
      }
    const int db = std::max(1, ilog2_ceil((u64) bblock->maxlen + 1));
    m.dbits = (pack_on && m.pbits + m.abits + bbits + db <= 64) ? db : 0;
  }

  /* ---- merge: COUNT sweep, scan of the tile totals, (lower cap and COUNT again), EMIT sweep ---- */
  const u32 mtiles = damar_merge_tiles(alen);
  void *mw;
  u32  *tcount;
  u64  *tot;
  u64   total = 0;

  if (Q_flush > 0)
    { Q_ms[3] += now_ms() - Q_flush;  Q_flush = 0; }
  Q_seg[3] += now_ms() - f->t_entry;
  tick(0);
  arena_reserve(&G_work, pad256(damar_merge_workspace_bytes(alen)) + pad256(damar_scan_workspace_bytes(mtiles)) + 4096);
  mw = arena_take(&G_work, damar_merge_workspace_bytes(alen));
  tcount = damar_merge_tile_counts(mw, alen);
  tot = (u64 *) arena_take(&G_work, 64);
  void *mscw = arena_take(&G_work, damar_scan_workspace_bytes(mtiles));
  damar_launch_merge_count(&m, mw, NULL, 0, G_st);
  stage("merge_count");
  damar_exclusive_scan_u32(tcount, tcount, mtiles, mscw, tot, G_st);
  stage("merge_scan");
  HIP_CHECK(hipMemcpyAsync(&total, tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
  stream_wait(G_st);
  if (MEM_LIMIT > 0)
    { /* filter.c:2634-2699.  The counts above keep every run below MAXGRAM; the reference lowers
         that cap to the first mutual count at which the kept seeds no longer fit `avail`.  That
         only happens under memory pressure, so the histogram is built only then. */
      int   limit = MAXGRAM;
      int64 avail = (int64) (MEM_LIMIT - (uint64) (sizeof_db(ablock) + sizeof_db(bblock))) / 16;
      if (aidx == bidx || avail > (int64) alen + 2 * (int64) blen)
        avail = (avail - alen) / 2;
      else
        avail = avail - ((int64) alen + blen);
      avail = (int64) (avail * .98);
      if ((int64) total > avail)
        { std::vector<unsigned long long> histo(MAXGRAM);
          unsigned long long *dgram = (unsigned long long *) dmalloc(sizeof(unsigned long long) * MAXGRAM);
          HIP_CHECK(hipMemsetAsync(dgram, 0, sizeof(unsigned long long) * MAXGRAM, G_st));
          damar_launch_merge_count(&m, mw, dgram, MAXGRAM, G_st);          /* the same sweep, with the run histogram */
          HIP_CHECK(hipMemcpyAsync(histo.data(), dgram, sizeof(unsigned long long) * MAXGRAM, hipMemcpyDeviceToHost, G_st));
          HIP_CHECK(hipStreamSynchronize(G_st));
          HIP_CHECK(hipFree(dgram));
          int64 tom = 0;
          int   j;
          for (j = 0; j < MAXGRAM; j++)
            { tom += (int64) j * (int64) histo[j];
              if (tom > avail)
                break;
            }
          limit = j;
          if (limit <= 1)
            { fprintf(stderr, "\nError: Insufficient ");
              if (MEM_LIMIT == MEM_PHYSICAL)
                fprintf(stderr, " physical memory (%.1fGb), reduce block size\n", (1. * MEM_LIMIT) / 0x40000000ll);
              else
                { fprintf(stderr, " memory allocation (%.1fGb),", (1. * MEM_LIMIT) / 0x40000000ll);
                  fprintf(stderr, " reduce block size or increase allocation\n");
                }
              fflush(stderr);
              exit(1);
            }
          if (limit < 10)
            { fprintf(stderr, "\nWarning: Sensitivity hampered by low ");
              if (MEM_LIMIT == MEM_PHYSICAL)
                fprintf(stderr, " physical memory (%.1fGb), reduce block size\n", (1. * MEM_LIMIT) / 0x40000000ll);
              else
                { fprintf(stderr, " memory allocation (%.1fGb),", (1. * MEM_LIMIT) / 0x40000000ll);
                  fprintf(stderr, " reduce block size or increase allocation\n");
                }
              fflush(stderr);
            }
          /* count again with the lower cap */
          m.limit = (u32) limit;
          damar_launch_merge_count(&m, mw, NULL, 0, G_st);
          damar_exclusive_scan_u32(tcount, tcount, mtiles, mscw, tot, G_st);
          HIP_CHECK(hipMemcpyAsync(&total, tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
          HIP_CHECK(hipStreamSynchronize(G_st));
        }
      G_limit = limit;
      if (VERBOSE)
        printf("\n   Capping mutual k-mer matches over %d (effectively -t%d)\n", limit, (int) sqrt(1. * limit));
    }
  else
    G_limit = 0x7fffffff;
  nhits = (int64) total;
  if (VERBOSE)
    { printf("   Hit count = %lld\n", (long long) nhits);
      fflush(stdout);
    }
  if (total >= 0xfffffff0ull)
    { fprintf(stderr, "damar: FATAL: %llu seed pairs exceed the 32-bit seed index of this build\n",
              (unsigned long long) total);
      die();
    }
  if (total == 0)
    return false;

  /* hits known.  The sorted seed pairs stay in this comparison's own arena until the report launch; everything else
     of the seed stage lives in arenas the comparisons share.  The sort ping-pongs: it is started from the side that
     makes the result land in the comparison's arena (the number of passes is known). */
  const int sbits = m.pbits + m.abits + bbits;
  const int spasses = (sbits + 7) / 8;                        /* sort_scan.hip: 8 bits per pass */
  const int minhit = (P_hitmin - 1) / P_kmer + 1;
  const int idbits = m.abits + bbits;                         /* a read pair as one number: bread << abits | aread */
  static int cut_on = -1;
  if (cut_on < 0)
    { const char *e = getenv("DAMAR_EARLY_CUT");
      cut_on = e ? atoi(e) : 0;      /* measured on config 2: 38 % of the seed pairs belong to read pairs with >= 3 seeds
                                        (76.4 M seeds, 42.9 M read pairs, 5.8 M of them with >= 3), so the cut saves 3.7 of
                                        the 6 sort passes but costs a 4-pass sort of the pair ids and two more passes over
                                        the seeds: merge + sort + work list 203 -> 209 ms per step.  Off unless asked for;
                                        with a B-read range (a pair split over GPUs) it keeps only that range's seeds. */
    }
  const bool ranged = P_bread_lo > 0 || P_bread_hi != 0xffffffffu;      /* one part of a block pair split over GPUs */
  const bool cut = (cut_on || ranged) && !G_keep_seeds && idbits <= 32;
  static int two_step = -1;                              /* DAMAR_WORK_TWOSTEP=1: heads, then their screen (rounds 1-5; tested) */
  if (two_step < 0)
    { const char *e = getenv("DAMAR_WORK_TWOSTEP");
      two_step = (e && atoi(e) > 0) ? 1 : 0;
    }
  /* The seed sort over the READ PAIR only (its abits + bbits of the key's sbits: 4 passes instead of 6).  The seeds of a
     pair then lie in index order; the screen of the run heads takes them in any order and the runs of the kept heads -- all
     the report kernel walks -- are put in order of their A positions where they lie (kernels/seed_merge.hip order_runs).
     Not when something else reads the seeds (the tests' seed list, the two-step work list), not for the unpacked layout.
     It pays where the seed stages are the longer side: first 300 block pairs of config 4 1.17 -> 1.11 s; config 2 gains
     nothing (its two passes ran in the shadow of the report kernel) and config 3 loses 4 % (ordering 9 500 runs of a few
     hundred seeds beside a report kernel that is the longer side there) -- so, like the report launch's shape (report_launch),
     it goes by the seed pairs per work item of the comparison before this one: more than DAMAR_ADAPT_RATIO (4 000).
     DAMAR_SORT_PAIR=0: never (the sort over all the bits, rounds 1-6), 1: always (both tested). */
  static int pair_sort = -1;
  static long long pair_ratio = 4000;
  static u64 prev_seeds = 0, prev_work = 1;              /* of the comparison before this one */
  if (pair_sort < 0)
    { const char *e = getenv("DAMAR_SORT_PAIR"), *r = getenv("DAMAR_ADAPT_RATIO");
      pair_sort = e ? atoi(e) : 2;
      if (r)
        pair_ratio = atoll(r);
    }
  const bool pair_on = pair_sort == 1 || (pair_sort == 2 && pair_ratio > 0 && prev_seeds / prev_work > (u64) pair_ratio);
  bool psort = pair_on && !cut && !two_step && !G_keep_seeds && m.dbits != 0 && m.pbits + 11 <= 32 && m.pbits >= 8;
  const int spasses_used = psort ? (sbits - m.pbits + 7) / 8 : spasses;
  u64 *keys, *tk;
  u32 *vals, *flags, *foff;
  void *scw2, *resort_ws = NULL;
  u64 *sends;
  int  hshift = P_nshift;                                     /* slices of the reference's threads in the head test */
  if (!cut)
    { arena_reserve(&G_hits, pad256(sizeof(u64) * (size_t) total) + pad256(sizeof(u32) * (size_t) total) + 4096);
      arena_reserve(&G_tmp,  pad256(sizeof(u64) * (size_t) total) + 3 * pad256(sizeof(u32) * (size_t) total) +
                             pad256(damar_sort_workspace_bytes(total)) + pad256(damar_scan_workspace_bytes(total)) + 8192);
      u64 *pk = (u64 *) arena_take(&G_hits, sizeof(u64) * (size_t) total);
      u32 *pv = m.dbits ? NULL : (u32 *) arena_take(&G_hits, sizeof(u32) * (size_t) total);
      tk = (u64 *) arena_take(&G_tmp, sizeof(u64) * (size_t) total);
      u32 *tv = m.dbits ? NULL : (u32 *) arena_take(&G_tmp, sizeof(u32) * (size_t) total);
      u64 *k0 = (spasses_used & 1) ? tk : pk, *k1 = (spasses_used & 1) ? pk : tk;
      u32 *v0 = (spasses_used & 1) ? tv : pv, *v1 = (spasses_used & 1) ? pv : tv;
      void *sw = arena_take(&G_tmp, damar_sort_workspace_bytes(total));
      resort_ws = sw;
      flags = (u32 *) arena_take(&G_tmp, sizeof(u32) * (size_t) total);
      foff  = (u32 *) arena_take(&G_tmp, std::max(sizeof(u32) * (size_t) total, bit_words_bytes(total)));   /* (also the heads' bit words) */
      scw2  = arena_take(&G_tmp, damar_scan_workspace_bytes(total));
      sends = (u64 *) arena_take(&G_tmp, 64 * sizeof(u64));

      damar_launch_merge_emit(&m, mw, total, k0, v0, NULL, G_st);
      stage("merge_emit");
      tick(1);
      int side = m.dbits ? damar_radix_sort_keys_u64(k0, k1, total, m.dbits + (psort ? m.pbits : 0), m.dbits + sbits, sw, G_st)
                         : damar_radix_sort_u64(k0, v0, k1, v1, total, sbits, sw, G_st);
      sort_check(sw);
      keys = side ? k1 : k0;
      vals = side ? v1 : v0;
      if (keys != pk)
        { fprintf(stderr, "damar: internal error, the seed sort ended on the wrong side\n");
          die();
        }
    }
  else
    { /* The early cut (kernels/seed_merge.hip): only the seeds of the read pairs report_thread enters go through the
         seed sort.  The pair ids are sorted on their own (4 B per seed instead of 12, 4 passes instead of 6), the
         reference's head test runs on them, and the seeds of the surviving pairs -- a few per cent -- are compacted
         out of the unsorted seeds. */
      const size_t bmwords = (((size_t) 1 << idbits) + 31) / 32;
      arena_reserve(&G_tmp, pad256(sizeof(u64) * (size_t) total) + 4 * pad256(sizeof(u32) * (size_t) total) +
                            pad256(sizeof(u32) * ((size_t) total / minhit + 64)) +
                            pad256(damar_sort_workspace_bytes(total)) + pad256(damar_scan_workspace_bytes(total)) +
                            pad256(sizeof(u32) * bmwords) + 16384);
      u64 *uk   = (u64 *) arena_take(&G_tmp, sizeof(u64) * (size_t) total);
      u32 *uv   = (u32 *) arena_take(&G_tmp, sizeof(u32) * (size_t) total);
      u32 *pid0 = (u32 *) arena_take(&G_tmp, sizeof(u32) * (size_t) total);
      u32 *pid1 = (u32 *) arena_take(&G_tmp, sizeof(u32) * (size_t) total);
      u32 *hbit = (u32 *) arena_take(&G_tmp, sizeof(u32) * (size_t) total);          /* head bits: total / 8 bytes used */
      u32 *hd   = (u32 *) arena_take(&G_tmp, sizeof(u32) * ((size_t) total / minhit + 64));
      void *sw  = arena_take(&G_tmp, damar_sort_workspace_bytes(total));
      void *scc = arena_take(&G_tmp, damar_scan_workspace_bytes(total));
      u32 *bitmap = (u32 *) arena_take(&G_tmp, sizeof(u32) * bmwords);
      u64 *snd  = (u64 *) arena_take(&G_tmp, 64 * sizeof(u64));

      damar_launch_merge_emit(&m, mw, total, uk, uv, pid0, G_st);
      stage("merge_emit");
      tick(1);
      const u32 *spid = damar_radix_sort_keys_u32(pid0, pid1, total, idbits, sw, G_st) ? pid1 : pid0;
      sort_check(sw);
      u64 n64 = 0;
      damar_launch_pair_heads_ids(spid, total, m.abits, minhit, P_nshift, snd, (u64 *) hbit, scc, tot, hd, G_st);
      HIP_CHECK(hipMemsetAsync(bitmap, 0, sizeof(u32) * bmwords, G_st));
      HIP_CHECK(hipMemcpyAsync(&n64, tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
      HIP_CHECK(hipStreamSynchronize(G_st));
      damar_launch_pair_bitmap(spid, hd, (u32) n64, m.abits, P_bread_lo, P_bread_hi, bitmap, G_st);
      damar_launch_seed_cut_count(uk, total, m.pbits + m.dbits, bitmap, (u32 *) scc, tot, G_st);
      HIP_CHECK(hipMemcpyAsync(&n64, tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
      HIP_CHECK(hipStreamSynchronize(G_st));
      stage("early_cut");
      const u64 nsurv = n64;
      G_cnt[6] += (int64) nsurv;
      if (nsurv == 0)
        { tick(2);  tick(3);
          G_ms[DAMAR_T_MERGE] += lap(0, 1);
          G_ms[DAMAR_T_SSORT] += lap(1, 2);
          G_cnt[0] += nhits;
          job->counts[0] = nhits;
          return false;
        }
      arena_reserve(&G_hits, pad256(sizeof(u64) * (size_t) nsurv) + pad256(sizeof(u32) * (size_t) nsurv) + 4096);
      arena_reserve(&G_tmp2, pad256(sizeof(u64) * (size_t) nsurv) + 3 * pad256(sizeof(u32) * (size_t) nsurv) +
                             pad256(damar_sort_workspace_bytes(nsurv)) + pad256(damar_scan_workspace_bytes(nsurv)) + 8192);
      u64 *pk = (u64 *) arena_take(&G_hits, sizeof(u64) * (size_t) nsurv);
      u32 *pv = m.dbits ? NULL : (u32 *) arena_take(&G_hits, sizeof(u32) * (size_t) nsurv);
      tk = (u64 *) arena_take(&G_tmp2, sizeof(u64) * (size_t) nsurv);
      u32 *tv = m.dbits ? NULL : (u32 *) arena_take(&G_tmp2, sizeof(u32) * (size_t) nsurv);
      u64 *k0 = (spasses & 1) ? tk : pk, *k1 = (spasses & 1) ? pk : tk;
      u32 *v0 = (spasses & 1) ? tv : pv, *v1 = (spasses & 1) ? pv : tv;
      void *sw2 = arena_take(&G_tmp2, damar_sort_workspace_bytes(nsurv));
      flags = (u32 *) arena_take(&G_tmp2, sizeof(u32) * (size_t) nsurv);
      foff  = (u32 *) arena_take(&G_tmp2, std::max(sizeof(u32) * (size_t) nsurv, bit_words_bytes(nsurv)));
      scw2  = arena_take(&G_tmp2, damar_scan_workspace_bytes(nsurv));
      sends = (u64 *) arena_take(&G_tmp2, 64 * sizeof(u64));
      damar_launch_seed_cut_scatter(uk, m.dbits ? NULL : uv, total, m.pbits + m.dbits, bitmap, (const u32 *) scc, k0, v0, G_st);
      int side = m.dbits ? damar_radix_sort_keys_u64(k0, k1, nsurv, m.dbits, m.dbits + sbits, sw2, G_st)
                         : damar_radix_sort_u64(k0, v0, k1, v1, nsurv, sbits, sw2, G_st);
      sort_check(sw2);
      keys = side ? k1 : k0;
      vals = side ? v1 : v0;
      if (keys != pk)
        { fprintf(stderr, "damar: internal error, the seed sort ended on the wrong side\n");
          die();
        }
      total  = nsurv;              /* from here on: the seeds of the read pairs that are entered, nothing else */
      hshift = -1;                 /* (the slice rule of the head test has been applied on the pair ids) */
    }
  stage("seed_sort");
  tick(2);

  /* ---- work list ---- */
  u64 nwork64 = 0;
  u32 *work = NULL;
  if (!two_step)
    { /* heads and screen in one pass over the seeds: a bit per seed, the work list expanded from the bits once its
         length is known (the work list and its processing order outlive the seed stage: the comparison's second arena) */
      u64 got[2] = { 0, 0 };                               /* work items; a run order_runs would not sort */
      if (cut)
        psort = false;                                     /* (the early cut's survivors were sorted over all the bits) */
      damar_launch_pair_work(keys, vals, total, m.pbits, m.dbits, m.abits, minhit, hshift, sends, (u64 *) foff /* bit words */,
                             scw2, tot, P_binshift, P_kmer, P_hitmin, P_bread_lo, P_bread_hi, psort ? 1 : 0, G_st);
      stage("run_heads");
      tick(3);                                             /* (the expansion of the bits behind it is 4 us: outside the clock) */
      HIP_CHECK(hipMemcpyAsync(got, tot, 2 * sizeof(u64), hipMemcpyDeviceToHost, G_st));
      stream_wait(G_st);                                   /* the comparison's last wait: nothing below needs the host again */
      if (psort && got[1] != 0)
        { /* a run of more than 2048 seeds among the kept ones (a tandem array against itself, a satellite): this comparison's
             seeds are sorted over all the bits after all -- from where they lie, the order of equal keys has not changed --
             and the work list is made again */
          G_cnt[7] += 1;
          u64 *other = (keys == tk) ? NULL : tk;
          if (other == NULL || resort_ws == NULL)
            { fprintf(stderr, "damar: internal error, no room to sort the seeds again\n");
              die();
            }
          int side = damar_radix_sort_keys_u64(keys, other, total, m.dbits, m.dbits + sbits, resort_ws, G_st);
          sort_check(resort_ws);
          if (side)
            HIP_CHECK(hipMemcpyAsync(keys, other, sizeof(u64) * (size_t) total, hipMemcpyDeviceToDevice, G_st));
          damar_launch_pair_work(keys, vals, total, m.pbits, m.dbits, m.abits, minhit, hshift, sends, (u64 *) foff, scw2, tot,
                                 P_binshift, P_kmer, P_hitmin, P_bread_lo, P_bread_hi, 0, G_st);
          HIP_CHECK(hipMemcpyAsync(got, tot, 2 * sizeof(u64), hipMemcpyDeviceToHost, G_st));
          stream_wait(G_st);
          psort = false;
        }
      nwork64 = got[0];
      arena_reserve(&G_ord, 5 * pad256(sizeof(u32) * (size_t) nwork64) + pad256(damar_sort_workspace_bytes(nwork64)) + 8192);
      work = (u32 *) arena_take(&G_ord, sizeof(u32) * ((size_t) nwork64 + 1));
      if (nwork64 > 0)
        { damar_launch_pair_work_expand((const u64 *) foff, scw2, total, work, G_st);
          if (psort)                                       /* the runs the report kernel will walk, in the order of their A positions */
            damar_launch_order_runs(keys, total, m.pbits, m.dbits, work, (u32) nwork64, G_st);
        }
      stage("work_list");
    }
  else
    { u32 *heads = (u32 *) tk;                              /* the idle key buffer holds the run heads */
      damar_launch_pair_heads(keys, total, m.pbits + m.dbits, m.abits, minhit, hshift, sends, (u64 *) foff /* bit words */,
                              scw2, tot, heads, G_st);
      stage("run_heads");
      HIP_CHECK(hipMemcpyAsync(&nwork64, tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
      HIP_CHECK(hipStreamSynchronize(G_st));
      const u32 nheads = (u32) nwork64;
      /* screen the heads (dense, one thread each), compact the survivors: flags/foff are reused */
      arena_reserve(&G_ord, 5 * pad256(sizeof(u32) * (size_t) nheads) + pad256(damar_sort_workspace_bytes(nheads)) + 8192);
      work = (u32 *) arena_take(&G_ord, sizeof(u32) * ((size_t) nheads + 1));
      nwork64 = 0;
      if (nheads > 0)
        { damar_launch_pair_screen(keys, vals, total, m.pbits, m.dbits, heads, nheads, minhit, P_binshift, P_kmer, P_hitmin, m.abits,
                                   P_bread_lo, P_bread_hi, flags, G_st);
          damar_exclusive_scan_u32(flags, foff, nheads, scw2, tot, G_st);
          damar_launch_compact_u32(heads, flags, foff, nheads, work, G_st);
          stage("work_list");
          HIP_CHECK(hipMemcpyAsync(&nwork64, tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
        }
    }
  if (two_step)
    { tick(3);
      HIP_CHECK(hipStreamSynchronize(G_st));
    }
  sort_verify();
  const u32 nwork = (u32) nwork64;
  G_ms[DAMAR_T_MERGE] += lap(0, 1);
  G_ms[DAMAR_T_SSORT] += lap(1, 2);
  G_ms[DAMAR_T_WORK]  += lap(2, 3);
  G_cnt[0] += nhits;  G_cnt[1] += nwork;
  job->counts[0] = nhits;
  prev_seeds = total;  prev_work = nwork > 0 ? nwork : 1;

  if (G_keep_seeds)
    { G_seed_keys.resize(total);  G_seed_vals.resize(total);
      HIP_CHECK(hipMemcpy(G_seed_keys.data(), keys, sizeof(u64) * (size_t) total, hipMemcpyDeviceToHost));
      if (vals != NULL)
        HIP_CHECK(hipMemcpy(G_seed_vals.data(), vals, sizeof(u32) * (size_t) total, hipMemcpyDeviceToHost));
      G_seed_pbits = m.pbits;  G_seed_abits = m.abits;  G_seed_dbits = m.dbits;
    }

  /* ---- largest pairs first (the order only schedules the kernel: records carry their work
          item's rank in the reference's order) ---- */
  const u32 *order = NULL;
  static int order_mode = -1;
  if (order_mode < 0)
    { const char *e = getenv("DAMAR_ORDER");
      order_mode = e ? atoi(e) : 11;     /* 0 = reference order, 1 = most seeds first, 2 = size classes, 9 = longest seed
                                            extent first, 10 = longest geometric overlap first, 11 = the larger of the two
                                            (default: 356 ms of report kernel per config-2 step against 370 for 1, 361 for
                                            9, 411 for 10), other n = runs of >= n seeds first */
    }
  if (nwork > 1 && order_mode > 0)
    { u32 *ok0 = (u32 *) arena_take(&G_ord, sizeof(u32) * (size_t) nwork);
      u32 *ov0 = (u32 *) arena_take(&G_ord, sizeof(u32) * (size_t) nwork);
      u32 *ok1 = (u32 *) arena_take(&G_ord, sizeof(u32) * (size_t) nwork);
      u32 *ov1 = (u32 *) arena_take(&G_ord, sizeof(u32) * (size_t) nwork);
      void *osw = arena_take(&G_ord, damar_sort_workspace_bytes(nwork));
      const u32 cmode = order_mode == 2 ? 1u : order_mode == 9 ? 0xffffffffu : order_mode == 10 ? 0xfffffffeu :
                        order_mode == 11 ? 0xfffffffdu : order_mode == 12 ? 0xfffffffcu : (order_mode > 2 ? (u32) order_mode : 0u);
      damar_launch_work_cost(keys, vals, total, m.pbits, m.abits, m.dbits, m.ablk.boff, m.bblk.boff, work, nwork, cmode, ok0, ov0, G_st);
      order = damar_radix_sort_u32(ok0, ov0, ok1, ov1, nwork, WORK_COST_BITS, osw, G_st) ? ov1 : ov0;
      sort_check(osw);
      stage("work_order");
    }


  f->keys = keys;  f->vals = vals;  f->total = total;
  f->work = work;  f->nwork = nwork;  f->order = order;
  f->pbits = m.pbits;  f->abits = m.abits;  f->dbits = m.dbits;
  f->bytes = G_hits.cap + G_ord.cap;
  return nwork > 0;
}

/* A report launch over up to DAMAR_MAX_JOBS comparisons, in two halves.  In asynchronous mode the launch goes to its own
   stream and is completed only when the NEXT launch is due (or at a drain): the kernel is bound by instruction issue and
   dependent latency, the seed stage of the following comparisons by HBM, and the two share the machine (tools/corun.hip:
   a register-heavy persistent kernel and a streaming kernel finish together in 80 % of the time they take one after the
   other).  Up to two launches are in flight -- one running, one queued behind it on the stream, so that the device goes
   from one to the next without the host -- while the seed stages of later comparisons accumulate; every comparison holds
   one of DAMAR_MAX_JOBS slots (arenas of its sorted seeds and work list, its SCORE/TABLE copy) from its seed stage until
   its launch has been completed. */
struct Pending
{ bool live;
  int  n;
  int  slot[DAMAR_MAX_JOBS];
  damar_match_job  job[DAMAR_MAX_JOBS];           /* copies: the caller's array may be gone when the launch completes */
  damar_match_job *orig[DAMAR_MAX_JOBS];          /* the caller's structs (counts) while its call is still running */
  Front fr[DAMAR_MAX_JOBS];
  const damar_dev_block *ablk[DAMAR_MAX_JOBS], *bblk[DAMAR_MAX_JOBS];     /* (an index may be released before a re-launch) */
  int  amax, bmax, tsmin;
  u32  cell_cap, rec_cap, tp_cap;
  int  t8;                                        /* the launch's trace values leave the device as bytes (ReportArgs.t8) */
  bool wide_ok;                                   /* the two-pair kernel runs: pairs beyond the packed pebble format go to the wide kernel */
  bool wide_done;                                 /* the wide kernel has run behind this attempt's launch */
  u32  wm_off[DAMAR_MAX_JOBS + 1];                /* the jobs' bit maps in RS.widemap (words) */
  int  attempt;
  int  oset;                                      /* the set of record buffers, counters and timers this launch uses */
  hipEvent_t done;                                /* behind the kernel */
  hipStream_t st;
  double t_launch;
  ReportArgs ra[DAMAR_MAX_JOBS];                  /* what the launch was given (the upload is asynchronous) */
  std::vector<TailJob *> writes;                  /* damar_write_overlaps requests that wait for this launch's tails */
};
static Pending *PQ = new Pending[2]();               /* PQ[i] uses output set i */
static int      PQ_head = 0, PQ_n = 0;             /* the oldest launch in flight, how many there are */
static bool     G_slot_busy[DAMAR_MAX_JOBS];

/* Comparisons whose seed stages are done and whose report launch has not been made yet.  In asynchronous mode a launch is
   held back while it would be SMALL: a launch ends by waiting for its longest alignment (a 15 kb read pair is some 10 ms of
   serial wave steps), so a launch over two comparisons of a sparse block pair (config 4: ~1 600 alignments each) costs as
   much as one over sixteen.  The comparisons of later calls join it until it holds batch_work() read pairs or its set of job
   arenas is full; the files of the block pairs in it wait with it (damar_write_overlaps). */
struct Accum
{ int    n;
  int    slot[DAMAR_MAX_JOBS / 2];
  damar_match_job  job[DAMAR_MAX_JOBS / 2];
  damar_match_job *orig[DAMAR_MAX_JOBS / 2];
  Front  fr[DAMAR_MAX_JOBS / 2];
  const damar_dev_block *ablk[DAMAR_MAX_JOBS / 2], *bblk[DAMAR_MAX_JOBS / 2];
  size_t bytes;
  u64    nwork;
  std::vector<TailJob *> writes;
};
static Accum &AC = *new Accum();


static int64   A_nfilt = 0;                        /* totals of the asynchronous mode (damar_async_counts) */
static int64   W_tot[3] = { 0, 0, 0 };             /* band cells, wave steps per pass, wave-loop iterations (damar_wave_totals) */
static double  A_report_ms = 0;
static int64   A_launches = 0;

static void report_launch(Pending &pd)
{ ReportArgs *const ra = pd.ra;
  const hipStream_t st = pd.st;
  double q0 = now_ms();
  RS.cur = pd.oset;
  scratch_prepare(pd.amax, pd.bmax, pd.fr[0].jp.binshift, pd.tsmin, pd.cell_cap, st);
  double q1 = now_ms();
  scratch_outputs(pd.rec_cap, pd.tp_cap);
  double q2 = now_ms();
  Q_ms[0] += q1 - q0;  Q_ms[1] += q2 - q1;
  bool packed = true;
  for (int j = 0; j < pd.n; j++)
    { const damar_match_job &jb = pd.job[j];
      fill_report_args(&ra[j], pd.ablk[j], pd.bblk[j], jb.comp, jb.self, jb.spec, st, j, pd.slot[j], pd.fr[j].jp);
      ra[j].keys = pd.fr[j].keys;  ra[j].vals = pd.fr[j].vals;  ra[j].nhits = pd.fr[j].total;
      ra[j].work = pd.fr[j].work;  ra[j].nwork = pd.fr[j].nwork;
      ra[j].pbits = pd.fr[j].pbits;  ra[j].abits = pd.fr[j].abits;  ra[j].dbits = pd.fr[j].dbits;
      ra[j].order = pd.fr[j].order;
      ra[j].t8 = pd.t8;
      ra[j].widemap = NULL;  ra[j].wcells = RS.wcells;  ra[j].wcell_cap = RS.wcell_cap;  ra[j].cell_max = max_cells();
      packed = packed && use_packed(&ra[j], pd.amax, pd.bmax);
      if (ra[j].mscore != ra[0].mscore || ra[j].dscore != ra[0].dscore)
        { fprintf(stderr, "damar: internal error, the comparisons of one report launch differ in their -e\n");
          die();
        }
    }
  pd.wide_ok = packed;  pd.wide_done = false;
  if (!packed && std::max(pd.amax, pd.bmax) / std::max(1, pd.tsmin) + 8 > DAMAR_MAX_MARKS)
    { fprintf(stderr, "damar: FATAL: reads of %d bases need a trace spacing (-s) of at least %d when one read pair per wavefront "
                      "is asked for (DAMAR_PACKED=0): the wide kernel runs behind the two-pair kernel only\n",
              std::max(pd.amax, pd.bmax), std::max(pd.amax, pd.bmax) / (DAMAR_MAX_MARKS - 8) + 1);
      die();
    }
  if (packed)
    { /* one bit per work item and job, zero at every launch: which read pairs are the wide kernel's (kernels/report.hip) */
      u32 words = 0;
      for (int j = 0; j < pd.n; j++)
        { pd.wm_off[j] = words;
          words += (ra[j].nwork + 31) / 32 + 1;
        }
      pd.wm_off[pd.n] = words;
      const int c = pd.oset;
      if (RS.widemap_cap[c] < words)
        { if (RS.widemap[c]) HIP_CHECK(hipFree(RS.widemap[c]));
          RS.widemap_cap[c] = words + (words >> 2) + 1024;
          RS.widemap[c] = (u32 *) dmalloc(sizeof(u32) * (size_t) RS.widemap_cap[c]);
        }
      HIP_CHECK(hipMemsetAsync(RS.widemap[c], 0, sizeof(u32) * (size_t) words, st));
      for (int j = 0; j < pd.n; j++)
        ra[j].widemap = RS.widemap[c] + pd.wm_off[j];
    }
  HIP_CHECK(hipMemsetAsync(RS.ctr, 0, sizeof(u32) * DAMAR_COUNTER_WORDS, st));
  tick_on(16 + 4 * pd.oset, st);
  if (packed)
    { /* How much of the machine a launch takes (r6).  A launch that fills every wave slot keeps the seed stream's next
         kernel waiting until its wavefronts retire; where the seed stages are the longer side -- a plan of many sparse block
         pairs: config 4 has 9 000 seed pairs per read pair that reaches the kernel, config 2 1 300, config 3 300 -- the
         launch runs on DAMAR_ADAPT_WPS wavefronts per SIMD (3) instead of all 8: it takes a quarter longer and the seed
         kernels run beside it the whole time (first 300 block pairs of config 4: 1.69 -> 1.58 s, profiles/r06_sweeps.txt).
         DAMAR_ADAPT_RATIO = seed pairs per work item above which that happens (4 000; 0 = never). */
      static long long ratio = -1;
      static int wps = 0;
      if (ratio < 0)
        { const char *e = getenv("DAMAR_ADAPT_RATIO"), *w = getenv("DAMAR_ADAPT_WPS");
          ratio = e ? atoll(e) : 4000;
          wps = w ? atoi(w) : 3;
          if (wps < 1) wps = 1;
        }
      int nslots = RS.nslots;
      if (ratio > 0 && corun_on() && pd.st == G_rep)
        { u64 seeds = 0, work = 0;
          for (int j = 0; j < pd.n; j++)
            { seeds += pd.fr[j].total;  work += pd.fr[j].nwork; }
          const int low = G_prop.multiProcessorCount * 4 * damar_report2_slots_per_wave() * wps;
          if (work > 0 && seeds / work > (u64) ratio && low < nslots)
            nslots = low;
        }
      damar_launch_report2(ra, pd.n, NULL, 0, nslots, st);
    }
  else
    damar_launch_report(ra, pd.n, RS.nslots, st);
  tick_on(17 + 4 * pd.oset, st);
  if (pd.done == NULL)
    HIP_CHECK(hipEventCreateWithFlags(&pd.done, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(pd.done, st));
  pd.live = true;
  pd.t_launch = now_ms();
}

/* Completes the launch in flight: waits for it, re-launches with larger buffers after an overflow, starts the download
   of the records and hands the comparisons to the host tail in job order (filter.c:2442-2483 per read pair). */
static void report_finish(Pending &pd)
{ if (!pd.live)
    return;
  u32 hc[DAMAR_COUNTER_WORDS];
  const hipStream_t st = pd.st;
  const int n = pd.n;
  const double h2 = now_ms();
  for (;;)
    { /* this launch only: a younger one may be queued behind it on the same stream */
      HIP_CHECK(hipEventSynchronize(pd.done));
      late_streams();
      HIP_CHECK(hipMemcpyAsync(hc, RS.counters + (size_t) pd.oset * DAMAR_COUNTER_WORDS, sizeof(hc), hipMemcpyDeviceToHost, G_ctl));
      HIP_CHECK(hipStreamSynchronize(G_ctl));
      HIP_CHECK(hipGetLastError());
      { const float ms = lap(16 + 4 * pd.oset, 17 + 4 * pd.oset);
        G_ms[DAMAR_T_REPORT] += ms;
        if (A_on)
          { std::lock_guard<std::mutex> lk(A_mu);
            A_report_ms += ms;
            A_launches += 1;
          }
      }
      G_cnt[5] += 1;
      /* Read pairs beyond the packed pebble format (reads of more than DAMAR_MAX_MARKS trace spacings; alignments that
         overflowed the pebble pool at its largest, 2^18): the wide kernel takes them, into the same record buffers, behind
         the launch that is otherwise complete.  Until round 5 both were fatal. */
      if (pd.wide_ok && !pd.wide_done && hc[DAMAR_CNT_WIDE] > 0)
        { u32 flags = hc[3];
          if (pd.cell_cap >= max_cells())
            flags &= ~DAMAR_ERR_CELLS;                         /* (those pairs are in the map: nothing to repeat for them) */
          if (flags == 0)
            { if (RS.wcells == NULL)
                { RS.wslots = std::min(RS.nslots, 64);
                  RS.wcell_cap = 1u << 21;
                  RS.wcells = dmalloc((size_t) 16 * RS.wcell_cap * (size_t) RS.wslots);
                }
              for (int j = 0; j < n; j++)
                { pd.ra[j].wcells = RS.wcells;  pd.ra[j].wcell_cap = RS.wcell_cap; }
              if (VERBOSE)
                fprintf(stderr, "damar: %u read pair(s) beyond the packed pebble format: wide kernel\n", hc[DAMAR_CNT_WIDE]);
              { u32 *const ctr = RS.counters + (size_t) pd.oset * DAMAR_COUNTER_WORDS;
                HIP_CHECK(hipMemsetAsync(ctr + DAMAR_CNT_CURSOR, 0, sizeof(u32) * DAMAR_MAX_JOBS, st));      /* the work lists once more */
                HIP_CHECK(hipMemsetAsync(ctr + 3, 0, sizeof(u32), st));
              }
              damar_launch_report_wide(pd.ra, n, RS.wslots, st);
              HIP_CHECK(hipEventRecord(pd.done, st));
              pd.wide_done = true;
              continue;                                         /* wait for it, read the counters again */
            }
        }
      if (pd.wide_done && (hc[3] & DAMAR_ERR_CELLS))
        { /* the wide kernel's own pool: four times the pebbles, and everything once more */
          if (RS.wcell_cap >= (1u << 26))
            { fprintf(stderr, "damar: FATAL: an alignment needs more than %u trace pebbles even in the wide format\n", RS.wcell_cap);
              die();
            }
          HIP_CHECK(hipStreamSynchronize(st));
          HIP_CHECK(hipFree(RS.wcells));
          RS.wcell_cap *= 4;
          RS.wslots = std::max(8, RS.wslots / 2);
          RS.wcells = dmalloc((size_t) 16 * RS.wcell_cap * (size_t) RS.wslots);
          hc[3] &= ~DAMAR_ERR_CELLS;
          hc[3] |= 0x40000000u;                                /* (something to repeat the launch for) */
        }
      else if (pd.wide_ok && pd.cell_cap >= max_cells() && hc[DAMAR_CNT_WIDE] > 0)
        hc[3] &= ~DAMAR_ERR_CELLS;                             /* the packed kernel's overflows at 2^18 are the wide kernel's pairs */
      if (hc[3] == 0)
        break;
      hc[3] &= ~0x40000000u;
      if ((hc[3] & DAMAR_ERR_BAND) && !(hc[3] & (DAMAR_ERR_CELLS | DAMAR_ERR_WIDE)))
        { fprintf(stderr, "damar: FATAL: a Local_Alignment wave exceeded its loop bound (where=%u)\n", hc[6]);
          die();
        }
      if (pd.attempt >= 6)
        { fprintf(stderr, "damar: FATAL: report kernel keeps overflowing its buffers (flags %u)\n", hc[3]);
          die();
        }
      pd.attempt += 1;
      if (hc[3] & DAMAR_ERR_CELLS) pd.cell_cap = grow_cells(pd.cell_cap);
      if (hc[3] & DAMAR_ERR_WIDE)  G_ring *= 4;
      if (hc[3] & DAMAR_ERR_T8)    pd.t8 = 0;     /* a value above 255: 16-bit values again, and the reference's check on what is written */
      if (hc[3] & DAMAR_ERR_RECS)  pd.rec_cap = std::max(2 * pd.rec_cap, hc[1] + 1024);
      if (hc[3] & DAMAR_ERR_TPOOL)
        { if (pd.tp_cap >= 0xe0000000u)
            { fprintf(stderr, "damar: FATAL: more than 2^32 trace values in one comparison, use smaller blocks\n");
              die();
            }
          pd.tp_cap = (u32) std::min<u64>(0xe0000000ull, std::max<u64>(2ull * pd.tp_cap, (u64) hc[2] + 65536));
        }
      if (VERBOSE)
        fprintf(stderr, "damar: report kernel overflow (flags %u), retrying with larger buffers\n", hc[3]);
      report_launch(pd);
    }
  pd.live = false;
  tick_on(18 + 4 * pd.oset, st);
  const double h3 = now_ms();
  HostBuf *hb = hostbuf_get(hc[1], hc[2]);
  hb->users = n;
  late_streams();
  hipStream_t cs = A_on ? G_copy : st;
  if (A_on)
    HIP_CHECK(hipEventRecord(hb->e0, cs));       /* (the report kernel has completed: the host synced on it) */
  if (hc[1] > 0)
    { HIP_CHECK(hipMemcpyAsync(hb->recs, RS.recs_set[pd.oset], sizeof(LaRecord) * (size_t) hc[1], hipMemcpyDeviceToHost, cs));
      HIP_CHECK(hipMemcpyAsync(hb->tpool, RS.tpool_set[pd.oset], (pd.t8 ? sizeof(u8) : sizeof(u16)) * (size_t) hc[2], hipMemcpyDeviceToHost, cs));
    }
  hb->t8 = pd.t8;
  hb->nwide = pd.wide_done ? hc[DAMAR_CNT_WIDE] : 0;
  if (hb->nwide > 0)
    { const u32 words = pd.wm_off[n];
      if (hb->wmap_cap < words)
        { free(hb->wmap);
          hb->wmap_cap = words + 1024;
          hb->wmap = (u32 *) malloc(sizeof(u32) * hb->wmap_cap);
        }
      memcpy(hb->wm_off, pd.wm_off, sizeof(hb->wm_off));
      HIP_CHECK(hipMemcpy(hb->wmap, RS.widemap[pd.oset], sizeof(u32) * (size_t) words, hipMemcpyDeviceToHost));
    }
  tick_on(19 + 4 * pd.oset, st);
  if (A_on)
    { /* asynchronous mode: the download runs on its own stream beside the next comparison's merge and sorts; the
         tail thread waits for it, and so does the next report kernel (which would overwrite the device buffers) */
      HIP_CHECK(hipEventRecord(hb->e1, cs));
      hb->pending = true;
      /* (an event of the SET, not hb->e1: the host buffer is recycled, and a wait on its event would mean its next
         download -- the one out of the other set that has just been started) */
      HIP_CHECK(hipEventRecord(G_set_d2h[pd.oset], cs));
      G_last_d2h[pd.oset] = G_set_d2h[pd.oset];
    }
  else
    { HIP_CHECK(hipStreamSynchronize(st));
      G_ms[DAMAR_T_D2H] += lap(18 + 4 * pd.oset, 19 + 4 * pd.oset);
    }
  G_cnt[3] += hc[1];  G_cnt[4] += hc[2];
  { std::lock_guard<std::mutex> lk(A_mu);
    W_tot[0] += (int64) (((u64) hc[DAMAR_CNT_CELLS + 1] << 32) | hc[DAMAR_CNT_CELLS]);
    W_tot[1] += (int64) (((u64) hc[DAMAR_CNT_HALFSTEPS + 1] << 32) | hc[DAMAR_CNT_HALFSTEPS]);
    W_tot[2] += (int64) (((u64) hc[DAMAR_CNT_ITERS + 1] << 32) | hc[DAMAR_CNT_ITERS]);
  }
  const double h4 = now_ms();
  for (int j = 0; j < n; j++)
    { const damar_match_job &jb = pd.job[j];
      if (pd.orig[j] != NULL)
        pd.orig[j]->counts[1] = hc[DAMAR_CNT_NFILT + j];
      G_cnt[2] += hc[DAMAR_CNT_NFILT + j];
      if (A_on)
        { { std::lock_guard<std::mutex> lk(A_mu);
            A_nfilt += hc[DAMAR_CNT_NFILT + j];
          }
          TailJob *tj = new TailJob();
          tj->kind = 0;
          tj->hb = hb;  tj->jobid = j;  tj->njobs = n;
          tj->ablock = *jb.ablock;  tj->bblock = *jb.bblock;
          tj->self = jb.self;  tj->comp = jb.comp;  tj->spec = jb.spec;  tj->jp = pd.fr[j].jp;
          tj->got = (G_tail_reports && pd.orig[j] != NULL) ? &pd.orig[j]->counts[2] : NULL;   /* (only for a caller that drains before its job struct goes: damar_match) */
          async_submit(tj);
        }
      else
        { double t0 = now_ms();
          const int64 got = run_tail(hb->recs, hb->nrec, hb->tpool, hb->t8, hb->nwide ? hb->wmap + hb->wm_off[j] : (const u32 *) NULL, jb.ablock, jb.bblock, jb.self, jb.comp, jb.spec, pd.fr[j].jp, j, n);
          if (pd.orig[j] != NULL)
            pd.orig[j]->counts[2] = got;
          if (--hb->users == 0)
            hostbuf_put(hb);
          G_ms[DAMAR_T_TAIL] += now_ms() - t0;
        }
    }
  for (TailJob *w : pd.writes)                   /* the files of these comparisons: behind their tails */
    async_submit(w);
  pd.writes.clear();
  const double h5 = now_ms();
  H_ms[3] += h3 - h2;  H_ms[4] += h4 - h3;  H_ms[5] += h5 - h4;
}

/* completes the oldest launch in flight and releases its comparisons' slots */
static void finish_oldest(void)
{ if (PQ_n == 0)
    return;
  Pending &pd = PQ[PQ_head];
  report_finish(pd);
  for (int j = 0; j < pd.n; j++)
    G_slot_busy[pd.slot[j]] = false;
  pd.n = 0;
  PQ_head ^= 1;
  PQ_n -= 1;
}

static void flush_accum(void);

/* a free comparison slot; when there is none, the oldest launch is completed (or, with nothing in flight, the comparisons
   held back are launched) */
static int slot_take(void)
{ for (;;)
    { for (int i = 0; i < DAMAR_MAX_JOBS; i++)
        if (!G_slot_busy[i])
          { G_slot_busy[i] = true;
            return i;
          }
      if (PQ_n > 0)
        finish_oldest();
      else
        flush_accum();
    }
}

static int batch_limit(void)
{ static int n = 0;
  if (n == 0)
    { const char *e = getenv("DAMAR_BATCH");
      n = e ? atoi(e) : 0;      /* report ms per config-2 step: 1 -> 357, 2 -> 340, 4 -> 335.5, 16 -> 335.3; every job in
                                   flight keeps its sorted seed pairs (12 B each) in HBM, which a cold process pays for
                                   at ~25 ms per GB */
      if (n == 0)
        return corun_on() ? 2 : 4;            /* overlapped launches: short ones interleave better (548 against 590 ms) */
      if (n < 1) n = 1;
      if (n > DAMAR_MAX_JOBS) n = DAMAR_MAX_JOBS;
    }
  return n;
}

/* Several comparisons with ONE report launch each time their seed stages are done (kernels.h: DAMAR_MAX_JOBS): the seed
   stages run one after the other, each into its own arena, then every wavefront of the report kernel works through all
   the work lists.  The output is that of damar_match called for jobs[0], jobs[1], ... in this order.  Launches are cut
   at DAMAR_BATCH jobs (default 4, at most DAMAR_MAX_JOBS) and whenever the seed arenas would pass a quarter of HBM. */
/* Write_Overlap_Buffer + Reset_Overlap_Buffer (daligner.c:1020-1021, 1055-1056), queued behind
 * the pending tails in asynchronous mode, immediate otherwise. */
extern "C" void damar_write_overlaps(Align_Spec *spec, const char *d1, const char *d2,
                                     const char *ablock, const char *bblock, int lastRead)
{ if (!A_on)
    { Write_Overlap_Buffer(spec, (char *) d1, (char *) d2, (char *) ablock, (char *) bblock, lastRead);
      Reset_Overlap_Buffer(spec);
      return;
    }
  TailJob *job = new TailJob();
  job->kind = 1;  job->spec = spec;
  job->has1 = d1 != NULL;  job->has2 = d2 != NULL;
  if (d1) job->d1 = d1;
  if (d2) job->d2 = d2;
  job->a = ablock;  job->b = bblock;  job->last = lastRead;
  for (int j = 0; j < AC.n; j++)                 /* comparisons of this spec are still waiting for their launch */
    if (AC.job[j].spec == spec)
      { AC.writes.push_back(job);
        return;
      }
  for (int q = PQ_n - 1; q >= 0; q--)            /* the tails of a launch in flight are not queued yet: behind them */
    { Pending &pd = PQ[(PQ_head + q) & 1];         /* (the youngest launch that holds the spec) */
      for (int j = 0; j < pd.n; j++)
        if (pd.job[j].spec == spec)
          { pd.writes.push_back(job);
            return;
          }
    }
  async_submit(job);
}

/* SURVEY 8(d)'s secondary unit of K6, counted by the packed report kernel itself (scalar adds in its wave loop): band
   cells (diagonals computed, summed over all wave steps), wave steps counted per alignment pass (= per half-wavefront),
   and iterations of the wave loop (each steps one or two halves) -- totals since the last call (drains first) */
extern "C" void damar_wave_totals(int64 *cells, int64 *half_steps, int64 *iterations)
{ damar_async_drain();
  std::lock_guard<std::mutex> lk(A_mu);
  if (cells)      *cells = W_tot[0];
  if (half_steps) *half_steps = W_tot[1];
  if (iterations) *iterations = W_tot[2];
  W_tot[0] = W_tot[1] = W_tot[2] = 0;
}

/* totals of the asynchronous mode since the last call (drains first): seed hits (what damar_match's counts[1] reports in
   the synchronous mode) and the report kernel's milliseconds */
extern "C" void damar_async_counts(int64 *nfilt, double *report_ms, int64 *launches)
{ damar_async_drain();
  std::lock_guard<std::mutex> lk(A_mu);
  if (nfilt)     *nfilt = A_nfilt;
  if (report_ms) *report_ms = A_report_ms;
  if (launches)  *launches = A_launches;
  A_nfilt = 0;  A_report_ms = 0;  A_launches = 0;
}

/* DAMAR_OVERLAP: 1 = the report launch runs BESIDE the next comparisons' seed stages (its own stream, 4 of 5 wavefronts per
   SIMD); 2 = it runs on its own stream but the next seed stages wait for it on the device: the kernels follow each other
   as in mode 0, while the host (completion of the launch, download, tails) stays pipelined; 0 = launch, wait, tail. */
static int overlap_mode(void)
{ static int on = -1;
  if (on < 0)
    { const char *e = getenv("DAMAR_OVERLAP");
      on = e ? atoi(e) : 1;
    }
  /* only behind the asynchronous host tail (nobody reads the records at return), and not when the caller wants the
     per-comparison counts printed (-v) or the seeds kept */
  return (A_on && !VERBOSE && !G_keep_seeds) ? on : 0;
}
static bool overlap_on(void) { return overlap_mode() != 0; }
static bool corun_on(void)   { return overlap_mode() == 1; }

static u64 batch_work(void)
{ static long long w = -1;
  if (w < 0)
    { const char *e = getenv("DAMAR_BATCH_WORK");
      w = e ? atoll(e) : 32768;           /* read pairs: four per wave slot of the launch */
    }
  return (u64) w;
}

static void flush_accum(void)
{ if (AC.n == 0)
    return;
  const bool defer = overlap_on();
  /* an entry of the launch queue, i.e. a set of output buffers: the oldest launch in flight is completed first when both
     are taken (it ran beside these seed stages) */
  /* How many launches may be in flight.  Two (the next one queued behind the running one) takes the host out of the
     hand-over, but measured SLOWER on config 2: 434 - 445 ms per step against 398 - 410 with one.  With one, the host
     waits here for the running launch, so the seed stream is idle during the last third of every launch -- the phase in
     which the kernel works through the many short read pairs, whose band filters hammer the bucket arrays; seed sorts that
     run beside that phase take 180 instead of 100 ms per step. */
  static int depth = 0;
  if (depth == 0)
    { const char *e = getenv("DAMAR_LAUNCH_QUEUE");
      depth = (e && atoi(e) == 2) ? 2 : 1;
    }
  while (PQ_n >= depth)
    finish_oldest();
  Pending &pd = PQ[(PQ_head + PQ_n) & 1];
  const int n = AC.n;
  pd.n = n;
  pd.oset = (int) (&pd - PQ);
  pd.amax = pd.bmax = 0;  pd.tsmin = 0x7fffffff;
  for (int j = 0; j < n; j++)
    { pd.job[j] = AC.job[j];  pd.orig[j] = AC.orig[j];  pd.fr[j] = AC.fr[j];  pd.slot[j] = AC.slot[j];
      pd.ablk[j] = AC.ablk[j];  pd.bblk[j] = AC.bblk[j];
      pd.amax = std::max(pd.amax, AC.job[j].ablock->maxlen);  pd.bmax = std::max(pd.bmax, AC.job[j].bblock->maxlen);
      pd.tsmin = std::min(pd.tsmin, Trace_Spacing(AC.job[j].spec));
    }
  const u32 rec_have = RS.rec_cap_set[pd.oset], tp_have = RS.tpool_cap_set[pd.oset];     /* (of ITS set: asking one set for the
                                                                                             other's size plus headroom would grow both for ever) */
  pd.cell_cap = std::min<u32>(RS.cell_cap ? RS.cell_cap : DEFAULT_CELLS, max_cells());
  pd.rec_cap  = (u32) std::min<u64>(0x7fffffffu, std::max<u64>(rec_have, 2 * AC.nwork + 4096));
  pd.tp_cap   = (u32) std::min<u64>(0xe0000000ull, std::max<u64>(tp_have, (u64) pd.rec_cap * 256u));
  { /* trace values as bytes off the device (align.c:3375-3396 Compress_TraceTo8, K8 of SURVEY 8a19) when every comparison
       of the launch writes byte traces (-s <= 125); DAMAR_DEVICE_T8=0: 16-bit values, compressed by the host tail as
       until round 4 */
    static int want = -1;
    if (want < 0)
      want = getenv("DAMAR_DEVICE_T8") ? atoi(getenv("DAMAR_DEVICE_T8")) : 1;
    pd.t8 = want;
    for (int j = 0; j < n; j++)
      if (Trace_Spacing(AC.job[j].spec) > TRACE_XOVR)
        pd.t8 = 0;
  }
  if (getenv("DAMAR_TEST_SMALL_CAPS") && RS.rec_cap_set[0] == 0 && RS.rec_cap_set[1] == 0)        /* tests: start far too small, so that the
                                                                  overflow flags and the re-launch are exercised */
    { pd.cell_cap = 64;  pd.rec_cap = 16;  pd.tp_cap = 512; }
  pd.attempt = 0;
  late_streams();
  pd.st = defer ? G_rep : G_st;
  for (TailJob *w : AC.writes)
    pd.writes.push_back(w);
  AC.writes.clear();
  if (defer)                                   /* the launch reads what the seed stages wrote on the other stream */
    { HIP_CHECK(hipEventRecord(G_front_done, G_st));
      HIP_CHECK(hipStreamWaitEvent(G_rep, G_front_done, 0));
    }
  PQ_n += 1;
  report_launch(pd);
  if (!defer)
    finish_oldest();
  else if (!corun_on())                        /* in order on the device: the next seed stages start behind this launch */
    { HIP_CHECK(hipEventRecord(G_rep_done, G_rep));
      HIP_CHECK(hipStreamWaitEvent(G_st, G_rep_done, 0));
    }
  AC.n = 0;  AC.bytes = 0;  AC.nwork = 0;
  Q_flush = now_ms();
}

/* everything this library still owes: the comparisons held back, then the launch in flight */
static void finish_all(void)
{ flush_accum();
  while (PQ_n > 0)
    finish_oldest();
}

extern "C" void damar_match_batch(damar_match_job *jobs, int njobs)
{ ensure_init();
  const double h0 = now_ms();
  if (Q_exit > 0)
    Q_ms[2] += h0 - Q_exit;
  memset(G_cnt, 0, sizeof(G_cnt));
  for (int i = DAMAR_T_MERGE; i < DAMAR_T_COUNT; i++)
    G_ms[i] = 0;
  const bool defer = overlap_on();
  if (!defer)
    finish_all();
  const size_t budget = G_prop.totalGlobalMem / 8;                        /* per launch; up to three groups of comparisons hold arenas */
  const int    hard = DAMAR_MAX_JOBS / 2;                                 /* two sets of job arenas */
  const int    soft = std::min(batch_limit(), hard);
  for (int i = 0; i < njobs; i++)
    { const double f0 = now_ms();
      /* one launch, one set of filter parameters and one scoring (the kernel's trim table): comparisons held back under
         other options go first */
      if (AC.n > 0)
        { const JobParams now = params_now();
          const JobParams &was = AC.fr[0].jp;
          const int16 *s0 = damar_spec_score_table(AC.job[0].spec), *s1 = damar_spec_score_table(jobs[i].spec);
          bool go = now.kmer != was.kmer || now.hitmin != was.hitmin || now.binshift != was.binshift ||
                    s0[32767] != s1[32767] || s0[0] != s1[0];
          /* an Align_Spec whose files are already asked for (damar_write_overlaps) belongs to a finished block pair: a
             new pair on the same spec must not reach its overlap buffers before that write has detached them */
          for (TailJob *w : AC.writes)
            go = go || w->spec == jobs[i].spec;
          if (go)
            flush_accum();
        }
      const int slot = slot_take();               /* (may complete a launch, or launch what is held back) */
      const int n = AC.n;
      if (i == 0)
        Q_seg[2] += now_ms() - h0;
      if (!match_front(&jobs[i], slot, &AC.fr[n]))
        G_slot_busy[slot] = false;
      else
        { AC.job[n] = jobs[i];  AC.orig[n] = &jobs[i];  AC.slot[n] = slot;
          AC.ablk[n] = jobs[i].aidx->blk;  AC.bblk[n] = jobs[i].bidx->blk;
          AC.bytes += AC.fr[n].bytes;
          AC.nwork += AC.fr[n].nwork;
          AC.n = n + 1;
        }
      H_ms[1] += now_ms() - f0;
      if (AC.n >= hard || AC.bytes > budget || (AC.n >= soft && (!defer || AC.nwork >= batch_work())))
        flush_accum();
    }
  if (!defer)
    flush_accum();
  for (int j = 0; j < DAMAR_MAX_JOBS; j++)         /* the caller's job structs end with this call */
    PQ[0].orig[j] = PQ[1].orig[j] = NULL;
  for (int j = 0; j < DAMAR_MAX_JOBS / 2; j++)
    AC.orig[j] = NULL;
  Q_exit = now_ms();
  if (Q_flush > 0)
    Q_seg[0] += Q_exit - Q_flush;
  H_ms[6] += Q_exit - h0;
}

extern "C" void damar_match(const HITS_DB *ablock, const HITS_DB *bblock,
                            damar_dev_index *aidx, damar_dev_index *bidx,
                            int self, int comp, Align_Spec *spec, int64 *counts)
{ damar_match_job job;
  memset(&job, 0, sizeof(job));
  job.ablock = ablock;  job.bblock = bblock;  job.aidx = aidx;  job.bidx = bidx;
  job.self = self;  job.comp = comp;  job.spec = spec;
  G_tail_reports = VERBOSE != 0;                 /* -v: the confirmed hits are counted by the host tail, which may run behind this call */
  damar_match_batch(&job, 1);
  if (G_tail_reports)
    damar_async_drain();
  G_tail_reports = false;
  if (counts)
    { counts[0] = job.counts[0];  counts[1] = job.counts[1];  counts[2] = job.counts[2]; }
  if (VERBOSE)
    { printf("\n     %lld %d-mers\n     %lld seed hits\n     %lld confirmed hits\n",
             (long long) job.counts[0], P_kmer, (long long) job.counts[1], (long long) job.counts[2]);
      fflush(stdout);
    }
}

extern "C" void Match_Filter(char *aname, HITS_DB *ablock, char *bname, HITS_DB *bblock,
                             void *atable, int alen, void *btable, int blen, int comp, Align_Spec *asettings)
{ (void) alen;  (void) blen;
  if (VERBOSE)
    { if (comp) printf("\nComparing %s to c(%s)\n", aname, bname);
      else      printf("\nComparing %s to %s\n", aname, bname);
    }
  damar_match(ablock, bblock, (damar_dev_index *) atable, (damar_dev_index *) btable,
              aname == bname, comp, asettings, NULL);             /* filter.c:2603 pointer equality */
  if (atable != btable && btable != NULL)                         /* filter.c:2722-2731, 2880-2881 */
    damar_index_free((damar_dev_index *) btable);
}


/***** datander: Match_Self (scrub/tandem.c:1182-1428) ************************************************/

static int T_kmer = 12, T_binshift = 4, T_hitmin = 35, T_nshift = 2;

/* scrub/tandem.h:58 Set_Filter_Params(kmer, binshift, hitmin, nthreads): the 4-argument
 * variant of datander.  It cannot share a name with filter.h's 5-argument function inside one
 * library, so it is exported under this name; libdamar_tandem.so (csrc/tandem_abi.c) carries
 * the reference name for a datander.c that links against it. */
extern "C" int damar_tandem_set_params(int kmer, int binshift, int hitmin, int nthreads)
{ if (kmer <= 1)
    return 1;
  T_kmer = kmer;  T_binshift = binshift;  T_hitmin = hitmin;
  T_nshift = 0;
  while ((2 << T_nshift) <= nthreads)
    T_nshift += 1;
  return 0;
}

/* The wide kernel behind a two-pair launch of ONE job on G_st (datander: dist, the batch entry: tasks), for the reads /
   tasks that launch left to it (DAMAR_CNT_WIDE: reads of more than DAMAR_MAX_MARKS trace spacings, alignments that
   overflowed the pebble pool at its largest).  hc = the counters as read behind the launch, updated; a flag left in hc[3]
   means that the whole launch is to be repeated.  Returns whether the wide kernel ran (then the job's map says whose
   two-pair records are to be dropped). */
static bool wide_behind(ReportArgs &ra, u32 *hc, u32 cell_cap, const int *dist, const LaTask *tasks, u32 ntasks)
{ if (ra.widemap == NULL || hc[DAMAR_CNT_WIDE] == 0)
    return false;
  u32 flags = hc[3];
  if (cell_cap >= max_cells())
    flags &= ~DAMAR_ERR_CELLS;                     /* (those are in the map: nothing to repeat for them) */
  if (flags != 0)
    return false;                                  /* something else ran over: the caller grows it and repeats the launch */
  if (RS.wcells == NULL)
    { RS.wslots = std::min(RS.nslots, 64);
      RS.wcell_cap = 1u << 21;
      RS.wcells = dmalloc((size_t) 16 * RS.wcell_cap * (size_t) RS.wslots);
    }
  ra.wcells = RS.wcells;  ra.wcell_cap = RS.wcell_cap;
  if (VERBOSE || getenv("DAMAR_TEST_MAX_CELLS") != NULL)
    fprintf(stderr, "damar: %u read(s) / task(s) beyond the packed pebble format: wide kernel\n", hc[DAMAR_CNT_WIDE]);
  HIP_CHECK(hipMemsetAsync(RS.ctr + DAMAR_CNT_CURSOR, 0, sizeof(u32) * DAMAR_MAX_JOBS, G_st));
  HIP_CHECK(hipMemsetAsync(RS.ctr + 3, 0, sizeof(u32), G_st));
  if (dist != NULL)
    damar_launch_tandem_report_wide(&ra, dist, RS.wslots, G_st);
  else
    damar_launch_la_batch_wide(&ra, tasks, ntasks, RS.wslots, G_st);
  HIP_CHECK(hipMemcpyAsync(hc, RS.ctr, sizeof(u32) * DAMAR_COUNTER_WORDS, hipMemcpyDeviceToHost, G_st));
  HIP_CHECK(hipStreamSynchronize(G_st));
  HIP_CHECK(hipGetLastError());
  if (hc[3] & DAMAR_ERR_CELLS)                     /* the wide kernel's own pool: four times the pebbles, and everything once more */
    { if (RS.wcell_cap >= (1u << 26))
        { fprintf(stderr, "damar: FATAL: an alignment needs more than %u trace pebbles even in the wide format\n", RS.wcell_cap);
          die();
        }
      HIP_CHECK(hipFree(RS.wcells));
      RS.wcell_cap *= 4;
      RS.wslots = std::max(8, RS.wslots / 2);
      RS.wcells = dmalloc((size_t) 16 * RS.wcell_cap * (size_t) RS.wslots);
      hc[3] = (hc[3] & ~DAMAR_ERR_CELLS) | 0x40000000u;
    }
  return true;
}

/* the records of a launch behind which the wide kernel ran: what the two-pair kernel wrote for a read / task before it
   gave it up goes, the wide kernel's records lose their mark */
static void wide_filter(std::vector<LaRecord> &recs, const u32 *dmap, u32 words)
{ std::vector<u32> wmap(words);
  HIP_CHECK(hipMemcpy(wmap.data(), dmap, sizeof(u32) * (size_t) words, hipMemcpyDeviceToHost));
  size_t n = 0;
  for (size_t q = 0; q < recs.size(); q++)
    { u32 it = recs[q].item;
      if (it & DAMAR_ITEM_WIDE)
        recs[q].item = it & ~DAMAR_ITEM_WIDE;
      else if ((wmap[it >> 5] >> (it & 31)) & 1u)
        continue;
      recs[n++] = recs[q];
    }
  recs.resize(n);
}

extern "C" void damar_match_self(const HITS_DB *ablock, damar_dev_block *blk, Align_Spec *spec, int64 *counts)
{ finish_all();
  ensure_init();
  int64 nfilt = 0, ncheck = 0;
  int   n = 0;
  if (counts)
    counts[0] = counts[1] = counts[2] = 0;
  memset(G_cnt, 0, sizeof(G_cnt));
  damar_dev_index *ix = index_build_k(blk, 0, &n, T_kmer, 0, 0);
  if (ix == NULL)
    return;
  const int ts = Trace_Spacing(spec);
  int *dist = (int *) dmalloc(sizeof(int) * (size_t) n);
  tick(0);
  damar_launch_tandem_links(&blk->d, T_kmer, ix->codes, ix->wide, ix->pos, (u32) n, dist, G_st);
  tick(1);

  std::vector<LaRecord> recs;
  std::vector<u16>      tpool;
  u32 hc[DAMAR_COUNTER_WORDS];
  u32 *wmap = NULL;
  const u32 wwords = ((u32) ablock->nreads + 31) / 32 + 1;
  bool wide_ran = false;
  { const int sk = P_kmer, sh = P_hitmin, sb = P_binshift;        /* the report args read the P_* set */
    P_kmer = T_kmer;  P_hitmin = T_hitmin;  P_binshift = T_binshift;
    u32 cell_cap = std::min<u32>(RS.cell_cap ? RS.cell_cap : DEFAULT_CELLS, max_cells());
    u32 rec_cap  = std::max(RS.rec_cap, (u32) (4 * ablock->nreads + 4096));
    u32 tp_cap   = std::max(RS.tpool_cap, rec_cap * 64u);
    for (int attempt = 0; ; attempt++)
      { ReportArgs ra;
        scratch_prepare(ablock->maxlen, ablock->maxlen, T_binshift, ts, cell_cap, G_st);
        scratch_outputs(rec_cap, tp_cap);
        fill_report_args(&ra, blk, blk, 0, 1, spec, G_st, 0, 0, params_now());
        ra.nwork = (u32) ablock->nreads;
        const bool packed = use_packed(&ra, ablock->maxlen, ablock->maxlen);
        if (!packed)
          marks_must_fit(ablock->maxlen, ablock->maxlen, ts);      /* (one read per wavefront, DAMAR_PACKED=0: no wide kernel behind it) */
        else
          { /* one bit per read: which are the wide kernel's (kernels/report.hip tandem_wide_kernel) */
            if (wmap == NULL)
              wmap = (u32 *) dmalloc(sizeof(u32) * (size_t) wwords);
            HIP_CHECK(hipMemsetAsync(wmap, 0, sizeof(u32) * (size_t) wwords, G_st));
            ra.widemap = wmap;  ra.wcells = RS.wcells;  ra.wcell_cap = RS.wcell_cap;  ra.cell_max = max_cells();
          }
        HIP_CHECK(hipMemsetAsync(RS.ctr, 0, sizeof(u32) * DAMAR_COUNTER_WORDS, G_st));
        tick(4);
        if (packed)
          damar_launch_tandem_report2(&ra, dist, RS.nslots, G_st);
        else
          damar_launch_tandem_report(&ra, dist, RS.nslots, G_st);
        tick(5);
        HIP_CHECK(hipMemcpyAsync(hc, RS.ctr, sizeof(hc), hipMemcpyDeviceToHost, G_st));
        HIP_CHECK(hipStreamSynchronize(G_st));
        HIP_CHECK(hipGetLastError());
        G_ms[DAMAR_T_REPORT] = lap(4, 5);
        wide_ran = packed && wide_behind(ra, hc, cell_cap, dist, NULL, 0);
        if (packed && !wide_ran && cell_cap >= max_cells() && hc[DAMAR_CNT_WIDE] > 0)
          hc[3] &= ~DAMAR_ERR_CELLS;
        if (hc[3] == 0)
          break;
        hc[3] &= ~0x40000000u;
        if (((hc[3] & DAMAR_ERR_BAND) && !(hc[3] & (DAMAR_ERR_CELLS | DAMAR_ERR_WIDE))) || attempt >= 6)
          { fprintf(stderr, "damar: FATAL: tandem report kernel failed (flags %u, where=%u)\n", hc[3], hc[6]);
            die();
          }
        if (hc[3] & DAMAR_ERR_CELLS) cell_cap = grow_cells(cell_cap);
        if (hc[3] & DAMAR_ERR_WIDE)  G_ring *= 4;
        if (hc[3] & DAMAR_ERR_RECS)  rec_cap = std::max(2 * rec_cap, hc[1] + 1024);
        if (hc[3] & DAMAR_ERR_TPOOL) tp_cap  = std::max(2 * tp_cap, hc[2] + 65536);
      }
    P_kmer = sk;  P_hitmin = sh;  P_binshift = sb;
  }
  recs.resize(hc[1]);
  tpool.resize(hc[2]);
  if (hc[1] > 0)
    { HIP_CHECK(hipMemcpy(recs.data(), RS.recs, sizeof(LaRecord) * (size_t) hc[1], hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(tpool.data(), RS.tpool, sizeof(u16) * (size_t) hc[2], hipMemcpyDeviceToHost));
    }
  nfilt = hc[DAMAR_CNT_NFILT];
  if (wide_ran)
    wide_filter(recs, wmap, wwords);
  if (wmap != NULL)
    HIP_CHECK(hipFree(wmap));
  HIP_CHECK(hipFree(dist));
  damar_index_free(ix);

  /* host tail: scrub/tandem.c:1140-1166 (fusion/containment only, A records only) */
  std::sort(recs.begin(), recs.end(), RecOrder());
  Overlap_IO_Buffer *obuf = OVL_IO_Buffer(spec);
  std::vector<damar_path> am;
  damar_tpool tp = { NULL, 0, 0 };
  size_t i = 0;
  while (i < recs.size())
    { size_t j = i;
      while (j < recs.size() && recs[j].item == recs[i].item)
        j += 1;
      am.clear();  tp.top = 0;
      for (size_t q = i; q < j; q++)
        { const LaRecord &r = recs[q];
          damar_path p;
          p.tlen = r.atlen;  p.diffs = r.diffs;
          p.abpos = r.abpos;  p.bbpos = r.bbpos;  p.aepos = r.aepos;  p.bepos = r.bepos;
          p.toff = damar_tpool_push(&tp, tpool.data() + r.toff, r.atlen);
          am.push_back(p);
        }
      const int ar = recs[i].aread + ablock->ufirst;
      damar_emit_pair(am.data(), (int) am.size(), NULL, 0, &tp, 0, ts, ar, ar, NULL, obuf, &ncheck);
      i = j;
    }
  free(tp.val);
  G_cnt[0] = n;  G_cnt[2] = nfilt;  G_cnt[3] = hc[1];
  if (counts)
    { counts[0] = n;  counts[1] = nfilt;  counts[2] = ncheck; }
  if (VERBOSE)
    { printf("\n     %lld seed hits\n     %lld confirmed hits\n", (long long) nfilt, (long long) ncheck);
      fflush(stdout);
    }
}

extern "C" void Match_Self(char *aname, HITS_DB *ablock, Align_Spec *settings)
{ if (VERBOSE)
    printf("\nIndexing %s\n\nComparing %s to itself\n", aname, aname);
  damar_dev_block *b = damar_block_upload(ablock);
  damar_match_self(ablock, b, settings, NULL);
  damar_block_free(b);
}

/***** test hooks ***************************************************************************************/

extern "C" int64 damar_last_seeds(void *out, int64 cap)
{ struct SP { int diag, apos, aread, bread; } *sp = (SP *) out;
  if (out == NULL)
    { G_keep_seeds = (cap != 0);
      return 0;
    }
  int64 n = (int64) G_seed_keys.size();
  for (int64 i = 0; i < n && i < cap; i++)
    { u64 k = G_seed_keys[(size_t) i];
      const int bpos = (int) (k & ((1ull << G_seed_dbits) - 1));
      k >>= G_seed_dbits;
      sp[i].apos  = (int) (k & ((1ull << G_seed_pbits) - 1));
      sp[i].aread = (int) ((k >> G_seed_pbits) & ((1ull << G_seed_abits) - 1));
      sp[i].bread = (int) (k >> (G_seed_pbits + G_seed_abits));
      sp[i].diag  = G_seed_dbits ? sp[i].apos - bpos : (int) G_seed_vals[(size_t) i];
    }
  return n;
}

extern "C" int damar_local_alignment_batch(damar_dev_block *ablk, damar_dev_block *bblk, int comp,
                                           Align_Spec *spec, const int *tasks, int ntasks,
                                           int *paths, int64 *trace_off, uint16 *traces, int64 trace_cap)
{ finish_all();
  ensure_init();
  if (ntasks <= 0)
    return 0;
  const int ts = Trace_Spacing(spec);
  u32 cell_cap = std::min<u32>(RS.cell_cap ? RS.cell_cap : DEFAULT_CELLS, max_cells()), rec_cap = 2 * (u32) ntasks + 16,
      tp_cap = (u32) std::min<int64>(trace_cap + 1024, 0x7fffffff);
  LaTask *dt = (LaTask *) dmalloc(sizeof(LaTask) * (size_t) ntasks);
  HIP_CHECK(hipMemcpy(dt, tasks, sizeof(LaTask) * (size_t) ntasks, hipMemcpyHostToDevice));
  u32 hc[DAMAR_COUNTER_WORDS];
  u32 *wmap = NULL;
  const u32 wwords = ((u32) ntasks + 31) / 32 + 1;
  bool wide_ran = false;
  for (int attempt = 0; ; attempt++)
    { ReportArgs ra;
      scratch_prepare(ablk->d.maxlen, bblk->d.maxlen, P_binshift, ts, cell_cap, G_st);
      stage("la_scratch");
      scratch_outputs(rec_cap, tp_cap);
      fill_report_args(&ra, ablk, bblk, comp, 0, spec, G_st, 0, 0, params_now());
      const bool packed = use_packed(&ra, ablk->d.maxlen, bblk->d.maxlen);
      if (!packed)
        marks_must_fit(ablk->d.maxlen, bblk->d.maxlen, ts);
      else
        { if (wmap == NULL)
            wmap = (u32 *) dmalloc(sizeof(u32) * (size_t) wwords);
          HIP_CHECK(hipMemsetAsync(wmap, 0, sizeof(u32) * (size_t) wwords, G_st));
          ra.widemap = wmap;  ra.wcells = RS.wcells;  ra.wcell_cap = RS.wcell_cap;  ra.cell_max = max_cells();
        }
      HIP_CHECK(hipMemsetAsync(RS.ctr, 0, sizeof(u32) * DAMAR_COUNTER_WORDS, G_st));
      stage("la_setup");
      if (packed)
        damar_launch_report2(&ra, 1, dt, (u32) ntasks, RS.nslots, G_st);
      else
        damar_launch_la_batch(&ra, dt, (u32) ntasks, RS.nslots, G_st);
      stage("la_kernel");
      HIP_CHECK(hipMemcpyAsync(hc, RS.ctr, sizeof(hc), hipMemcpyDeviceToHost, G_st));
      HIP_CHECK(hipStreamSynchronize(G_st));
      HIP_CHECK(hipGetLastError());
      wide_ran = packed && wide_behind(ra, hc, cell_cap, NULL, dt, (u32) ntasks);
      if (packed && !wide_ran && cell_cap >= max_cells() && hc[DAMAR_CNT_WIDE] > 0)
        hc[3] &= ~DAMAR_ERR_CELLS;
      if (hc[3] == 0)
        break;
      hc[3] &= ~0x40000000u;
      if (((hc[3] & DAMAR_ERR_BAND) && !(hc[3] & (DAMAR_ERR_CELLS | DAMAR_ERR_WIDE))) || attempt >= 6)
        { fprintf(stderr, "damar: FATAL: batch Local_Alignment failed (flags %u, where=%u)\n", hc[3], hc[6]);
          die();
        }
      if (hc[3] & DAMAR_ERR_CELLS) cell_cap = grow_cells(cell_cap);
      if (hc[3] & DAMAR_ERR_WIDE)  G_ring *= 4;
      if (hc[3] & DAMAR_ERR_RECS)
        rec_cap = std::max(2 * rec_cap, hc[1] + 1024);
      if (hc[3] & DAMAR_ERR_TPOOL)
        { HIP_CHECK(hipFree(dt));
          if (wmap != NULL)
            HIP_CHECK(hipFree(wmap));
          return -1;
        }
    }
  HIP_CHECK(hipFree(dt));
  std::vector<LaRecord> recs(hc[1]);
  std::vector<u16> tp(hc[2]);
  HIP_CHECK(hipMemcpy(recs.data(), RS.recs, sizeof(LaRecord) * (size_t) hc[1], hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(tp.data(), RS.tpool, sizeof(u16) * (size_t) hc[2], hipMemcpyDeviceToHost));
  if (wide_ran)
    wide_filter(recs, wmap, wwords);
  if (wmap != NULL)
    HIP_CHECK(hipFree(wmap));
  std::sort(recs.begin(), recs.end(), RecOrder());
  std::vector<u32> boffa((size_t) ablk->nreads + 1), boffb((size_t) bblk->nreads + 1);
  HIP_CHECK(hipMemcpy(boffa.data(), ablk->boff, sizeof(u32) * boffa.size(), hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(boffb.data(), bblk->boff, sizeof(u32) * boffb.size(), hipMemcpyDeviceToHost));
  int64 top = 0;
  for (size_t i = 0; i < recs.size(); i++)
    { const LaRecord &r = recs[i];
      int *p = paths + 12 * r.item;
      int  al = (int) (boffa[r.aread + 1] - boffa[r.aread] - 1), bl = (int) (boffb[r.bread + 1] - boffb[r.bread] - 1);
      p[0] = r.abpos;  p[1] = r.bbpos;  p[2] = r.aepos;  p[3] = r.bepos;  p[4] = r.diffs;  p[5] = r.atlen;
      if (comp)
        { p[6] = bl - r.bepos;  p[7] = al - r.aepos;  p[8] = bl - r.bbpos;  p[9] = al - r.abpos; }
      else
        { p[6] = r.bbpos;  p[7] = r.abpos;  p[8] = r.bepos;  p[9] = r.aepos; }
      p[10] = r.diffs;  p[11] = r.btlen;
      if (top + r.atlen + r.btlen > trace_cap)
        return -1;
      trace_off[2 * r.item] = top;
      memcpy(traces + top, tp.data() + r.toff, sizeof(u16) * (size_t) r.atlen);
      top += r.atlen;
      trace_off[2 * r.item + 1] = top;
      memcpy(traces + top, tp.data() + r.toff + r.atlen, sizeof(u16) * (size_t) r.btlen);
      top += r.btlen;
    }
  return 0;
}

extern "C" double damar_bench_sort_u32(uint32_t n, int nbits, int reps, uint32_t seed)
{ ensure_init();
  std::vector<u32> h(n);
  u32 x = seed ? seed : 1u, mask = (nbits >= 32) ? 0xffffffffu : ((1u << nbits) - 1);
  for (u32 i = 0; i < n; i++)
    { x ^= x << 13;  x ^= x >> 17;  x ^= x << 5;
      h[i] = x & mask;
    }
  u32 *src = (u32 *) dmalloc(sizeof(u32) * (size_t) n);
  u32 *k0 = (u32 *) dmalloc(sizeof(u32) * (size_t) n), *v0 = (u32 *) dmalloc(sizeof(u32) * (size_t) n);
  u32 *k1 = (u32 *) dmalloc(sizeof(u32) * (size_t) n), *v1 = (u32 *) dmalloc(sizeof(u32) * (size_t) n);
  void *sw = dmalloc(damar_sort_workspace_bytes(n));
  HIP_CHECK(hipMemcpy(src, h.data(), sizeof(u32) * (size_t) n, hipMemcpyHostToDevice));
  double total = 0;
  for (int r = 0; r < reps + 1; r++)
    { HIP_CHECK(hipMemcpyAsync(k0, src, sizeof(u32) * (size_t) n, hipMemcpyDeviceToDevice, G_st));
      HIP_CHECK(hipMemcpyAsync(v0, src, sizeof(u32) * (size_t) n, hipMemcpyDeviceToDevice, G_st));
      tick(8);
      damar_radix_sort_u32(k0, v0, k1, v1, n, nbits, sw, G_st);
      tick(9);
      HIP_CHECK(hipStreamSynchronize(G_st));
      if (r > 0)
        total += lap(8, 9);
    }
  HIP_CHECK(hipFree(src));  HIP_CHECK(hipFree(k0));  HIP_CHECK(hipFree(v0));
  HIP_CHECK(hipFree(k1));   HIP_CHECK(hipFree(v1));  HIP_CHECK(hipFree(sw));
  return total / reps;
}

/***** align.h:223-259: Work_Data and the single-call Local_Alignment ***********************************/

struct WorkData
{ Path   bpath;
  std::vector<uint16> atrace, btrace;
  std::vector<int>    script;          /* Compute_Trace_PTS */
};

extern "C" Work_Data *New_Work_Data(void)             /* align.c:135-160 */
{ return (Work_Data *) new WorkData();
}

extern "C" void Free_Work_Data(Work_Data *work)       /* align.c:162-173 */
{ delete (WorkData *) work;
}

/* align.c:1904-2097 for one pair of sequences, computed by the same wave kernel as Match_Filter
 * (one task through the batch path: the two sequences travel to HBM and back per call, so this
 * entry point is for callers that need the reference's API, not for throughput).  Supported call
 * shape: the one filter.c:2316 uses, low == hgh (seed diagonal) and no borders (lbord, hbord < 0);
 * anything else exits as the reference's batch-mode errors do.  align->aseq/bseq follow the
 * block layout (a 4 before the first and after the last base). */
extern "C" Path *Local_Alignment(Alignment *align, Work_Data *work, Align_Spec *spec,
                                 int low, int hgh, int anti, int lbord, int hbord)
{ ensure_init();
  WorkData *w = (WorkData *) work;
  if (low != hgh || lbord >= 0 || hbord >= 0)
    { fprintf(stderr, "damar: Local_Alignment: only the seed-diagonal call shape (low == hgh, no borders) is built\n");
      exit(1);
    }
  HITS_DB   db[2];
  HITS_READ rd[2][2];
  std::vector<char> buf[2];
  const char *seq[2] = { align->aseq, align->bseq };
  const int   len[2] = { align->alen, align->blen };
  const int   same = (align->aseq == align->bseq);
  for (int i = 0; i < 2; i++)
    { memset(&db[i], 0, sizeof(HITS_DB));
      memset(rd[i], 0, sizeof(rd[i]));
      buf[i].assign((size_t) len[i] + 2, 4);
      memcpy(buf[i].data() + 1, seq[i], (size_t) len[i]);
      rd[i][0].rlen = len[i];  rd[i][0].boff = 0;  rd[i][1].boff = len[i] + 1;
      db[i].nreads = db[i].ureads = 1;  db[i].maxlen = len[i];  db[i].totlen = len[i];
      db[i].bases = buf[i].data() + 1;  db[i].reads = rd[i];
    }
  damar_dev_block *ab = damar_block_upload(&db[0]);
  damar_dev_block *bb = same ? ab : damar_block_upload(&db[1]);
  const int mtp = 2 * (std::max(len[0], len[1]) / Trace_Spacing(spec) + 2) + 8;
  std::vector<uint16> tr((size_t) 4 * mtp + 64);
  int   task[4] = { 0, 0, low, anti }, paths[12];
  int64 toff[2];
  if (damar_local_alignment_batch(ab, bb, (int) (align->flags & COMP_FLAG), spec, task, 1, paths, toff,
                                  tr.data(), (int64) tr.size()))
    { fprintf(stderr, "damar: Local_Alignment: trace buffer too small\n");
      exit(1);
    }
  damar_block_free(ab);
  if (!same)
    damar_block_free(bb);
  Path *ap = align->path;
  ap->abpos = paths[0];  ap->bbpos = paths[1];  ap->aepos = paths[2];  ap->bepos = paths[3];
  ap->diffs = paths[4];  ap->tlen = paths[5];
  w->atrace.assign(tr.begin() + toff[0], tr.begin() + toff[0] + paths[5]);
  ap->trace = w->atrace.data();
  w->bpath.abpos = paths[6];  w->bpath.bbpos = paths[7];  w->bpath.aepos = paths[8];  w->bpath.bepos = paths[9];
  w->bpath.diffs = paths[10];  w->bpath.tlen = paths[11];
  w->btrace.assign(tr.begin() + toff[1], tr.begin() + toff[1] + paths[11]);
  w->bpath.trace = w->btrace.data();
  return &w->bpath;
}

/***** (f)4: trace-point expansion, align.c:5577-5692 Compute_Trace_PTS for batches of records *********/

struct DBuf
{ void  *p = nullptr;
  size_t cap = 0;
  void *need(size_t n)
  { if (n > cap)
      { if (p) HIP_CHECK(hipFree(p));
        cap = n + n / 4 + 4096;
        HIP_CHECK(hipMalloc(&p, cap));
      }
    return p;
  }
  void drop() { if (p) HIP_CHECK(hipFree(p));  p = nullptr;  cap = 0; }
};

static DBuf T_recs, T_pts, T_segs, T_count, T_dist, T_segoff, T_stage, T_vf, T_hf, T_over, T_ctr, T_tlen,
            T_diffs, T_script, T_scan, T_bvf, T_bhf, T_mid, T_segs2, T_key, T_val, T_key1, T_val1, T_sortw;
static double T_ms[4];        /* of the last damar_trace_pts: trace_waves kernel, layout..pack on the device, whole call, inside the batches (wall) */
static int64  T_cnt[4];       /* records, segments, deferred segments, script values */

extern "C" void damar_trace_release(void)
{ DBuf *all[] = { &T_recs, &T_pts, &T_segs, &T_count, &T_dist, &T_segoff, &T_stage, &T_vf, &T_hf, &T_over, &T_ctr,
                  &T_tlen, &T_diffs, &T_script, &T_scan, &T_bvf, &T_bhf, &T_mid, &T_segs2, &T_key, &T_val, &T_key1, &T_val1, &T_sortw };
  for (DBuf *b : all) b->drop();
}

extern "C" void damar_trace_last(double *ms, int64 *cnt)
{ for (int i = 0; i < 4; i++) { ms[i] = T_ms[i];  cnt[i] = T_cnt[i]; }
}

/* staging slots one record needs = sum over its segments of dmax + |M - N| (the script of a segment has at
   most D + |del| <= dmax + |del| values); also dmax and the segment count */
template <typename PT>
static void trace_record_shape(const Path *path, const PT *p, int tspace, int *dmax_out, int *nseg_out, int64 *slots_out)
{ const int tlen = path->tlen;
  int dmax = 0;
  for (int d = 0; d + 1 < tlen; d += 2)
    if ((int) p[d] > dmax) dmax = (int) p[d];
  const int nseg = tlen >= 2 ? tlen / 2 : 1;
  int ab = path->abpos, ae = (ab / tspace) * tspace, bb = path->bbpos;
  int64 slots = 0;
  for (int s = 0; s < nseg; s++)
    { int be;
      if (s + 1 < nseg) { ae += tspace; be = bb + (int) p[2 * s + 1]; }
      else              { ae = path->aepos; be = path->bepos; }
      const int del = (ae - ab) - (be - bb);
      slots += dmax + (del < 0 ? -del : del);
      ab = ae;
      bb = be;
    }
  *dmax_out = dmax;  *nseg_out = nseg;  *slots_out = slots;
}

/* One wave phase over `nwork` segments: the slot kernel, then the stripe kernel for what it deferred.
   kind 0 = scripts, 1 = mid points.  Returns the error flags, or ~0u after a message. */
static u32 trace_wave_phase(TraceArgs t, int mode, int kind, u32 nwork, u32 rows, u32 maxblocks, u32 *d_ctr,
                            u32 *d_key, u32 *d_val, hipEvent_t e1, hipEvent_t e2)
{ const u32 nblocks = std::min(maxblocks, (nwork + 63) / 64);
  /* work order: segments with equal difference counts side by side in a wavefront */
  static int ordered = -1;
  if (ordered < 0)
    { const char *e = getenv("DAMAR_TRACE_ORDER");
      ordered = e ? atoi(e) : 1;
    }
  const u32 *order = NULL;
  if (ordered)
    { u32 *k1 = (u32 *) T_key1.need(sizeof(u32) * (size_t) nwork), *v1 = (u32 *) T_val1.need(sizeof(u32) * (size_t) nwork);
      void *sw = T_sortw.need(damar_sort_workspace_bytes(nwork));
      order = damar_radix_sort_u32(d_key, d_val, k1, v1, nwork, 8, sw, G_st) ? v1 : d_val;
      sort_check(sw);
    }
  const size_t area = damar_trace_slot_area_cells();
  t.vf = (short *) T_vf.need((size_t) nblocks * damar_trace_slot_vf_bytes(mode, kind));
  t.hf = (signed char *) T_hf.need((size_t) nblocks * area);
  t.cap = rows;
  t.list = order;  t.nwork = nwork;
  t.next = d_ctr + 4;
  HIP_CHECK(hipMemsetAsync(d_ctr, 0, 8, G_st));
  HIP_CHECK(hipMemsetAsync(d_ctr + 4, 0, 4, G_st));
  HIP_CHECK(hipEventRecord(e1, G_st));
  damar_launch_trace_waves_slots(&t, mode, kind, nblocks, G_st);
  HIP_CHECK(hipEventRecord(e2, G_st));
  u32 ctr[4];
  HIP_CHECK(hipMemcpyAsync(ctr, d_ctr, sizeof(ctr), hipMemcpyDeviceToHost, G_st));
  HIP_CHECK(hipStreamSynchronize(G_st));
  float wms = 0;
  HIP_CHECK(hipEventElapsedTime(&wms, e1, e2));
  T_ms[0] += wms;
  if (ctr[0] > 0 && !(ctr[2] & DAMAR_TRACE_ERR_POINTS))
    { /* segments the slot kernel does not take: one lane each on stripes that hold dmax + 3 rows */
      const u32 need = ctr[1];
      size_t bthreads = ((size_t) ctr[0] + 63) / 64 * 64;
      const size_t budget = (size_t) 8 << 30;
      while (bthreads > 64 && bthreads * need * 3 > budget) bthreads /= 2;
      bthreads = (bthreads + 63) / 64 * 64;
      if (bthreads * need * 3 > ((size_t) 64 << 30))
        { fprintf(stderr, "damar: trace expansion: a segment needs %u wave cells, more than this build provides\n", need);
          return ~0u;
        }
      t.vf = (short *) T_bvf.need(sizeof(short) * bthreads * need);
      t.hf = (signed char *) T_bhf.need(bthreads * need);
      t.cap = need;
      t.list = t.over;  t.nwork = ctr[0];
      HIP_CHECK(hipMemsetAsync(d_ctr, 0, 8, G_st));
      HIP_CHECK(hipEventRecord(e1, G_st));
      damar_launch_trace_waves(&t, mode, kind, (u32) (bthreads / 64), G_st);
      HIP_CHECK(hipEventRecord(e2, G_st));
      u32 c2[4];
      HIP_CHECK(hipMemcpyAsync(c2, d_ctr, sizeof(c2), hipMemcpyDeviceToHost, G_st));
      HIP_CHECK(hipStreamSynchronize(G_st));
      HIP_CHECK(hipEventElapsedTime(&wms, e1, e2));
      T_ms[0] += wms;
      T_cnt[2] += ctr[0];
      if (c2[0] != 0)
        { fprintf(stderr, "damar: trace expansion: internal error, %u segments deferred twice\n", c2[0]);
          return ~0u;
        }
      ctr[2] |= c2[2];
    }
  if (ctr[2] & DAMAR_TRACE_ERR_POINTS)
    { fprintf(stderr, "damar: Trace point out of bounds (Compute_Trace), source DB likely incorrect\n");   /* align.c:5575 */
      return ~0u;
    }
  if (ctr[2] & DAMAR_TRACE_ERR_ALIGN)
    { fprintf(stderr, "damar: Bad alignment between trace points (Compute_Trace), source DB likely incorrect\n");   /* :4890 */
      return ~0u;
    }
  if (ctr[2] & DAMAR_TRACE_ERR_INTERNAL)
    { fprintf(stderr, "damar: trace expansion: internal error, staging bound of the mid-point pieces violated\n");
      return ~0u;
    }
  return ctr[2];
}

/* kind 0: Compute_Trace_PTS, the script of every trace-point segment.  kind 1: Compute_Trace_MID: a first
   wave phase finds the mid point of every segment, the pieces between successive mid points (one more
   than segments per record) are laid out anew and the script phase runs on those. */
static int trace_batch(const DevBlock *ad, const DevBlock *bd, int64 r0, int64 r1, int tbytes, int tspace, int mode, int kind,
                       const std::vector<TraceRecIn> &recs, const std::vector<u8> &pts,
                       u32 nsegs, u64 nslots, int64 *soff, int *diffs, int **script, int64 *nscript)
{ const u32 nrecs = (u32) (r1 - r0);
  static u32 rows = 0, maxblocks = 0;
  if (rows == 0)
    { const char *e = getenv("DAMAR_TRACE_ROWS");        /* test hook: fewer rows -> more segments deferred */
      rows = e ? (u32) atoi(e) : 64u;
      if (rows < 4) rows = 4;
      e = getenv("DAMAR_TRACE_BLOCKS");
      maxblocks = e ? (u32) atoi(e) : (u32) (G_prop.multiProcessorCount * 16);
      if (maxblocks < 1) maxblocks = 1;
    }
  const double h0 = now_ms();
  const u32 nwork = nsegs + (kind ? nrecs : 0u);             /* segments of the script phase */
  TraceRecIn *d_recs = (TraceRecIn *) T_recs.need(sizeof(TraceRecIn) * (size_t) nrecs);
  void       *d_pts  = T_pts.need(pts.size() + 64);
  TraceSeg   *d_segs = (TraceSeg *) T_segs.need(sizeof(TraceSeg) * (size_t) nsegs);
  u32 *d_count  = (u32 *) T_count.need(sizeof(u32) * (size_t) nwork);
  int *d_dist   = (int *) T_dist.need(sizeof(int) * (size_t) nwork);
  u32 *d_segoff = (u32 *) T_segoff.need(sizeof(u32) * (size_t) nwork);
  int *d_stage  = (int *) T_stage.need(sizeof(int) * (size_t) (nslots + 16));
  u32 *d_over   = (u32 *) T_over.need(sizeof(u32) * (size_t) nwork);
  u32 *d_ctr    = (u32 *) T_ctr.need(256);                    /* [0] deferred, [1] cells needed, [2] error flags; u64 total at +64 */
  u32 *d_tlen   = (u32 *) T_tlen.need(sizeof(u32) * (size_t) (nrecs + 1));
  int *d_diffs  = (int *) T_diffs.need(sizeof(int) * (size_t) nrecs);
  void *d_scan  = T_scan.need(damar_scan_workspace_bytes(nrecs));
  u32 *d_key    = (u32 *) T_key.need(sizeof(u32) * (size_t) nwork);
  u32 *d_val    = (u32 *) T_val.need(sizeof(u32) * (size_t) nwork);

  struct Events            /* destroyed on every return path */
  { hipEvent_t e[4];
    Events()  { for (int i = 0; i < 4; i++) HIP_CHECK(hipEventCreate(&e[i])); }
    ~Events() { for (int i = 0; i < 4; i++) (void) hipEventDestroy(e[i]); }
  } evs;
  hipEvent_t e0 = evs.e[0], e1 = evs.e[1], e2 = evs.e[2], e3 = evs.e[3];
  HIP_CHECK(hipMemcpyAsync(d_recs, recs.data(), sizeof(TraceRecIn) * (size_t) nrecs, hipMemcpyHostToDevice, G_st));
  if (!pts.empty())
    HIP_CHECK(hipMemcpyAsync(d_pts, pts.data(), pts.size(), hipMemcpyHostToDevice, G_st));
  HIP_CHECK(hipMemsetAsync(d_ctr, 0, 256, G_st));
  HIP_CHECK(hipEventRecord(e0, G_st));
  damar_launch_trace_layout(d_recs, nrecs, d_pts, tbytes, tspace, ad, bd, d_segs, d_key, d_val, d_ctr + 2, G_st);
  TraceArgs t;
  memset(&t, 0, sizeof(t));
  t.segs = d_segs;
  t.abases = ad->bases;  t.bbases = bd->bases;
  t.apk = ad->pk;  t.bpk = bd->pk;
  t.stage = d_stage;  t.count = d_count;  t.dist = d_dist;
  t.over = d_over;  t.over_cap = nwork;  t.nover = d_ctr;  t.need = d_ctr + 1;  t.err = d_ctr + 2;
  if (kind)
    { t.mid = (int *) T_mid.need(sizeof(int) * 2 * (size_t) nsegs);
      if (trace_wave_phase(t, mode, 1, nsegs, rows, maxblocks, d_ctr, d_key, d_val, e1, e2) == ~0u)
        return 1;
      TraceSeg *d_segs2 = (TraceSeg *) T_segs2.need(sizeof(TraceSeg) * (size_t) nwork);
      damar_launch_trace_mid_layout(d_recs, nrecs, d_segs, t.mid, ad, bd, d_segs2, d_key, d_val, d_ctr + 2, G_st);
      t.segs = d_segs = d_segs2;
    }
  if (trace_wave_phase(t, mode, 0, nwork, rows, maxblocks, d_ctr, d_key, d_val, e1, e2) == ~0u)
    return 1;
  damar_launch_trace_gather(d_recs, nrecs, kind, d_count, d_dist, d_segoff, d_tlen, d_diffs, G_st);
  u64 *d_tot = (u64 *) ((char *) d_ctr + 64);
  damar_exclusive_scan_u32(d_tlen, d_tlen, nrecs, d_scan, d_tot, G_st);
  u64 total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, d_tot, sizeof(u64), hipMemcpyDeviceToHost, G_st));
  HIP_CHECK(hipStreamSynchronize(G_st));
  if (total >= 0xfffffff0ull)
    { fprintf(stderr, "damar: trace expansion: batch script exceeds 32-bit offsets\n");
      return 1;
    }
  int *d_script = (int *) T_script.need(sizeof(int) * (size_t) (total + 16));
  damar_launch_trace_pack(d_segs, nwork, d_count, d_segoff, d_tlen, d_stage, d_script, G_st);
  HIP_CHECK(hipEventRecord(e3, G_st));
  std::vector<u32> hoff(nrecs);
  const size_t base = (size_t) *nscript;
  { int *grown = (int *) realloc(*script, sizeof(int) * (base + (size_t) total + 1));
    if (grown == NULL)
      { fprintf(stderr, "damar: out of memory (edit scripts)\n");
        return 1;
      }
    *script = grown;
    *nscript = (int64) (base + (size_t) total);
  }
  HIP_CHECK(hipMemcpyAsync(hoff.data(), d_tlen, sizeof(u32) * (size_t) nrecs, hipMemcpyDeviceToHost, G_st));
  HIP_CHECK(hipMemcpyAsync(diffs + r0, d_diffs, sizeof(int) * (size_t) nrecs, hipMemcpyDeviceToHost, G_st));
  if (total > 0)
    HIP_CHECK(hipMemcpyAsync(*script + base, d_script, sizeof(int) * (size_t) total, hipMemcpyDeviceToHost, G_st));
  HIP_CHECK(hipStreamSynchronize(G_st));
  float dms = 0;
  HIP_CHECK(hipEventElapsedTime(&dms, e0, e3));
  T_ms[1] += dms;
  for (u32 i = 0; i < nrecs; i++)
    soff[r0 + i] = (int64) base + hoff[i];
  soff[r1] = (int64) base + (int64) total;
  T_ms[3] += now_ms() - h0;
  T_cnt[0] += nrecs;  T_cnt[1] += nwork;  T_cnt[3] += (int64) total;
  return 0;
}

/* ovls[i].path.trace = the record's trace points as read from the .las (tbytes 1 or 2 per value);
 * aread / bread are DB read ids, afirst / bfirst the ids of the blocks' first reads.  same != 0: A and B of
 * every record are one buffer (align.c:4933-4951; LAshow never is).  On return *script_out is a malloc'ed
 * array holding all edit scripts, record i at [soff[i], soff[i+1]), diffs[i] its summed distance. */
static int trace_expand(damar_dev_block *ablk, int afirst, damar_dev_block *bblk, int bfirst,
                        const Overlap *ovls, int64 novl, int tbytes, int tspace, int mode, int same, int kind,
                        int64 *soff, int *diffs, int **script_out)
{ ensure_init();
  const double t0 = now_ms();
  for (int i = 0; i < 4; i++) { T_ms[i] = 0;  T_cnt[i] = 0; }
  int  *script = NULL;
  int64 nscript = 0;
  *script_out = NULL;
  std::vector<TraceRecIn> recs;
  std::vector<u8> pts;
  u32 max_segs = 1u << 24;                                   /* per batch: segments, staging slots */
  const u64 max_slots = 1ull << 30;
  { const char *e = getenv("DAMAR_TRACE_MAXSEGS");           /* test hook: many small batches */
    if (e && atoi(e) > 0) max_segs = (u32) atoi(e);
  }
  int64 r0 = 0;
  u32   nsegs = 0;
  u64   nslots = 0;
  soff[0] = 0;
  for (int64 i = 0; i <= novl; i++)
    { int dmax = 0, nseg = 0;
      int64 slots = 0;
      if (i < novl)
        { const Overlap *o = ovls + i;
          const int64 ar = (int64) o->aread - afirst, br = (int64) o->bread - bfirst;
          if (ar < 0 || ar >= ablk->nreads || br < 0 || br >= bblk->nreads || o->path.tlen < 0 || (o->path.tlen & 1))
            { fprintf(stderr, "damar: trace expansion: record %lld does not belong to the two blocks\n", (long long) i);
              free(script);
              return 1;
            }
          if (tbytes == 1) trace_record_shape(&o->path, (const uint8 *) o->path.trace, tspace, &dmax, &nseg, &slots);
          else             trace_record_shape(&o->path, (const uint16 *) o->path.trace, tspace, &dmax, &nseg, &slots);
          if (kind)            /* pieces between mid points: |del'| <= |del| of the two segments + 2 (dmax + |del|) drift */
            slots = 3 * slots + 3 * (int64) dmax;
        }
      if (i == novl || nsegs + (u32) nseg > max_segs || nslots + (u64) slots > max_slots)
        { if (i > r0)
            { if (trace_batch(&ablk->d, &bblk->d, r0, i, tbytes, tspace, mode, kind, recs, pts,
                              nsegs, nslots, soff, diffs, &script, &nscript))
                { free(script);
                  return 1;
                }
            }
          recs.clear();  pts.clear();
          r0 = i;  nsegs = 0;  nslots = 0;
          if (i == novl) break;
          if ((u64) slots > max_slots || (u32) nseg > max_segs)
            { fprintf(stderr, "damar: trace expansion: record %lld is larger than a batch\n", (long long) i);
              free(script);
              return 1;
            }
        }
      const Overlap *o = ovls + i;
      TraceRecIn in;
      in.aread = (u32) (o->aread - afirst);  in.bread = (u32) (o->bread - bfirst);
      in.flags = (o->flags & COMP_FLAG ? 1u : 0u) | (same ? 2u : 0u);
      in.abpos = o->path.abpos;  in.bbpos = o->path.bbpos;  in.aepos = o->path.aepos;  in.bepos = o->path.bepos;
      in.poff = (u32) (pts.size() / (size_t) tbytes);
      in.tlen = o->path.tlen;
      in.seg0 = nsegs;
      in.stage0 = (u32) nslots;
      in.dmax = dmax;
      in.slots = (u32) slots;
      recs.push_back(in);
      const u8 *src = (const u8 *) o->path.trace;
      pts.insert(pts.end(), src, src + (size_t) o->path.tlen * (size_t) tbytes);
      nsegs += (u32) nseg;
      nslots += (u64) slots;
    }
  if (script == NULL)
    script = (int *) malloc(sizeof(int));
  *script_out = script;
  T_ms[2] = now_ms() - t0;
  return 0;
}

extern "C" int damar_trace_pts(damar_dev_block *ablk, int afirst, damar_dev_block *bblk, int bfirst,
                               const Overlap *ovls, int64 novl, int tbytes, int tspace, int mode, int same,
                               int64 *soff, int *diffs, int **script_out)
{ return trace_expand(ablk, afirst, bblk, bfirst, ovls, novl, tbytes, tspace, mode, same, 0, soff, diffs, script_out);
}

/* the same arguments, Compute_Trace_MID (align.c:5694-5830): the script between the segments' mid points */
extern "C" int damar_trace_mid(damar_dev_block *ablk, int afirst, damar_dev_block *bblk, int bfirst,
                               const Overlap *ovls, int64 novl, int tbytes, int tspace, int mode, int same,
                               int64 *soff, int *diffs, int **script_out)
{ return trace_expand(ablk, afirst, bblk, bfirst, ovls, novl, tbytes, tspace, mode, same, 1, soff, diffs, script_out);
}

/* align.c:5577-5692 for one record, through the batch path (the two sequences travel to HBM per call: this
 * entry point is for callers that need the reference's API; LAshow-like loops over a .las belong on
 * damar_trace_pts).  As in the reference, align->path->trace holds 16-bit trace-point pairs on entry (after
 * Decompress_TraceTo16), bseq is already complemented where the record says so, and on return path->trace
 * points to the edit script inside `work`, tlen and diffs are updated.  Returns 0, or 1 after the
 * reference's message where the reference's EXIT(1) paths are. */
static int compute_trace(Alignment *align, Work_Data *work, int trace_spacing, int mode, int kind)
{ ensure_init();
  WorkData *w = (WorkData *) work;
  HITS_DB   db[2];
  HITS_READ rd[2][2];
  std::vector<char> buf[2];
  const char *seq[2] = { align->aseq, align->bseq };
  const int   len[2] = { align->alen, align->blen };
  const int   same = (align->aseq == align->bseq);
  for (int i = 0; i < 2; i++)
    { memset(&db[i], 0, sizeof(HITS_DB));
      memset(rd[i], 0, sizeof(rd[i]));
      buf[i].assign((size_t) len[i] + 2, 4);
      memcpy(buf[i].data() + 1, seq[i], (size_t) len[i]);
      rd[i][0].rlen = len[i];  rd[i][0].boff = 0;  rd[i][1].boff = len[i] + 1;
      db[i].nreads = db[i].ureads = 1;  db[i].maxlen = len[i];  db[i].totlen = len[i];
      db[i].bases = buf[i].data() + 1;  db[i].reads = rd[i];
    }
  damar_dev_block *ab = damar_block_upload(&db[0]);
  damar_dev_block *bb = same ? ab : damar_block_upload(&db[1]);
  Overlap o;
  memset(&o, 0, sizeof(o));
  o.path = *align->path;
  int64 soff[2];
  int   diffs = 0, *script = NULL;
  const int rc = trace_expand(ab, 0, bb, 0, &o, 1, 2, trace_spacing, mode, same, kind, soff, &diffs, &script);
  damar_block_free(ab);
  if (!same)
    damar_block_free(bb);
  if (rc)
    return 1;
  w->script.assign(script, script + soff[1]);
  free(script);
  align->path->trace = w->script.data();
  align->path->tlen  = (int) soff[1];
  align->path->diffs = diffs;
  return 0;
}

extern "C" int Compute_Trace_PTS(Alignment *align, Work_Data *work, int trace_spacing, int mode)   /* align.c:5577 */
{ return compute_trace(align, work, trace_spacing, mode, 0);
}

extern "C" int Compute_Trace_MID(Alignment *align, Work_Data *work, int trace_spacing, int mode)   /* align.c:5694 */
{ return compute_trace(align, work, trace_spacing, mode, 1);
}

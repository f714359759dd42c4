/* kernels.h -- host-callable launchers of the gfx950 kernels (internal to the shim). */
#ifndef DAMAR_KERNELS_H
#define DAMAR_KERNELS_H

#include "dev_common.h"

/* sort_scan.hip */
size_t damar_scan_workspace_bytes(u64 n);
void   damar_exclusive_scan_u32(const u32 *in, u32 *out, u64 n, void *work, u64 *total_dev, hipStream_t st);
#define DAMAR_SCAN_TILE 4096      /* items per scan tile (sort_scan.hip) */
void   damar_tile_offsets_u32(const u32 *in, u64 n, void *work, u64 *total_dev, hipStream_t st);
void   damar_scan_tile_counts(u32 *tcount, u32 ntiles, u64 *total_dev, hipStream_t st);
/* radix_sort.hip: stable LSD radix sorts; each returns the side (0: k0/v0, 1: k1/v1) the result is on */
size_t damar_sort_workspace_bytes(u64 n);
void   damar_sort_set_threads(int threads);               /* tile shape: 256 or 512 threads per workgroup */
const u32 *damar_sort_error_word(const void *work);       /* device word, non-zero after a sort whose look-back timed out */
int    damar_radix_sort_keys_u64(u64 *k0, u64 *k1, u64 n, int lobit, int hibit, void *work, hipStream_t st);
void   damar_radix_sort_split_u64(u64 *k0, u64 *k1, u64 n, int lobit, int hibit, u32 *ohi, u32 *olo, void *work,
                                  hipStream_t st);
int    damar_radix_sort_u32(u32 *k0, u32 *v0, u32 *k1, u32 *v1, u64 n, int nbits, void *work, hipStream_t st);
int    damar_radix_sort_keys_u32(u32 *k0, u32 *k1, u64 n, int nbits, void *work, hipStream_t st);    /* keys only */
int    damar_radix_sort_u64(u64 *k0, u32 *v0, u64 *k1, u32 *v1, u64 n, int nbits, void *work, hipStream_t st);

/* A read block resident in HBM. */
typedef struct
{ const u8  *bases;     /* byte per base 0..3, 4 = terminator; bases[-1] == 4 (reference layout) */
  const u32 *pk;        /* the same positions at 2 bits per base (terminators read as 0): word w
                           holds bases 16w .. 16w+15, base 16w in bits 0-1; PK_PAD words either side */
  u32        rbias;     /* the bases in REVERSE order follow in the same array: base total-1-j sits at biased position
                           rbias + j (position of base p in the forward part: p + 16 * PK_PAD), see pack_bases */
  const u32 *boff;      /* [nreads+1] offset of read i in bases                                   */
  const u32 *coarse;    /* [(total >> COARSE_SHIFT) + 2] read containing position q<<COARSE_SHIFT */
  u32        nreads;
  u32        total;     /* boff[nreads]                                                           */
  int        maxlen;
  int        rpbits;    /* > 0: the index of this block keeps a k-mer's position as read << rpbits | offset in the
                           read (both fit 32 bits); 0: as offset in the block (decoded through coarse / boff)      */
  const u32 *moff;      /* mask track (-m): intervals of read i are mdat[2j], mdat[2j+1] for       */
  const int *mdat;      /* j in [moff[i]/2, moff[i+1]/2); NULL when the block carries no mask      */
} DevBlock;

/* Seed-side kernels run beside a resident report launch that is bound by vector instruction issue: four report
   wavefronts and one seed wavefront share a SIMD, and at equal priority the seed wavefront -- which mostly waits for memory --
   gets a fifth of the issue slots whenever it is ready.  Raised priority lets it issue at once and go back to waiting.
   Which kernels do so is a run-time choice per source file (damar_*_set_prio; shim.hip: DAMAR_SEED_PRIO). */
#define SEED_PRIO_VAR(name, dflt) static __device__ int name = dflt;  static int name##_host = dflt;
#define SEED_PRIO(name)     do { if (name) __builtin_amdgcn_s_setprio(3); } while (0)
/* (a copy to a device symbol loads the file's code object and waits for the copy: only when the value differs from what
   the code object already holds -- the defaults are the compiled-in values, so that a cold command sets nothing) */
#define SEED_PRIO_SETTER(fn, name) \
  void fn(int on) { if (on != name##_host) { HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(name), &on, sizeof(int)));  name##_host = on; } }

#define COARSE_SHIFT 9
#define PK_PAD 4

#ifdef __HIPCC__
/* read containing base offset p (p is inside a read, never on a terminator) */
__device__ __forceinline__ u32 read_of_pos(const DevBlock &b, u32 p)
{ u32 r = b.coarse[p >> COARSE_SHIFT];
  while (b.boff[r + 1] <= p)
    r += 1;
  return r;
}
/* the position word of an index entry -> read and offset of the k-mer's last base in it: free when the block's index
   is packed (rpbits), two dependent look-ups otherwise */
__device__ __forceinline__ void pos_decode(const DevBlock &b, u32 v, u32 *r, u32 *x)
{ if (b.rpbits)
    { *r = v >> b.rpbits;
      *x = v & ((1u << b.rpbits) - 1u);
    }
  else
    { const u32 rr = read_of_pos(b, v);
      *r = rr;
      *x = v - b.boff[rr];
    }
}
__device__ __forceinline__ u32 pos_encode(const DevBlock &b, u32 r, u32 x, u32 p)
{ return b.rpbits ? ((r << b.rpbits) | x) : p; }
#endif

/* pk[w] for w in [-PK_PAD, total/16 + PK_PAD] from bases (which carry 64 padding bytes either side), then as many words
   of the reversed copy: pk points PK_PAD words into an array of 2 * damar_pack_words(total) words */
long long damar_pack_words(u32 total);
void damar_launch_pack_bases(const u8 *bases, u32 total, u32 *pk, hipStream_t st);
/* the block out of its .bps stretch (kmer_index.hip unpack_bps); blk needs boff, coarse, total; bases preset to 4 */
void damar_launch_unpack_bps(const u8 *raw, const u32 *foff, const DevBlock *blk, int comp, u8 *bases, hipStream_t st);

/* kmer_index.hip */
/* codes: u32 (k <= 16) or, with wide != 0, u64 (k <= 32) */
void damar_preload_index(void);  void damar_preload_scan(void);  void damar_preload_sort(void);
void damar_preload_merge(void);  void damar_preload_report(void);
void damar_sort_set_prio(int on);
void damar_merge_set_prio(int on);
void damar_index_set_prio(int on);
void damar_launch_kmer_tuples(const DevBlock *blk, int kmer, u32 nkmers, void *codes, int wide, u32 *pos, hipStream_t st);
/* keep[i] = 1 iff k-mer i lies inside one unmasked stretch of its read (filter.c:474-526) */
void damar_launch_mask_flags(const DevBlock *blk, int kmer, const u32 *pos, u32 n, u32 *keep, hipStream_t st);
/* -b: keep[] (cleared by the caller, blk->total entries) marks the block offsets where a k-mer ends */
void damar_launch_biased_tuples(const DevBlock *blk, int kmer, const int *logbase, void *codes, int wide, u32 *pos,
                                u32 *keep, hipStream_t st);
void damar_launch_suppress_flags(const void *codes, int wide, u32 n, int suppress, u32 *keep, hipStream_t st);
void damar_launch_compact_pairs(const void *k, int wide, const u32 *v, const u32 *keep, const u32 *off, u32 n,
                                void *ko, u32 *vo, hipStream_t st);

/* datander: distance to the previous equal k-mer of the same read, scattered back to position
 * order (dist[k-mer index]); scrub/tandem.c:556-589 + the (read,rpos) re-sort of :1298 */
void damar_launch_tandem_links(const DevBlock *blk, int kmer, const void *codes, int wide, const u32 *pos, u32 n,
                               int *dist, hipStream_t st);

/* seed_merge.hip */
typedef struct
{ const void *acode;  const u32 *apos;  u32 alen;                         /* codes: u32, or u64 when wide */
  const void *bcode;  const u32 *bpos;  u32 blen;
  int  wide;
  int  kbits;
  int  self, comp, identity;
  u32  limit;
  DevBlock ablk, bblk;
  int  pbits, abits;          /* key = bread << (abits+pbits) | aread << pbits | apos, all of it << dbits */
  int  dbits;                 /* > 0: bpos rides in the key's low dbits (packed seeds, no vals array); 0: diag in vals */
} MergeArgs;

typedef struct { u32 b0, b1, ja, ia;                   /* per tile of A entries: its piece of B, the ends of its border runs, */
                 u32 sh, pad[3]; } MergeTile;           /* and the shift that maps (code - first code of the tile) onto < 2048 buckets */

/* The merge is two sweeps over tiles of the A index (seed_merge.hip): COUNT leaves the hits per tile in the workspace
   (damar_merge_tile_counts; with gram != NULL also hitgram[ct] for ct < ngram, filter.c:1039-1165), the caller scans
   them in place (damar_scan_tile_counts) and EMIT writes the seed pairs. */
size_t damar_merge_workspace_bytes(u32 alen);
u32   *damar_merge_tile_counts(void *work, u32 alen);
u32    damar_merge_tiles(u32 alen);
void   damar_launch_merge_count(const MergeArgs *m, void *work, unsigned long long *gram, u32 ngram, hipStream_t st);
/* pid (optional): the read pair of every seed, bread << abits | aread, for the early cut */
void damar_launch_merge_emit(const MergeArgs *m, void *work, u64 nhits, u64 *keys, u32 *vals, u32 *pid, hipStream_t st);
/* the early cut (seed_merge.hip): heads of the runs report_thread enters, on the SORTED pair ids; their pairs into a
   bitmap over the pair ids; the seeds of those pairs out of the unsorted seeds */
void damar_launch_pair_heads_ids(const u32 *pids, u64 nhits, int abits, int minhit, int nshift,
                                 u64 *send, u64 *bits, void *scan_work, u64 *total_dev, u32 *heads, hipStream_t st);
void damar_launch_pair_bitmap(const u32 *pids, const u32 *heads, u32 nheads, int abits, u32 b_lo, u32 b_hi, u32 *bitmap,
                              hipStream_t st);
void damar_launch_seed_cut_count(const u64 *keys, u64 nhits, int pbits, const u32 *bitmap, u32 *tcount, u64 *total_dev,
                                 hipStream_t st);
void damar_launch_seed_cut_scatter(const u64 *keys, const u32 *vals, u64 nhits, int pbits, const u32 *bitmap,
                                   const u32 *toff, u64 *okeys, u32 *ovals, hipStream_t st);
void damar_launch_pair_heads(const u64 *keys, u64 nhits, int pbits, int abits, int minhit, int nshift,
                             u64 *send /* 64 entries of scratch */, u64 *bits, void *scan_work, u64 *total_dev,
                             u32 *heads, hipStream_t st);
/* run heads + screen in one pass (pair_work_mark): the first half leaves the number of work items in *total_dev */
/* (unsorted: the seed sort went over the read pair only -- the screen takes a run's seeds in any order, the caller puts the
   kept heads' runs in order of their A positions (damar_launch_order_runs) unless total_dev[1] != 0: a run too long for that,
   sort over all the bits) */
void damar_launch_pair_work(u64 *keys, const u32 *vals, u64 nhits, int ppos, int dbits, int abits, int minhit, int nshift,
                            u64 *send /* 64 entries of scratch */, u64 *bits, void *scan_work, u64 *total_dev /* [2] */,
                            int binshift, int kmer, int hitmin, u32 b_lo, u32 b_hi, int unsorted, hipStream_t st);
/* (unsorted only) the runs of the work list's heads into the order of their A positions, in place */
void damar_launch_order_runs(u64 *keys, u64 nhits, int ppos, int dbits, const u32 *work, u32 nwork, hipStream_t st);
void damar_launch_pair_work_expand(const u64 *bits, const void *scan_work, u64 nhits, u32 *work, hipStream_t st);
#define WORK_COST_BITS 16
#define WORK_COST_MAX  ((1u << WORK_COST_BITS) - 1)
void damar_launch_work_cost(const u64 *keys, const u32 *vals, u64 nhits, int pbits, int abits, int dbits, const u32 *aboff,
                            const u32 *bboff, const u32 *work, u32 nwork, u32 coarse, u32 *key, u32 *val, hipStream_t st);
void damar_launch_pair_screen(const u64 *keys, const u32 *vals, u64 nhits, int pbits, int dbits, const u32 *heads, u32 nheads,
                              int minhit, int binshift, int kmer, int hitmin, int abits, u32 b_lo, u32 b_hi, u32 *keep, hipStream_t st);
void damar_launch_compact_u32(const u32 *src, const u32 *keep, const u32 *off, u32 n, u32 *out, hipStream_t st);
void damar_launch_compact_index(const u32 *flags, const u32 *off, u64 n, u32 *out, hipStream_t st);

#ifdef __HIPCC__
/* diagonal (apos - bpos) of seed i: from the packed key, or from the vals array of the unpacked layout */
__device__ __forceinline__ int seed_diag(u64 k, const u32 *__restrict__ vals, u64 i, u64 pmask, int dbits)
{ if (dbits)
    return (int) ((k >> dbits) & pmask) - (int) (k & ((1ull << dbits) - 1));
  return (int) vals[i];
}
#endif

/* report.hip */
typedef struct
{ int abpos, bbpos, aepos, bepos, diffs;
  int atlen, btlen;
  int aread, bread;       /* block-local ids */
  u32 item, seq;          /* work item and order of discovery inside it */
  u32 toff;               /* offset of the A trace in the trace pool; B trace follows */
} LaRecord;

typedef struct
{ /* inputs */
  const u64 *keys;  const u32 *vals;  u64 nhits;
  const u32 *work;  u32 nwork;
  DevBlock ablk, bblk;
  int  pbits, abits, dbits;        /* seed key layout, see MergeArgs */
  int  kmer, hitmin, binshift, minhit;
  int  comp, self, symmetric, minover, hgap_min;
  int  tspace, ave_path, reach;
  const short *score, *table;      /* SCORE[32768], TABLE[32768] */
  int  mscore, dscore;             /* the two column scores the tables are built from (align.c:282-284) */
  /* per-slot scratch */
  void *state;   u64 state_stride;   int span;        /* ping-pong diagonal state: rings of `span` (2^n) diagonals */
  int  *marks;   u64 marks_stride;                    /* NA/NB                     */
  void *cells;   u32 cell_cap;                        /* pebbles                   */
  int  *buckets; u64 bucket_stride;  int bwidth;      /* score|lastp|lasta          */
  int  bucket_bits;                                   /* bits needed for bucket - mindiag */
  u16  *ttmp;    u32 ttmp_stride;                     /* 2 centred trace buffers   */
  /* outputs */
  LaRecord *recs;  u32 rec_cap;
  u16  *tpool;     u32 tpool_cap;
  u32  *widemap;                   /* one bit per work item: the two-pair kernel gave the pair up (pebbles beyond the packed format);
                                      NULL: no wide path behind this launch (the limits are fatal as until round 4) */
  void *wcells;    u32 wcell_cap;  /* the wide kernel's 16-byte pebbles: wcell_cap per slot */
  u32  cell_max;                   /* the largest pool a slot can get (DAMAR_MAX_CELLS; a test hook lowers it): a pair that overflows it is the wide kernel's */
  int  t8, t8max;                  /* t8: trace values leave as BYTES (tspace <= 125: align.c:3375-3396 Compress_TraceTo8 on the device, the
                                      pool then holds tpool_cap bytes' worth of values in its first half); a value above t8max (255) raises
                                      DAMAR_ERR_T8 and the host repeats the launch with 16-bit values, so that the reference's own check
                                      decides (it looks only at the records that are written) */
  const u32 *order; /* processing order of the work items (largest first), or NULL          */
  u32  *counters;  /* [1] records, [2] trace words, [3] error flags, [6] [7] where a wave gave up; shared by the jobs of a launch */
  u32  *cursor;    /* next work item of THIS job (counters + DAMAR_CNT_CURSOR + job) */
  u32  *nfilt;     /* seed hits of THIS job      (counters + DAMAR_CNT_NFILT + job)  */
  int   job;       /* index of this job in the launch: rides in the top byte of LaRecord.seq */
} ReportArgs;

/* One launch of a report kernel works through up to DAMAR_MAX_JOBS comparisons (ReportArgs each, in constant memory:
   wave-uniform reads of a job's fields are scalar loads).  Every wavefront starts with job (block index mod njobs) and
   moves on to the next when a job's queue is empty, so the long alignments of ALL jobs start at once and the tail of
   the launch (waiting for the longest alignment) is paid once per launch, not once per comparison. */
#define DAMAR_MAX_JOBS 32
#define DAMAR_MAX_TSPACE 8192     /* consecutive pebbles of a chain then differ by < 2^15 diagonals and < 2^16 waves */
#define DAMAR_CNT_CELLS     8     /* counters[8..9]   (64 bits): band cells of the launch's wave steps (packed kernel) */
#define DAMAR_CNT_HALFSTEPS 10    /* counters[10..11] (64 bits): wave steps, counted per half-wavefront (= per alignment pass)   */
#define DAMAR_CNT_ITERS     12    /* counters[12..13] (64 bits): iterations of the wave loop (each steps one or two halves)      */
#define DAMAR_CNT_WIDE    14      /* counters[14]: read pairs left to the wide kernel (report_wide_kernel) */
#define DAMAR_ITEM_WIDE   0x80000000u   /* in LaRecord.item: written by the wide kernel */
#define DAMAR_CNT_CURSOR  16      /* counters[16 + job]: next work item of a job  */
#define DAMAR_CNT_NFILT   (DAMAR_CNT_CURSOR + DAMAR_MAX_JOBS)      /* counters[.. + job]: its seed hits */
#define DAMAR_COUNTER_WORDS (DAMAR_CNT_NFILT + DAMAR_MAX_JOBS)
#define DAMAR_SEQ_BITS 24

#define DAMAR_ERR_CELLS   1u
#define DAMAR_ERR_RECS    2u
#define DAMAR_ERR_TPOOL   4u
#define DAMAR_ERR_BAND    8u
#define DAMAR_ERR_T8     32u    /* a trace value does not fit a byte (see ReportArgs.t8) */
#define DAMAR_ERR_WIDE   16u    /* a band outgrew the ring of diagonals of the slot buffers: relaunch with a larger ring */

int  damar_report_waves_per_simd(void);
int  damar_report2_waves_per_simd(void);
int  damar_report2_slots_per_wave(void);     /* scratch slots (read pairs in flight) per wavefront of the packed kernel */
void damar_launch_report(const ReportArgs *jobs, int njobs, int nslots, hipStream_t st);
void damar_launch_report_wide(const ReportArgs *jobs, int njobs, int nslots, hipStream_t st);
/* datander report: one work item per read of a.ablk, dist as produced by damar_launch_tandem_links */
void damar_launch_tandem_report(const ReportArgs *a, const int *dist, int nslots, hipStream_t st);

/* batch Local_Alignment for tests: task i = (aread, bread, diag, anti) */
typedef struct { int aread, bread, diag, anti; } LaTask;
void damar_launch_la_batch(const ReportArgs *a, const LaTask *tasks, u32 ntasks, int nslots, hipStream_t st);
/* several read pairs (or batch tasks, tasks != NULL) per wavefront: report_packed.h; nslots a multiple of damar_report2_slots_per_wave() */
void damar_launch_report2(const ReportArgs *jobs, int njobs, const LaTask *tasks, u32 ntasks, int nslots, hipStream_t st);
void damar_launch_tandem_report2(const ReportArgs *a, const int *dist, int nslots, hipStream_t st);     /* datander through the same kernel */
/* the reads / tasks the two-pair kernel left to the 16-byte pebbles (a->widemap, reads beyond DAMAR_MAX_MARKS spacings) */
void damar_launch_tandem_report_wide(const ReportArgs *a, const int *dist, int nslots, hipStream_t st);
void damar_launch_la_batch_wide(const ReportArgs *a, const LaTask *tasks, u32 ntasks, int nslots, hipStream_t st);
#define DAMAR_MAX_MARKS 16000         /* trace-grid indexes ride in the top 14 bits of a chain head (report.hip PK_HBITS) */
#define DAMAR_MAX_CELLS (1u << 18)    /* pebbles per slot: 18 bits of a chain head */

u64 damar_report_state_stride(int span);

/* trace_pts.hip: Compute_Trace_PTS for batches of records (align.c:5577-5692, 4892-5261) */
typedef struct
{ u32 aread, bread;       /* block-local read ids */
  u32 flags;              /* bit 0 = B complemented (COMP_FLAG), bit 1 = A and B are one buffer */
  int abpos, bbpos, aepos, bepos;
  u32 poff;               /* first trace-point value of the record in the point array */
  int tlen;               /* trace-point values (pairs diffs, B length) */
  u32 seg0;               /* index of the record's first segment */
  u32 stage0;             /* first staging slot of the record (segment s gets dmax + |M_s - N_s| slots) */
  int dmax;               /* largest trace-point difference count of the record (align.c:5614-5621) */
  u32 slots;              /* staging slots set aside for the record */
} TraceRecIn;

typedef struct
{ u32 apos, bpos;         /* first base of the segment in the blocks' base arrays (B: last, if complemented) */
  int a0, b0;             /* the same as offsets into the two reads (script values)  */
  u32 mn;                 /* M | N << 16                                             */
  u32 flags;              /* TraceRecIn.flags & 3, bit 2 = void (record failed the bounds check), dmax << 8 */
  u32 stage;              /* slot of the segment in the staging area                 */
  u32 rec;
} TraceSeg;

typedef struct
{ const TraceSeg *segs;
  const u32 *list;  u32 nwork;          /* segment ids to do (NULL = 0 .. nwork-1)     */
  const u8 *abases, *bbases;
  const u32 *apk, *bpk;               /* the same bases at 2 bits (DevBlock.pk)      */
  short *vf;  signed char *hf;  u32 cap; /* per-thread stripes of cap cells             */
  int *stage;  u32 *count;  int *dist;
  int *mid;                             /* kind 1: A and B offset of each segment's mid point */
  u32 *over;  u32 over_cap;  u32 *nover;  u32 *need;  u32 *err;
  u32 *next;                            /* slot kernel: next batch of 64 segments to hand out */
} TraceArgs;

#define DAMAR_TRACE_ERR_POINTS 1u       /* trace point out of bounds (align.c:5575)   */
#define DAMAR_TRACE_ERR_ALIGN  2u       /* bad alignment between trace points (:4890) */
#define DAMAR_TRACE_ERR_INTERNAL 4u      /* staging bound of the mid-point pieces violated */

/* pts: the records' trace points as stored in the .las (tbytes 1 or 2) */
/* key / val: per segment its own difference count (<= 255) and its index, for the work order */
void damar_launch_trace_layout(const TraceRecIn *recs, u32 nrecs, const void *pts, int tbytes, int tspace,
                               const DevBlock *ablk, const DevBlock *bblk, TraceSeg *segs, u32 *key, u32 *val, u32 *err,
                               hipStream_t st);
void damar_launch_trace_waves(const TraceArgs *t, int mode, int kind, u32 nblocks, hipStream_t st);
size_t damar_trace_slot_area_cells(void);
size_t damar_trace_slot_vf_bytes(int mode, int kind);
void damar_launch_trace_waves_slots(const TraceArgs *t, int mode, int kind, u32 nblocks, hipStream_t st);
void damar_launch_trace_gather(const TraceRecIn *recs, u32 nrecs, int mid, const u32 *count, const int *dist, u32 *segoff,
                               u32 *tlen, int *diffs, hipStream_t st);
void damar_launch_trace_mid_layout(const TraceRecIn *recs, u32 nrecs, const TraceSeg *segs, const int *mid,
                                   const DevBlock *ablk, const DevBlock *bblk, TraceSeg *out, u32 *key, u32 *val, u32 *err,
                                   hipStream_t st);
void damar_launch_trace_pack(const TraceSeg *segs, u32 nsegs, const u32 *count, const u32 *segoff, const u32 *recoff,
                             const int *stage, int *script, hipStream_t st);
#endif

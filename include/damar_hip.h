/* damar_hip.h -- device-resident entry points of libdamar_hip.so (C ABI, plain pointers
 * and sizes only).
 *
 * The reference has no FFI boundary: dalign/daligner.c calls Sort_Kmers / Match_Filter
 * (dalign/filter.h:64-70, declared for this library in damar_filter.h) directly.  Those
 * two calls hand over HOST buffers, so each of them pays a PCIe copy of the block's
 * bases.  The functions below split them at the PCIe boundary so that a caller that
 * keeps blocks resident in HBM (the multi-GPU block-pair scheduler, bench.py) can time
 * and reuse the device-side work alone:
 *
 *   Sort_Kmers(block,&len)            == damar_block_upload + damar_index_build
 *   Match_Filter(..., atab, btab, ..) == damar_match (+ damar_index_free(btab) when
 *                                        atab != btab, filter.c:2722-2731, 2880-2881)
 *
 * All functions are blocking, single-caller and non-reentrant like the reference's
 * filter.c (file-scope state); fatal errors print to stderr and
 * exit(1) like the reference (db/DB.h:84-86).
 */
#ifndef DAMAR_HIP_H
#define DAMAR_HIP_H

#include "damar_db.h"
#include "damar_align.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Select the GPU (0-based HIP ordinal) and create the stream; returns the number of
 * visible devices.  Called implicitly with device 0 (or $DAMAR_DEVICE) on first use.
 * Exits with a message if no HIP device is present: there is no CPU fallback. */
int   damar_hip_init(int device);
/* NUMA node of the host memory next to GPU `device`, or -1 if the host does not say (callable before damar_hip_init:
 * the node scheduler binds a worker's threads and memory policy to it first, host/daligner.c: node_worker) */
int   damar_hip_numa_node(int device);
const char *damar_hip_device_name(void);
void  damar_hip_sync(void);          /* hipDeviceSynchronize on the selected GPU */
/* Grow (and release again) the process's HBM footprint by `gigabytes`: a cold process pays ~25 ms per GB the first
 * time; called on a thread of its own next to reading the input, the real allocations find the memory ready. */
void  damar_prewarm(int gigabytes);

/* A read block resident in HBM: bases (1 B/base with the reference's 4-terminators,
 * db/DB.c:1562-1605), read offsets, coarse position->read table. */
typedef struct damar_dev_block damar_dev_block;
damar_dev_block *damar_block_upload(const HITS_DB *block);
/* The same on a stream of its own, remembered by the address of block->bases until Sort_Kmers(block) takes
 * it: lets a second host thread upload the next block while the GPU works (one caller thread at a time). */
void damar_block_preload(const HITS_DB *block);
/* damar_block_upload on that second stream, returning the block: for a host thread that prepares blocks ahead. */
damar_dev_block *damar_block_upload_bg(const HITS_DB *block);
/* A block the host keeps PACKED (damar_read_block_packed, include/damar_db.h): its .bps stretch goes up once per strand
 * and the device unpacks it, comp = 1 into the reverse complement (daligner.c:511-570) -- no host unpacking, no host
 * complement, a quarter of the PCIe bytes.  `block` carries reads / freq / tracks of the strand wanted (for comp 1: what
 * damar_complement_copy made of the packed block); it stays registered for the host tail until damar_packed_forget. */
damar_dev_block *damar_block_upload_packed(const HITS_DB *block, const damar_packed *pk, int comp);
void             damar_packed_forget(const HITS_DB *block);
void             damar_block_free(damar_dev_block *blk);

/* The opaque index that Sort_Kmers returns (filter.c:753-994): sorted k-mer codes,
 * base offsets and the code prefix table, all in HBM.  `own_block` != 0 makes
 * damar_index_free release the block too. */
typedef struct damar_dev_index damar_dev_index;
damar_dev_index *damar_index_build(damar_dev_block *blk, int own_block, int *len);
void             damar_index_free(damar_dev_index *idx);
uint64_t         damar_index_bytes(const damar_dev_index *idx);   /* HBM the index holds, for residency caps */
uint64_t         damar_block_bytes(const damar_dev_block *blk);   /* HBM a resident block holds (bases, packed bases, tables) */
void             damar_hbm_info(uint64_t *free_bytes, uint64_t *total_bytes);   /* of the selected GPU (hipMemGetInfo) */
void             damar_pool_trim(void);                            /* release the library's parked index buffers */
/* Test hook: copy the index back as reference-layout KmerPos records
 * {uint64 code; int rpos; int read} (filter.c:121-126), out must hold *len records. */
void             damar_index_download(const damar_dev_index *idx, void *out);

/* Device part + host tail of Match_Filter (filter.c:2519-2929); records are appended
 * to OVL_IO_Buffer(spec)[0].  counts[0..2] = seed pairs, seed hits (Local_Alignment
 * calls), confirmed records -- the three numbers daligner -v prints. */
void damar_match(const HITS_DB *ablock, const HITS_DB *bblock,
                 damar_dev_index *aidx, damar_dev_index *bidx,
                 int self, int comp, Align_Spec *spec, int64 *counts);

/* Several comparisons behind ONE launch of the report kernel each time (4 per launch unless DAMAR_BATCH says otherwise, at most 16; fewer when their seed
 * pairs would pass a quarter of HBM): the seed stages run one after the other, then every wavefront works through the
 * work lists of all of them, so the wait for the longest alignment at the end of a launch is paid once per launch instead
 * of once per comparison.  Records are appended as if damar_match had been called for jobs[0], jobs[1], ... in order
 * (a daligner plan line, filter.c:2519 called per (B block, orientation) by daligner.c:993-1021, is one such batch);
 * counts as for damar_match. */
typedef struct
{ const HITS_DB *ablock, *bblock;
  damar_dev_index *aidx, *bidx;
  int self, comp;
  Align_Spec *spec;
  int64 counts[3];
} damar_match_job;
void damar_match_batch(damar_match_job *jobs, int njobs);

/* Restrict the following damar_match / Match_Filter calls to the read pairs whose B read (block-local
 * index) lies in [lo, hi); hi < 0 lifts the restriction.  The records of a range are exactly those the
 * unrestricted call writes for these B reads (the merge and the sort still cover the whole pair), so a
 * scheduler can split one block pair over several GPUs and merge the parts' files ("split big
 * pairs ... legal because report work is independent per (bread,aread) run"). */
void damar_set_bread_range(int lo, int hi);

/* Asynchronous host tail: with damar_set_async(1) the per-read-pair tail of damar_match /
 * Match_Filter (redundancy handling, trace compression, buffer append) and the sort + write of
 * damar_write_overlaps run on one worker thread in submission order while the GPU proceeds
 * with the next block pair.  Call damar_async_drain() before releasing the blocks or the
 * Align_Spec, and before reading record counts (counts[2] of damar_match is 0 in this mode;
 * damar_async_totals returns the sum). */
void damar_set_async(int on);
void damar_async_drain(void);
void damar_async_totals(int64 *ncheck, double *tail_ms, double *write_ms);
/* In asynchronous mode the report launch of a damar_match_batch call is also left in flight when the call returns
 * (its own stream; the seed stages of the NEXT call run beside it -- DAMAR_OVERLAP=0 turns that off): counts[1] of the
 * comparisons of that last launch is then 0 at return.  Totals since the last call (drains first): seed hits, and the
 * report kernel's milliseconds and launches. */
void damar_async_counts(int64 *seed_hits, double *report_ms, int64 *launches);
double damar_async_d2h_ms(void);        /* time of the asynchronous record downloads since the last call */
/* What the Local_Alignment waves (align.c:409-1898) of the report launches stepped through since the last call (drains
 * first): band cells = sum over all wave steps of the diagonals computed (the reference's WAVE_STATS unit, align.c:81,
 * 353-368), wave steps counted per alignment pass, and iterations of the kernel's wave loop (each steps one or two passes:
 * half_steps / iterations of 2 means no idle half-wavefront). */
void damar_wave_totals(int64 *cells, int64 *half_steps, int64 *iterations);
/* Write_Overlap_Buffer + Reset_Overlap_Buffer (daligner.c:1020-1021), queued in async mode */
void damar_write_overlaps(Align_Spec *spec, const char *dirName1, const char *dirName2,
                          const char *ablock, const char *bblock, int lastRead);

/* datander (scrub/tandem.h:58-60): the 4-argument parameter call under a library-unique
 * name, and Match_Self with the block already resident in HBM.  counts = k-mers, seed hits,
 * confirmed records. */
int  damar_tandem_set_params(int kmer, int binshift, int hitmin, int nthreads);
void damar_match_self(const HITS_DB *ablock, damar_dev_block *blk, Align_Spec *spec, int64 *counts);
/* scrub/tandem.h:60, scrub/tandem.c:1182 */
void Match_Self(char *aname, HITS_DB *ablock, Align_Spec *settings);

/* Test hook: seed pairs of the last damar_match call as reference-layout SeedPair
 * records {int diag, apos, aread, bread} (filter.c:128-134) in sorted order.
 * Returns the number of seed pairs; copies at most `cap` records. */
int64 damar_last_seeds(void *out, int64 cap);

/* The cap on mutual k-mer matches per code that the last damar_match applied (filter.c:2634-2702:
 * 10000 unless the host memory limit forces it lower; INT32_MAX when MEM_LIMIT is 0). */
int damar_last_limit(void);

/* -b: the reference derives its log base weights from the first block a PROCESS sorts and keeps
 * them (filter.c:774-789).  A caller that runs several jobs in one process calls this between
 * them to get what separate daligner processes would compute. */
void damar_bias_reset(void);

/* Test hook: batch Local_Alignment (align.c:1904 with low == hgh == diag, no borders)
 * on the GPU.  tasks[4*i..] = aread, bread, diag, anti (block-local read ids).
 * paths[12*i..] = A-view abpos,bbpos,aepos,bepos,diffs,tlen then the same for the
 * B-view; traces of task i start at trace_off[2*i] (A) and trace_off[2*i+1] (B) in
 * `traces` (capacity trace_cap values).  Returns 0, or -1 if trace_cap is too small. */
int damar_local_alignment_batch(damar_dev_block *ablk, damar_dev_block *bblk, int comp,
                                Align_Spec *spec, const int *tasks, int ntasks,
                                int *paths, int64 *trace_off, uint16 *traces, int64 trace_cap);

/* SURVEY 8(f)4, the next consumer of the records: Compute_Trace_PTS (align.c:5577-5692 + iter_np
 * :4892-5261) for every record of an Overlap array, the way utils/LAshow.c:245-262 calls it per record.
 * ablk / bblk hold the A and B reads (B forward: the complement of COMP records is read in place);
 * ovls[i].aread / bread are DB read ids, afirst / bfirst the ids of the blocks' first reads;
 * ovls[i].path.trace = the trace points as stored in the .las, tbytes (1 or 2) bytes per value.
 * same != 0 applies the one-buffer rule of align.c:4933-4951 (LAshow never does).
 * On success returns 0: *script = malloc'ed array of all edit scripts, record i at
 * [soff[i], soff[i+1]) (soff has novl + 1 entries), diffs[i] = its summed segment distances.
 * Returns 1 after the reference's message where the reference exits (trace point out of bounds,
 * bad alignment between trace points). */
int  damar_trace_pts(damar_dev_block *ablk, int afirst, damar_dev_block *bblk, int bfirst,
                     const Overlap *ovls, int64 novl, int tbytes, int tspace, int mode, int same,
                     int64 *soff, int *diffs, int **script);
/* The same for Compute_Trace_MID (align.c:5694-5830 + middle_np :5263-5573, corrector/LAcorrect.c:545) */
int  damar_trace_mid(damar_dev_block *ablk, int afirst, damar_dev_block *bblk, int bfirst,
                     const Overlap *ovls, int64 novl, int tbytes, int tspace, int mode, int same,
                     int64 *soff, int *diffs, int **script);
/* ms[4]: trace_waves kernel, all kernels of the call (HIP events), whole call, inside the batches (wall);
 * cnt[4]: records, segments, segments deferred to the large-stripe launch, script values */
void damar_trace_last(double *ms, int64 *cnt);
void damar_trace_release(void);          /* frees the cached device buffers of damar_trace_pts */

/* Phase timings (milliseconds, HIP events on the library's stream) of the last
 * damar_index_build / damar_match: see DAMAR_T_* below. */
enum { DAMAR_T_TUPLES = 0, DAMAR_T_KSORT, DAMAR_T_TABLE, DAMAR_T_MERGE, DAMAR_T_SSORT,
       DAMAR_T_WORK, DAMAR_T_REPORT, DAMAR_T_D2H, DAMAR_T_TAIL, DAMAR_T_COUNT };
void damar_last_timings(double *ms /* [DAMAR_T_COUNT] */);
/* bytes, files, records and aligned base pairs (sum of aepos - abpos) the .las writers have produced since the process started (host/las.c; with DAMAR_LAS_KEEP=<list>
   only the files whose path ends in a line of <list> reach the file system, the rest /dev/null: measurement of long plans) */
void damar_las_totals(int64 *out /* [4] */);

/* Counters of the last damar_match / damar_match_batch (summed over its comparisons): [0] seed pairs, [1] work items
 * (read pairs entered), [2] Local_Alignment calls, [3] records from the device, [4] trace values, [5] launches of the
 * report kernel (re-launches after a buffer overflow included). */
void damar_last_counters(int64 *c /* [8] */);

/* Sort kernels alone, for the roofline measurement: sorts n (u32 key, u32 payload)
 * pairs resident in HBM on `nbits` key bits `reps` times and returns the average
 * milliseconds per sort (HIP events). */
double damar_bench_sort_u32(uint32_t n, int nbits, int reps, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif

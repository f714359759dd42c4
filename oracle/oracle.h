/* oracle.h -- CPU restatement of the daligner overlap hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product (libdamar_hip.so, the daligner host binary)
 * never links, loads or executes this code.
 *
 * Parity is PINNED: every stage below is checked in tests/ against the real
 * reference compiled from /root/reference by oracle/Makefile.ref (oracle/_ref/),
 * and against the golden .las / stage fixtures under tests/golden/ that those
 * binaries produced (the reference ships no tests or vectors of its own,
 * SURVEY.md section 4).
 *
 * Each function cites the reference file:line whose behaviour it restates.  The
 * restatement is deliberately written in the data-parallel shape the HIP kernels
 * use (whole-array passes, wave steps that read the previous wave and write the
 * next) instead of the reference's threaded in-place sweeps.
 */
#ifndef DAMAR_ORACLE_H
#define DAMAR_ORACLE_H

#include "damar_db.h"
#include "damar_align.h"

#ifdef __cplusplus
extern "C" {
#endif

/* filter.c:121-134 (little-endian field order) */
typedef struct { uint64 code; int rpos; int read; } OKmer;
typedef struct { int diag; int apos; int aread; int bread; } OSeed;

typedef struct
{ int kmer;        /* -k */
  int binshift;    /* -w */
  int suppress;    /* -t (0 = off) */
  int hitmin;      /* -h */
  int nthreads;    /* -j, rounded down to 2^n as filter.c:192-198 */
  int minover;     /* 2 * -l, as daligner.c:861 */
  int hgap_min;    /* always 0 through the reference CLI */
  int symmetric;   /* !-A */
  int identity;    /* -I */
  int64 mem_limit; /* bytes; 0 = no cap (filter.c:2634-2702) */
  int   biased;    /* -b (filter.c:549-688); the log weights are fixed by the first block sorted */
} OParams;

/* K1+K2+K3: filter.c:458-547 tuple_thread, :328-435 lex_sort, :700-751 -t compaction,
 * :753-994 Sort_Kmers.  Returns malloc'ed array of *len records in (code, read, rpos)
 * order followed by the two sentinels, or NULL if the block has no k-mers. */
OKmer *oracle_sort_kmers(const HITS_DB *block, const OParams *prm, int *len);

/* K4: filter.c:1039-1165 count_thread, :1170-1358 merge_thread, :2634-2699 limit,
 * :2776 seed sort.  Returns malloc'ed seeds in (bread, aread, apos, bpos) order with
 * one sentinel record after *nhits. */
OSeed *oracle_seed_pairs(const HITS_DB *ablock, const HITS_DB *bblock,
                         const OKmer *asort, int alen, const OKmer *bsort, int blen,
                         int self, int comp, const OParams *prm, int64 *nhits, int *limit_out);

/* K6: align.c:1904-2097 Local_Alignment for the call shape of filter.c:2316
 * (low == hgh == diag, no borders).  Fills both paths; traces are written into
 * caller arrays of at least 2*(max(alen,blen)/tspace+2)+2 values each and the
 * paths' trace pointers point into them. */
typedef struct
{ int64 waves;      /* wave steps, forward + reverse */
  int64 cells;      /* sum of band widths over waves */
  int   maxband;
  int64 pebbles;
  int   empty_band; /* a wave ran on an empty band (undefined in the reference) */
  int64 bandhist[130]; /* wave steps by number of diagonals computed in the step (129 = more) */
  int64 dirs, dirs_over31, steps_after_over31;   /* directions, those that ever compute > 31 diagonals, their steps from then on */
  int   cur_over;
  /* per PASS (one direction of one Local_Alignment): the widest step of the pass and its cells, classified when the
     next pass starts: passes / cells of passes whose widest step computed <= 13, <= 14, <= 16, <= 29 diagonals, more */
  int   pass_max;
  int64 pass_cells, pass_n[5], pass_cellsum[5];
  /* a 16-lane quarter mode with parking, simulated per pass: steps spent at <= 14 computed diagonals, and the number of
     times a pass crosses from <= 12 (narrow again) to > 14 (must have a half) */
  int   pass_wide;
  int64 promotions, steps_narrow, steps_wide;
} OWaveStats;

void oracle_local_alignment(const char *aseq, int alen, const char *bseq, int blen,
                            uint32 flags, int diag, int anti, Align_Spec *spec,
                            Path *apath, Path *bpath, uint16 *atrace, uint16 *btrace,
                            OWaveStats *stats);

/* K5+K6+K7+K8: filter.c:2128-2511 report_thread over all read pairs, results
 * appended to OVL_IO_Buffer(spec)[0].  nfilt/ncheck as printed by -v. */
void oracle_report(const HITS_DB *ablock, const HITS_DB *bblock, const OSeed *hits, int64 nhits,
                   int self, int comp, const OParams *prm, Align_Spec *spec,
                   int64 *nfilt, int64 *ncheck, OWaveStats *stats);

/* filter.c:2519-2929 Match_Filter = oracle_seed_pairs + oracle_report */
void oracle_match_filter(const HITS_DB *ablock, const HITS_DB *bblock,
                         const OKmer *asort, int alen, const OKmer *bsort, int blen,
                         int self, int comp, const OParams *prm, Align_Spec *spec,
                         int64 *counts /* nhits, nfilt, ncheck */, OWaveStats *stats);

/* datander: scrub/tandem.c:1182-1428 Match_Self (k=12, w=4, h=35, l=500 by default,
 * Align_Spec built with reach = 0, scrub/datander.c:141-146, 251).  counts = k-mers, seed
 * hits, confirmed records. */
void oracle_match_self(const HITS_DB *block, const OParams *prm, Align_Spec *spec,
                       int64 *counts, OWaveStats *stats);

/* (f)4, the next consumer of the records: align.c:5577-5692 Compute_Trace_PTS + :4892-5261 iter_np.
 * path->trace holds the 16-bit trace-point pairs; script (at least aepos-abpos + bepos-bbpos values)
 * receives the edit script (negative = A position, positive = B position, 1-based), *diffs the summed
 * segment distances.  mode -1 / 0 / 1 = LOWERMOST / GREEDIEST / UPPERMOST.  Returns the script length, or
 * -1 where the reference exits ("Bad alignment between trace points"). */
int oracle_compute_trace_pts(const char *aseq, int alen, const char *bseq, int blen, const Path *path,
                             int tspace, int mode, int *script, int *diffs);
/* align.c:5694-5830 Compute_Trace_MID + :5263-5573 middle_np: the same between the segments' mid points */
int oracle_compute_trace_mid(const char *aseq, int alen, const char *bseq, int blen, const Path *path,
                             int tspace, int mode, int *script, int *diffs);

/* Redundancy handling (filter.c:1573-2077): the oracle's own statement in oracle/redundancy.c, behind the interface of
 * damar_amd/csrc/host/damar_host.h (the product's is damar_amd/csrc/host/redundancy.c and is not linked here). */

#ifdef __cplusplus
}
#endif
#endif

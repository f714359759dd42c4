/* damar_host.h -- internal host-side declarations shared by the C-ABI shim, the
 * daligner driver and the per-pair tail (redundancy.c, bridge.c). */
#ifndef DAMAR_HOST_H
#define DAMAR_HOST_H

#include "damar_db.h"
#include "damar_align.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A local alignment of one read pair while it is still being post-processed: the
 * reference keeps Path records whose trace field is an offset into a growing
 * uint16 pool (filter.c:1369-1374 Trace_Buffer, :2353-2357). */
typedef struct
{ int   tlen, diffs;
  int   abpos, bbpos;
  int   aepos, bepos;
  int64 toff;
} damar_path;

typedef struct
{ uint16 *val;
  int64   top, max;
} damar_tpool;

int64 damar_tpool_push(damar_tpool *tp, const uint16 *src, int n);

/* how often the rare branches of the host tail ran (tests assert the fixtures reach them) */
extern int64 damar_stat_redundancy_calls, damar_stat_fusions, damar_stat_bridges;

/* What Bridge needs beyond the paths: the two sequences (filter.c:1998, 2025). */
typedef struct
{ const char *aseq, *bseq;
  int         alen, blen;
} damar_bridge_ctx;

/* bridge.c: filter.c:1456-1571 Compute_Bridge_Path + :1747-1802 Bridge +
 * :1444-1454 Check_Bridge for one candidate (path1,path2).  Returns non-zero if the
 * candidate was skipped. */
int damar_bridge_pair(const damar_bridge_ctx *ctx, damar_path *jp, damar_path *kp,
                      damar_path *p1, damar_path *p2, damar_path *b1, damar_path *b2,
                      int aovl, int bovl, int comp, int ts, damar_tpool *tp,
                      damar_path *bm, int j);

void damar_bridge_release(void);       /* frees the calling thread's work buffers */

int  damar_handle_redundancies(damar_path *am, int n, damar_path *bm, int comp, int ts,
                               damar_tpool *tp, const damar_bridge_ctx *bridge);

void damar_emit_pair(damar_path *am, int na, damar_path *bm, int nb, damar_tpool *tp,
                     int comp, int ts, int aread, int bread,
                     const damar_bridge_ctx *bridge, Overlap_IO_Buffer *obuf, int64 *nrec);

int damar_append_overlap_buffer(Overlap_IO_Buffer *dst, const Overlap_IO_Buffer *src);

/* Detached overlap buffers for a writer thread (las.c) */
typedef struct { int trace_space, nthreads, symmetric, only_identity; } damar_write_params;
Overlap_IO_Buffer *damar_detach_overlap_buffers(Align_Spec *spec, damar_write_params *p);
void damar_write_detached(const damar_write_params *p, Overlap_IO_Buffer *bufs,
                          const char *dir1, const char *dir2, const char *ablock, const char *bblock, int lastRead);

#ifdef __cplusplus
}
#endif
#endif

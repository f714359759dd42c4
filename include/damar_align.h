/* damar_align.h -- alignment records and the Align_Spec / overlap-buffer interface.
 *
 * Same names, argument meaning and struct layouts as the reference's
 * dalign/align.h (Path :125-132, Alignment :172-180, Overlap :358-364,
 * Overlap_IO_Buffer :402-414; functions :223-259, :416-432) for the subset that
 * dalign/daligner.c and dalign/filter.c use.  A reference-side caller can link
 * against libdamar_hip.so in place of align.c for these entry points.
 */
#ifndef DAMAR_ALIGN_H
#define DAMAR_ALIGN_H

#include "damar_db.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TRACE_XOVR 125          /* align.h:58  trace values fit a byte up to this spacing */
#define COMP_FLAG  0x1          /* align.h:154 */
#define ACOMP_FLAG 0x2
#define COMP(x)    ((x) & COMP_FLAG)
#define ACOMP(x)   ((x) & ACOMP_FLAG)

typedef struct
{ void *trace;
  int   tlen;
  int   diffs;
  int   abpos, bbpos;
  int   aepos, bepos;
} Path;

typedef struct
{ Path  *path;
  uint32 flags;
  char  *aseq;
  char  *bseq;
  int    alen;
  int    blen;
} Alignment;

typedef struct
{ Path   path;
  uint32 flags;
  int    aread;
  int    bread;
} Overlap;

typedef struct
{ int      tbytes;
  uint64   tmax;
  uint64   ttop;
  void    *trace;
  int      no_trace;
  int      omax;
  int      otop;
  Overlap *ovls;
} Overlap_IO_Buffer;

typedef void Align_Spec;
typedef void Work_Data;

/* align.c:252-318.  Builds the 2 x 32768 int16 trim tables on the host and, when a
 * GPU is present, mirrors them to HBM on first use by Match_Filter. */
Align_Spec *New_Align_Spec(double ave_corr, int trace_space, float *freq, int nthreads,
                           int symmetric, int only_identity, int no_trace_points, int reach);
void        Free_Align_Spec(Align_Spec *spec);
int         Trace_Spacing(Align_Spec *spec);
double      Average_Correlation(Align_Spec *spec);
float      *Base_Frequencies(Align_Spec *spec);
int         Overlap_If_Possible(Align_Spec *spec);
int         Num_Threads(Align_Spec *spec);
int         Only_Identity(Align_Spec *spec);
int         Symmetric(Align_Spec *spec);

/* align.h:223-259.  Local_Alignment runs the same wave kernel as Match_Filter for ONE pair of
 * sequences; built for the call shape of filter.c:2316 (low == hgh = seed diagonal, lbord and
 * hbord < 0).  It fills align->path (A-view, trace = uint16 pairs) and returns the B-view path;
 * both point into `work` and stay valid until the next call with it (align.h:245-249). */
Work_Data *New_Work_Data(void);
void       Free_Work_Data(Work_Data *work);
Path      *Local_Alignment(Alignment *align, Work_Data *work, Align_Spec *spec,
                           int low, int hgh, int anti, int lbord, int hbord);

/* align.h:279-283, align.c:5577-5692: expand the trace points of align->path (16-bit pairs, as after
 * Decompress_TraceTo16; bseq already complemented for COMP records) into the edit script of the
 * alignment: a negative value -x = a dash before A[x], a positive value x = a dash before B[x] (1-based),
 * path->trace then points into `work`, path->tlen / path->diffs are updated.  Computed on the GPU by the
 * kernels of the batch entry damar_trace_pts (damar_hip.h), which is the one to use for whole .las files. */
#define LOWERMOST -1
#define GREEDIEST  0
#define UPPERMOST  1
int Compute_Trace_PTS(Alignment *align, Work_Data *work, int trace_spacing, int mode);
/* align.c:5694-5830: the same contract, the script computed between the MID points of the trace-point
 * segments (corrector/LAcorrect.c:545); path->diffs reproduces the reference's sum. */
int Compute_Trace_MID(Alignment *align, Work_Data *work, int trace_spacing, int mode);

/* align.c:5969-6102, 6166-6380 */
Overlap_IO_Buffer *CreateOverlapBuffer(int nthreads, int tbytes, int no_trace);
Overlap_IO_Buffer *OVL_IO_Buffer(Align_Spec *spec);
int  AddOverlapToBuffer(Overlap_IO_Buffer *iobuf, Overlap *ovl, int tbytes);
void Write_Overlap_Buffer(Align_Spec *spec, char *dirName1, char *dirName2,
                          char *ablock, char *bblock, int lastRead);
void Reset_Overlap_Buffer(Align_Spec *spec);

/* align.c:3365-3396 */
int  Write_Overlap(FILE *output, Overlap *ovl, int tbytes);
int  Compress_TraceTo8(Overlap *ovl, int check);

/* Accessors the device shim needs (not in the reference header). */
const int16 *damar_spec_score_table(Align_Spec *spec);   /* SCORE[32768] */
const int16 *damar_spec_trim_table(Align_Spec *spec);    /* TABLE[32768] */
int          damar_spec_ave_path(Align_Spec *spec);

#ifdef __cplusplus
}
#endif
#endif

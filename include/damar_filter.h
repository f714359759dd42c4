/* damar_filter.h -- the three-function seed filter interface of the overlap path.
 *
 * Exactly the reference's dalign/filter.h:54-70: same symbol names, signatures,
 * ownership rules and error behaviour (fatal errors print to stderr and exit(1);
 * only Set_Filter_Params returns a status).  In libdamar_hip.so these run on the
 * MI355X: the k-mer index returned by Sort_Kmers is an opaque handle to a
 * device-resident sorted index, and Match_Filter performs merge, seed sort,
 * diagonal-band filter and the Local_Alignment waves in HIP kernels, then
 * appends the resulting Overlap records to the Align_Spec's Overlap_IO_Buffer.
 */
#ifndef DAMAR_FILTER_H
#define DAMAR_FILTER_H

#include "damar_db.h"
#include "damar_align.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Globals shared with the caller (filter.h:54-62, defined in daligner.c:131-140).
 * The library carries default definitions; an executable that defines them itself
 * (as the reference daligner.c does) pre-empts these through normal ELF symbol
 * interposition. */
extern int    BIASED;
extern int    VERBOSE;
extern int    MINOVER;
extern int    HGAP_MIN;
extern int    SYMMETRIC;
extern int    IDENTITY;
extern uint64 MEM_LIMIT;
extern uint64 MEM_PHYSICAL;

/* filter.c:171-201.  Returns 1 for an illegal k (kmer <= 1), else 0. */
int   Set_Filter_Params(int kmer, int binshift, int suppress, int hitmin, int nthreads);

/* filter.c:753-994.  Returns an opaque index (malloc'ed handle; HBM-resident
 * payload) and the k-mer count in *len; NULL / 0 if the block has no k-mers. */
void *Sort_Kmers(HITS_DB *block, int *len);

/* filter.c:2519-2929.  MG_self is decided by pointer equality aname == bname.
 * Takes ownership of btable when atable != btable (released before return). */
void  Match_Filter(char *aname, HITS_DB *ablock, char *bname, HITS_DB *bblock,
                   void *atable, int alen, void *btable, int blen,
                   int comp, Align_Spec *asettings);

#ifdef __cplusplus
}
#endif
#endif

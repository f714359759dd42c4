/* tandem_abi.c -> libdamar_tandem.so: the reference names of scrub/tandem.h:58-60 for a
 * scrub/datander.c that links against this library instead of scrub/tandem.c.  datander's
 * Set_Filter_Params has 4 arguments while daligner's (filter.h:64) has 5, so the two cannot live
 * in one library; Match_Self and everything from align.h come from libdamar_hip.so. */
int damar_tandem_set_params(int kmer, int binshift, int hitmin, int nthreads);

char *SORT_PATH = "/tmp";      /* scrub/tandem.h:56, only used by compiled-out THREAD_OUTPUT */

int Set_Filter_Params(int kmer, int binshift, int hitmin, int nthreads)
{ return damar_tandem_set_params(kmer, binshift, hitmin, nthreads);
}

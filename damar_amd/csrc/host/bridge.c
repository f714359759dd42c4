/* bridge.c -- bridging of two narrowly parallel local alignments of one read pair by an
 * exact O(nd) realignment of the gap (reference filter.c:1444-1571, 1747-1802 on top of
 * align.c:4327-4869 Compute_Alignment(DIFF_TRACE)).
 *
 * NOT BUILT YET (round 1): a pair that reaches this point stops the run loudly instead
 * of emitting records that could differ from the reference.  Plain simulator reads never
 * get here (0 Bridge calls); tandem-rich genomes do.
 */
#include <stdio.h>
#include <stdlib.h>
#include "damar_host.h"

int damar_bridge_pair(const damar_bridge_ctx *ctx, damar_path *jp, damar_path *kp,
                      damar_path *p1, damar_path *p2, damar_path *b1, damar_path *b2,
                      int aovl, int bovl, int comp, int ts, damar_tpool *tp,
                      damar_path *bm, int j)
{ (void) ctx; (void) jp; (void) kp; (void) b1; (void) b2; (void) comp; (void) ts; (void) tp; (void) bm; (void) j;
  fprintf(stderr, "damar: FATAL: read pair needs a Bridge realignment "
                  "([%d,%d]x[%d,%d] vs [%d,%d]x[%d,%d], overlap %d/%d) -- not implemented yet\n",
          p1->abpos, p1->aepos, p1->bbpos, p1->bepos, p2->abpos, p2->aepos, p2->bbpos, p2->bepos,
          aovl, bovl);
  exit(3);
  return 1;
}

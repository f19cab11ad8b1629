/* layout_probe.c -- TEST ORACLE support: prints sizeof/offsetof of the level_type
 * family, compiled once against the reference's level.h (-DPROBE_REFERENCE) and
 * once against include/hpgmg_level.h; the two outputs must be identical. */
#include <stdio.h>
#include <stddef.h>
#ifdef PROBE_REFERENCE
#include "level.h"
#else
#include "hpgmg_level.h"
#endif
#define S(t) printf("sizeof(" #t ")=%zu\n", sizeof(t))
#define O(t, f) printf("offsetof(" #t "," #f ")=%zu\n", offsetof(t, f))
int main(void) {
  S(blockCopy_type); O(blockCopy_type, dim); O(blockCopy_type, read); O(blockCopy_type, write); O(blockCopy_type, read.ptr); O(blockCopy_type, write.kStride);
  S(communicator_type); O(communicator_type, recv_ranks); O(communicator_type, send_buffers); O(communicator_type, allocated_blocks); O(communicator_type, num_blocks); O(communicator_type, blocks);
  S(box_type); O(box_type, low); O(box_type, dim); O(box_type, jStride); O(box_type, numVectors); O(box_type, vectors); O(box_type, fp_base);
  S(level_type); O(level_type, h); O(level_type, active); O(level_type, box_dim); O(level_type, box_jStride); O(level_type, numVectors); O(level_type, tag);
  O(level_type, boxes_in); O(level_type, dim); O(level_type, rank_of_box); O(level_type, my_boxes); O(level_type, my_blocks); O(level_type, boundary_condition);
  O(level_type, exchange_ghosts); O(level_type, restriction); O(level_type, interpolation); O(level_type, dominant_eigenvalue_of_DinvA);
  O(level_type, must_subtract_mean); O(level_type, RedBlack_FP); O(level_type, fluxes); O(level_type, num_threads); O(level_type, timers);
  O(level_type, timers.ghostZone_wait); O(level_type, timers.Total); O(level_type, Krylov_iterations); O(level_type, vcycles_from_this_level);
  return 0;
}

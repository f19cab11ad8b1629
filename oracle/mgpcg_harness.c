/*
 * mgpcg_harness.c -- TEST ORACLE support: our own main() around the REFERENCE's third driver of the operator plugin, MGPCG (mg.c:1500-1605:
 * conjugate gradients preconditioned with one V-cycle per iteration).  The reference's hpgmg-fv.c never calls it, so neither its F-cycle
 * build nor its -DUSE_VCYCLES build exercises what MGPCG asks of a plugin: create_vectors() growing every level by three vectors after
 * MGBuild, MGVCycle on vector ids beyond VECTORS_RESERVED (z = M^-1 r), apply_op / dot / add_vectors on the fine level in between.
 *
 * oracle/Makefile compiles this file with the reference's unmodified timers.c level.c operators.<OP>.c mg.c solvers.c (`mgpcg-*`: the
 * reference itself) and with the reference's mg.c / solvers.c on the product plugin (`routeb-*-mgpcg`, INTEGRATION.md Route B);
 * tests/test_gpu_route_b.py compares what the two print, line by line.  The set-up is hpgmg-fv.c:283-308.
 *
 *   mgpcg_harness <log2_box_dim> <target_boxes_per_rank>
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "defines.h"
#include "level.h"
#include "operators.h"
#include "mg.h"

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s log2_box_dim target_boxes_per_rank\n", argv[0]); return 2; }
  const int log2_box_dim = atoi(argv[1]), target_boxes = atoi(argv[2]);
  const int box_dim = 1 << log2_box_dim;
  int boxes_in_i = (int)(cbrt((double)target_boxes) + 0.5);
  if (boxes_in_i < 1) boxes_in_i = 1;
  level_type fine;
  create_level(&fine, boxes_in_i, box_dim, stencil_get_radius(), VECTORS_RESERVED, BC_DIRICHLET, 0, 1);
#ifdef USE_HELMHOLTZ
  const double a = 1.0, b = 1.0;
#else
  const double a = 0.0, b = 1.0;
#endif
  const double h = 1.0 / ((double)boxes_in_i * (double)box_dim);
  initialize_problem(&fine, h, a, b);
  rebuild_operator(&fine, NULL, a, b);
  mg_type MG;
  MGBuild(&MG, &fine, a, b, 1);
  int solve;
  for (solve = 0; solve < 2; solve++) {                       /* twice: the second call finds the three extra vectors in place */
    MGPCG(&MG, 0, VECTOR_U, VECTOR_F, a, b, 1e-10);
    fprintf(stdout, "MGPCG solve %d: norm(u)=%1.15e  Krylov iterations on the fine level so far=%d\n", solve, norm(&fine, VECTOR_U), fine.Krylov_iterations);
  }
  fprintf(stdout, "MGPCG dot(u,f)=%1.15e  mean(u)=%1.15e\n", dot(&fine, VECTOR_U, VECTOR_F), mean(&fine, VECTOR_U));
  MGDestroy(&MG);
  destroy_level(&fine);
  return 0;
}

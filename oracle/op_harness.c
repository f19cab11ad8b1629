/*
 * op_harness.c -- TEST ORACLE support (SURVEY.md 8(c) "single-operator harness"): our own main() for the REFERENCE's operator layer.
 *
 * Compiled by oracle/Makefile together with the reference's unmodified timers.c level.c operators.<OP>.c mg.c solvers.c (everything but
 * hpgmg-fv.c, the same -D flags), where they lie under /root/reference.  It builds the 16^3 test problem the reference's main() builds
 * (hpgmg-fv.c:283-308: create_level, initialize_problem, rebuild_operator, MGBuild), then calls the operators of operators.h ONE BY ONE
 * in a fixed script and writes every vector they leave -- whole padded boxes: interior, ghost zones, row padding -- and every scalar
 * they return to a file.  tests/golden/make_ops_golden.py turns those files into the committed fixtures tests/golden/ops_golden.json;
 * tests/ops_script.py replays the same script on the CPU restatement and on the HIP plugin.  This pins the oracle PER OPERATOR and
 * per ghost cell, not only through the norms of whole F-cycles.
 *
 *   op_harness <boxes_in_i> <box_dim> <out file>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "defines.h"
#include "level.h"
#include "operators.h"
#include "mg.h"

void apply_BCs(level_type *level, int x_id, int shape);      /* the plugin's dispatch (operators.7pt.c:47, operators.fv4.c:51 ...) */

static FILE *out;
static void dump(const char *name, level_type *L, int lev, int id) {
  int b;
  fprintf(out, "DUMP %s %d %d %d %d\n", name, lev, id, L->num_my_boxes, L->box_volume);
  for (b = 0; b < L->num_my_boxes; b++) fwrite(L->my_boxes[b].vectors[id], sizeof(double), (size_t)L->box_volume, out);
  fprintf(out, "\n");
}
static void scalar(const char *name, double v) { fprintf(out, "SCALAR %s %.17g\n", name, v); }
static void geom(level_type *L, int lev) {
  fprintf(out, "GEOM %d %d %d %d %d %d %d %d\n", lev, L->dim.i, L->box_dim, L->box_ghosts, L->box_jStride, L->box_kStride, L->box_volume, L->num_my_boxes);
}

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s boxes_in_i box_dim outfile\n", argv[0]); return 2; }
  const int boxes_in_i = atoi(argv[1]), box_dim = atoi(argv[2]);
  out = fopen(argv[3], "wb");
  if (!out) { perror(argv[3]); return 2; }
#ifdef USE_HELMHOLTZ
  const double a = 1.0, b = 1.0;
#else
  const double a = 0.0, b = 1.0;
#endif
  level_type fine;
  create_level(&fine, boxes_in_i, box_dim, stencil_get_radius(), VECTORS_RESERVED, BC_DIRICHLET, 0, 1);
  const double h = 1.0 / ((double)boxes_in_i * (double)box_dim);
  initialize_problem(&fine, h, a, b);
  rebuild_operator(&fine, NULL, a, b);
  mg_type MG;
  MGBuild(&MG, &fine, a, b, 1);
  level_type *L0 = MG.levels[0], *L1 = MG.levels[1];
  const int shape = stencil_get_shape();
  fprintf(out, "CONFIG %d %d %g %g %d %d %d\n", boxes_in_i, box_dim, a, b, stencil_get_radius(), shape, VECTORS_RESERVED);
  geom(L0, 0); geom(L1, 1);

  /* setup: what initialize_problem and rebuild_operator left */
  dump("setup.beta_i", L0, 0, VECTOR_BETA_I); dump("setup.beta_j", L0, 0, VECTOR_BETA_J); dump("setup.beta_k", L0, 0, VECTOR_BETA_K);
#ifdef USE_HELMHOLTZ
  dump("setup.alpha", L0, 0, VECTOR_ALPHA);
#endif
  dump("setup.f", L0, 0, VECTOR_F); dump("setup.dinv", L0, 0, VECTOR_DINV);
  scalar("setup.eig0", L0->dominant_eigenvalue_of_DinvA);
  dump("setup.dinv1", L1, 1, VECTOR_DINV); dump("setup.beta_i1", L1, 1, VECTOR_BETA_I);
  scalar("setup.eig1", L1->dominant_eigenvalue_of_DinvA);

  /* SURVEY.md 8(c) known answers: one smooth() from U = 0, then the residual */
  zero_vector(L0, VECTOR_U);
  smooth(L0, VECTOR_U, VECTOR_F, a, b);
  dump("first.smooth.u", L0, 0, VECTOR_U); dump("first.smooth.temp", L0, 0, VECTOR_TEMP);
  scalar("first.norm_u", norm(L0, VECTOR_U));
  residual(L0, VECTOR_TEMP, VECTOR_U, VECTOR_F, a, b);
  dump("first.residual", L0, 0, VECTOR_TEMP);
  scalar("first.norm_res", norm(L0, VECTOR_TEMP));

  /* a rough field: the +-1 parity pattern of random_vector() (misc.c:478-505) on top of a multiple of F */
  random_vector(L0, VECTOR_U);
  scale_vector(L0, VECTOR_U, 0.001, VECTOR_U);
  add_vectors(L0, VECTOR_U, 1.0, VECTOR_U, 0.0001, VECTOR_F);
  dump("field.u", L0, 0, VECTOR_U);
  exchange_boundary(L0, VECTOR_U, shape);
  dump("exchange.u", L0, 0, VECTOR_U);
  apply_BCs(L0, VECTOR_U, shape);
  dump("bcs.u", L0, 0, VECTOR_U);
  exchange_boundary(L0, VECTOR_U, STENCIL_SHAPE_BOX);
  apply_BCs(L0, VECTOR_U, STENCIL_SHAPE_BOX);
  dump("bcs_box.u", L0, 0, VECTOR_U);
  smooth(L0, VECTOR_U, VECTOR_F, a, b);
  dump("smooth.u", L0, 0, VECTOR_U); dump("smooth.temp", L0, 0, VECTOR_TEMP);
  residual(L0, VECTOR_R, VECTOR_U, VECTOR_F, a, b);
  dump("residual.r", L0, 0, VECTOR_R); dump("residual.u", L0, 0, VECTOR_U);
  apply_op(L0, VECTOR_E, VECTOR_U, a, b);
  dump("apply_op.e", L0, 0, VECTOR_E);
  scalar("norm_r", norm(L0, VECTOR_R));
  scalar("dot_u_f", dot(L0, VECTOR_U, VECTOR_F));
  scalar("mean_u", mean(L0, VECTOR_U));

  /* restriction: cells, and the three face types on the coefficient vectors (the only vectors with values on the upper faces) */
  restriction(L1, VECTOR_R, L0, VECTOR_R, RESTRICT_CELL);
  dump("restrict.cell", L1, 1, VECTOR_R);
  restriction(L1, VECTOR_E, L0, VECTOR_BETA_I, RESTRICT_FACE_I);
  dump("restrict.face_i", L1, 1, VECTOR_E);
  restriction(L1, VECTOR_U, L0, VECTOR_BETA_J, RESTRICT_FACE_J);
  dump("restrict.face_j", L1, 1, VECTOR_U);
  restriction(L1, VECTOR_TEMP, L0, VECTOR_BETA_K, RESTRICT_FACE_K);
  dump("restrict.face_k", L1, 1, VECTOR_TEMP);

  /* interpolation of the restricted residual: V-cycle form added to U, F-cycle form into a zeroed E */
  interpolation_vcycle(L0, VECTOR_U, 1.0, L1, VECTOR_R);
  dump("interp_v.u", L0, 0, VECTOR_U); dump("interp_v.coarse", L1, 1, VECTOR_R);
  zero_vector(L0, VECTOR_E);
  interpolation_fcycle(L0, VECTOR_E, 0.0, L1, VECTOR_R);
  dump("interp_f.e", L0, 0, VECTOR_E); dump("interp_f.coarse", L1, 1, VECTOR_R);

  /* the remaining BLAS-1 of misc.c */
  mul_vectors(L0, VECTOR_TEMP, 2.0, VECTOR_U, VECTOR_F);
  invert_vector(L0, VECTOR_E, 1.0, VECTOR_DINV);
  shift_vector(L0, VECTOR_R, VECTOR_R, 0.5);
  dump("blas.mul", L0, 0, VECTOR_TEMP); dump("blas.invert", L0, 0, VECTOR_E); dump("blas.shift", L0, 0, VECTOR_R);
  scalar("error_u_e", error(L0, VECTOR_U, VECTOR_E));

  /* a second smooth() on the coarse level (other box size, agglomerated lists behind it) */
  zero_vector(L1, VECTOR_U);
  smooth(L1, VECTOR_U, VECTOR_R, a, b);
  dump("coarse.smooth.u", L1, 1, VECTOR_U);
  fprintf(out, "END\n");
  fclose(out);
  return 0;
}

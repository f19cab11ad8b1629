/*
 * config.c -- runtime form of the reference's compile-time plugin selection.
 *
 * The reference picks the stencil by compiling ONE of operators.{7pt,27pt,fv2,
 * fv4}.c and the smoother with -DUSE_CHEBY / -DUSE_GSRB / -DUSE_JACOBI
 * (finite-volume/source/README:66-96, local.mk:4).  Radius / shape / sweep
 * counts below are the constants those files hard-code:
 *   operators.7pt.c:92-93,256-264   operators.27pt.c:95-96,123-133
 *   operators.fv4.c:137-138,176-186 operators.fv2.c (same stencil as 7pt, FV BCs)
 */
#include "hpgmg_operators.h"

static hpgmg_config the_cfg = { HPGMG_OP_7PT, HPGMG_SMOOTH_CHEBY, 0, 1 };

int hpgmg_configure(const hpgmg_config *cfg) {
  if (cfg->op < HPGMG_OP_7PT || cfg->op > HPGMG_OP_FV2) return -1;
  if (cfg->smoother < HPGMG_SMOOTH_CHEBY || cfg->smoother > HPGMG_SMOOTH_JACOBI) return -1;
  if (cfg->op == HPGMG_OP_27PT && cfg->variable_coeff) return -1; /* operators.27pt.c:53-55 #error */
  the_cfg = *cfg;
  the_cfg.helmholtz = cfg->helmholtz ? 1 : 0;
  the_cfg.variable_coeff = cfg->variable_coeff ? 1 : 0;
  return 0;
}
void hpgmg_get_config(hpgmg_config *cfg) { *cfg = the_cfg; }
int hpgmg_vectors_reserved(void) { return the_cfg.helmholtz ? 11 : 9; }

int stencil_get_radius(void) { return the_cfg.op == HPGMG_OP_FV4 ? 2 : 1; }
int stencil_get_shape(void) {
  switch (the_cfg.op) {
    case HPGMG_OP_27PT: return STENCIL_SHAPE_BOX;
    case HPGMG_OP_FV4:  return the_cfg.variable_coeff ? STENCIL_SHAPE_NO_CORNERS : STENCIL_SHAPE_STAR; /* operators.fv4.c:137-143 */
    default:            return STENCIL_SHAPE_STAR;
  }
}

/* sweeps per smooth() call: Chebyshev degree x NUM_SMOOTHS, 2 x NUM_SMOOTHS colour
 * sweeps for GSRB, NUM_SMOOTHS for Jacobi */
static int sweeps_override = 0;
void hpgmg_set_smooth_sweeps(int n) { sweeps_override = n > 0 ? n : 0; }   /* experiments / tests only: the reference fixes the count at compile time */
int hpgmg_smooth_sweeps(void) {
  if (sweeps_override) return sweeps_override;
  const int fv = (the_cfg.op == HPGMG_OP_FV4 || the_cfg.op == HPGMG_OP_FV2);
  switch (the_cfg.smoother) {
    case HPGMG_SMOOTH_CHEBY:  return fv ? 6 : 4;  /* CHEBYSHEV_DEGREE 6 (fv2/fv4) or 4, NUM_SMOOTHS 1 */
    case HPGMG_SMOOTH_GSRB:   return fv ? 6 : 4;  /* NUM_SMOOTHS 3 (fv2/fv4) or 2, two colours each */
    default:                  return 6;           /* Jacobi: NUM_SMOOTHS 6 in every plugin */
  }
}
/* GSRB must read the previous iterate when the stencil couples same-colour cells
 * (operators.27pt.c:126, operators.fv4.c:178: #define GSRB_OOP) */
int hpgmg_gsrb_out_of_place(void) { return the_cfg.op == HPGMG_OP_27PT || the_cfg.op == HPGMG_OP_FV4; }

/* the host-driven bottom solver: the reference picks it with -DUSE_BICGSTAB (its default here) or -DUSE_CG (solvers.c:17-24) */
static int bottom_solver = HPGMG_BOTTOM_BICGSTAB;
void hpgmg_set_bottom_solver(int which) { bottom_solver = (which == HPGMG_BOTTOM_CG) ? HPGMG_BOTTOM_CG : HPGMG_BOTTOM_BICGSTAB; }   /* before MGBuild: it sets the number of work vectors */
int hpgmg_get_bottom_solver(void) { return bottom_solver; }

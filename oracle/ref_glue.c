/*
 * ref_glue.c -- TEST ORACLE support.  Lets oracle/operators_cpu.c be linked
 * under the REFERENCE's own driver (level.c mg.c solvers.c hpgmg-fv.c compiled
 * from /root/reference by oracle/Makefile) in place of operators.7pt.c.  The
 * reference fixes its configuration with -D flags; the same flags select the
 * oracle's runtime configuration here, before main() runs.
 */
#include <stdlib.h>
#include "hpgmg_level.h"
#include "hpgmg_operators.h"
#include "hpgmg_mg.h"

int hpgmg_verbose = 1;
hpgmg_solve_record hpgmg_last_solve;
const hpgmg_transport *hpgmg_get_transport(void) { return NULL; }
hpgmg_level_ext *hpgmg_level_ext_get(level_type *level) {
  static hpgmg_level_ext one; static int self = 0;
  one.level = level; one.active_ranks = &self; one.num_active_ranks = 1;
  return &one;
}

#ifndef GLUE_OP
#define GLUE_OP HPGMG_OP_7PT
#endif
#ifndef GLUE_VC
#define GLUE_VC 1
#endif
__attribute__((constructor)) static void glue_configure(void) {
  hpgmg_config c;
  c.op = GLUE_OP;
#if defined(USE_GSRB)
  c.smoother = HPGMG_SMOOTH_GSRB;
#elif defined(USE_JACOBI)
  c.smoother = HPGMG_SMOOTH_JACOBI;
#else
  c.smoother = HPGMG_SMOOTH_CHEBY;
#endif
#ifdef USE_HELMHOLTZ
  c.helmholtz = 1;
#else
  c.helmholtz = 0;
#endif
  c.variable_coeff = GLUE_VC;
  if (hpgmg_configure(&c)) abort();
}

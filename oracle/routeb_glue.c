/*
 * routeb_glue.c -- TEST support for INTEGRATION.md "Route B": the REFERENCE's own driver (timers.c, level.c with the
 * three storage lines patched by oracle/Makefile, mg.c, solvers.c, hpgmg-fv.c, compiled from /root/reference)
 * linked against the PRODUCT plugin (hpgmg_amd/csrc/host/operators_hip.c + config.c + libhpgmg_hip.so).  Supplies the
 * handful of symbols the plugin expects from its host layer and turns the reference's -D configuration into the
 * plugin's runtime configuration before main() runs.  The resulting binary is a parity check of the drop-in claim
 * (tests/test_gpu_route_b.py); it is not part of the product.
 */
#include <stdlib.h>
#include <string.h>
#include "hpgmg_level.h"
#include "hpgmg_operators.h"
#include "hpgmg_mg.h"

int hpgmg_verbose = 1;
int hpgmg_gather_dim = 0;                                   /* single rank: irrelevant */
static const hpgmg_transport *transport = NULL;
const hpgmg_transport *hpgmg_get_transport(void) { return transport; }
void hpgmg_set_transport(const hpgmg_transport *t) { transport = t; }

/* one side record per level, keyed by the level's address (the reference's level_type has no room for it) */
#define MAX_EXT 64
static hpgmg_level_ext table[MAX_EXT];
static int used = 0, self_rank = 0;
hpgmg_level_ext *hpgmg_level_ext_get(level_type *level) {
  int n;
  for (n = 0; n < used; n++) if (table[n].level == level) return &table[n];
  if (used == MAX_EXT) abort();
  memset(&table[used], 0, sizeof(table[used]));
  table[used].level = level; table[used].active_ranks = &self_rank; table[used].num_active_ranks = 1;
  return &table[used++];
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

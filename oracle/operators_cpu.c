/*
 * operators_cpu.c -- TEST ORACLE.  NOT PRODUCT CODE.
 *
 * A plain C (+OpenMP over tiles) restatement of the HPGMG-FV operator plugin
 * for host memory.  It implements include/hpgmg_operators.h and exists only so
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have
 * something to check the HIP path against; nothing under hpgmg_amd/ links,
 * loads or calls it.
 *
 * Pinning: PINNED.  oracle/Makefile builds (a) the reference itself, unmodified,
 * from /root/reference into oracle/_ref/ and (b) the reference's OWN driver
 * (level.c mg.c solvers.c hpgmg-fv.c) linked against THIS file in place of
 * operators.7pt.c; tests/test_oracle_vs_reference.py requires (a) and (b) to
 * print identical 15-digit norms, and tests/golden/ holds those norms so the
 * check still runs where /root/reference does not exist.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * finite-volume/source/).  Floating-point expressions keep the reference's
 * association; compile with -ffp-contract=off.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include "hpgmg_level.h"
#include "hpgmg_operators.h"
#include "hpgmg_mg.h"

int hpgmg_smooth_sweeps(void);
int hpgmg_gsrb_out_of_place(void);

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------------------------------------------------------- storage hooks */
const char *hpgmg_backend_name(void) { return "oracle-cpu"; }
double *hpgmg_vector_alloc(size_t n) {
  void *p = NULL;
  if (posix_memalign(&p, 4096, (n ? n : 1) * sizeof(double))) { fprintf(stderr, "oracle: out of memory\n"); exit(1); }
  memset(p, 0, (n ? n : 1) * sizeof(double));
  return (double *)p;
}
void hpgmg_vector_free(double *p) { free(p); }
void hpgmg_vector_copy(double *d, const double *s, size_t n) { memcpy(d, s, n * sizeof(double)); }
void hpgmg_vector_upload(double *d, const double *s, size_t n) { memcpy(d, s, n * sizeof(double)); }
void hpgmg_vector_download(double *d, const double *s, size_t n) { memcpy(d, s, n * sizeof(double)); }
void hpgmg_level_release(level_type *level) { (void)level; }
/* timing hooks: the oracle is synchronous, so a tick is the host clock */
hpgmg_tick hpgmg_tick_begin(level_type *L, double *acc, const char *what) { hpgmg_tick t; (void)L; (void)what; t.t0 = now(); t.acc = acc; t.slot = -1; t.range = 0; return t; }
void hpgmg_tick_end(hpgmg_tick t) { if (t.acc) *t.acc += now() - t.t0; }
void hpgmg_timers_settle(void) {}
void hpgmg_set_timer_mode(int mode) { (void)mode; }
int hpgmg_get_timer_mode(void) { return 0; }
void hpgmg_level_sync_counters(level_type *level) { (void)level; }
int hpgmg_restrict_zero_fused(level_type *c, int ic, level_type *f, int i_f, int z) { (void)c; (void)ic; (void)f; (void)i_f; (void)z; return 0; }
int hpgmg_residual_restrict_zero_fused(level_type *c, int ic, level_type *f, int x, int r, double a, double b, int z) { (void)c; (void)ic; (void)f; (void)x; (void)r; (void)a; (void)b; (void)z; return 0; }
void hpgmg_operators_flush(void) { }
int  hpgmg_norm_scale_restrict_fused_deferred(level_type *level, int F_id, int R_id, level_type *coarse) { (void)level; (void)F_id; (void)R_id; (void)coarse; return 0; }
double hpgmg_norm_deferred_fetch(level_type *level) { (void)level; return 0.0; }
int  hpgmg_zero_interpolation_fcycle_fused(level_type *fine, int id_f, level_type *coarse, int id_c) { (void)fine; (void)id_f; (void)coarse; (void)id_c; return 0; }
void hpgmg_set_lazy(int on) { (void)on; }
int hpgmg_smooth_in_cycle(level_type *l, int p, int r, double a, double b) { (void)l; (void)p; (void)r; (void)a; (void)b; return 0; }
int hpgmg_norm_scale_restrict_fused(level_type *l, int f, int r, level_type *c, double *o) { (void)l; (void)f; (void)r; (void)c; (void)o; return 0; }
int hpgmg_residual_norm_fused(level_type *l, int res, int x, int r, double a, double b, double *o) { (void)l; (void)res; (void)x; (void)r; (void)a; (void)b; (void)o; return 0; }
int hpgmg_interp_smooth_fused(level_type *f, int e, int R, level_type *c, double a, double b) { (void)f; (void)e; (void)R; (void)c; (void)a; (void)b; return 0; }
void hpgmg_set_graphs(int on) { (void)on; }
void hpgmg_set_smoother_precision(int bits) { (void)bits; }   /* the oracle is fp64 only */
int hpgmg_get_smoother_precision(void) { return 64; }
void hpgmg_segment_begin(long long key) { (void)key; }
void hpgmg_segment_end(void) {}
void hpgmg_solve_attempt_begin(void) {}      /* include/hpgmg_operators.h: no launch of this plugin can fail as a whole */
int hpgmg_solve_attempt_end(void) { return 0; }
int hpgmg_vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg) {
  (void)levels; (void)n; (void)e_id; (void)R_id; (void)a; (void)b; (void)leg; return 0; }
int hpgmg_bottom_solve_fused(level_type *L, int e_id, int R_id, double a, double b, double want) {
  (void)L; (void)e_id; (void)R_id; (void)a; (void)b; (void)want; return 0; }

/* interior-origin pointer of vector id in box b */
static inline double *vec(const level_type *L, int box, int id) {
  const box_type *B = &L->my_boxes[box];
  return B->vectors[id] + (size_t)B->ghosts * (size_t)(1 + B->jStride + B->kStride);
}

/* ---------------------------------------------------------------- stencils
 * operators.7pt.c:49-89.  C evaluates a*alpha*x - b*h2inv*(sum) as
 * ((a*alpha)*x) - ((b*h2inv)*sum), the sum strictly left to right. */
#define SUM6_BETA(x) ( + beta_i[ijk+1 ]*( x[ijk+1 ] - x[ijk] ) + beta_i[ijk]*( x[ijk-1 ] - x[ijk] ) \
                       + beta_j[ijk+jS]*( x[ijk+jS] - x[ijk] ) + beta_j[ijk]*( x[ijk-jS] - x[ijk] ) \
                       + beta_k[ijk+kS]*( x[ijk+kS] - x[ijk] ) + beta_k[ijk]*( x[ijk-kS] - x[ijk] ) )

#define SUFFIX _7pt_vc_helmholtz
#define USES_ALPHA 1
#define USES_BETA 1
#define APPLY_OP(x) ( a*alpha[ijk]*x[ijk] - b*h2inv*SUM6_BETA(x) )
#include "stencil_sweeps.inc"

#define SUFFIX _7pt_vc_poisson
#define USES_ALPHA 0
#define USES_BETA 1
#define APPLY_OP(x) ( -b*h2inv*SUM6_BETA(x) )
#include "stencil_sweeps.inc"

#define SUFFIX _7pt_cc
#define USES_ALPHA 0
#define USES_BETA 0
#define APPLY_OP(x) ( a*x[ijk] - b*h2inv*( + x[ijk+1] + x[ijk-1] + x[ijk+jS] + x[ijk-jS] + x[ijk+kS] + x[ijk-kS] - x[ijk]*6.0 ) )
#include "stencil_sweeps.inc"

/* operators.27pt.c:48-51,60-91: decimal literals, three weighted partial sums then the centre */
#define C27_0 (-4.2666666666666666666)
#define C27_1 ( 0.4666666666666666666)
#define C27_2 ( 0.1000000000000000000)
#define C27_3 ( 0.0333333333333333333)
#define SUFFIX _27pt_cc
#define USES_ALPHA 0
#define USES_BETA 0
#define APPLY_OP(x) ( a*x[ijk] - b*h2inv*( \
    C27_3*( x[ijk-kS-jS-1] + x[ijk-kS-jS+1] + x[ijk-kS+jS-1] + x[ijk-kS+jS+1] + x[ijk+kS-jS-1] + x[ijk+kS-jS+1] + x[ijk+kS+jS-1] + x[ijk+kS+jS+1] ) + \
    C27_2*( x[ijk-kS-jS] + x[ijk-kS-1] + x[ijk-kS+1] + x[ijk-kS+jS] + x[ijk-jS-1] + x[ijk-jS+1] + x[ijk+jS-1] + x[ijk+jS+1] + x[ijk+kS-jS] + x[ijk+kS-1] + x[ijk+kS+1] + x[ijk+kS+jS] ) + \
    C27_1*( x[ijk-kS] + x[ijk-jS] + x[ijk-1] + x[ijk+1] + x[ijk+jS] + x[ijk+kS] ) + \
    C27_0*( x[ijk] ) ) )
#include "stencil_sweeps.inc"

/* operators.fv4.c:53-114: 4th-order finite-volume operator: six face fluxes with a 15/-1 two-point
 * correction, plus twelve mixed terms (transverse beta difference) x (4-point x difference) */
#define FV4_TWELFTH ( 0.0833333333333333333)
#define FV4_SUM(x) ( \
      FV4_TWELFTH*( \
        + beta_i[ijk   ]*( 15.0*(x[ijk-1 ]-x[ijk]) - (x[ijk-2   ]-x[ijk+1 ]) ) \
        + beta_i[ijk+1 ]*( 15.0*(x[ijk+1 ]-x[ijk]) - (x[ijk+2   ]-x[ijk-1 ]) ) \
        + beta_j[ijk   ]*( 15.0*(x[ijk-jS]-x[ijk]) - (x[ijk-2*jS]-x[ijk+jS]) ) \
        + beta_j[ijk+jS]*( 15.0*(x[ijk+jS]-x[ijk]) - (x[ijk+2*jS]-x[ijk-jS]) ) \
        + beta_k[ijk   ]*( 15.0*(x[ijk-kS]-x[ijk]) - (x[ijk-2*kS]-x[ijk+kS]) ) \
        + beta_k[ijk+kS]*( 15.0*(x[ijk+kS]-x[ijk]) - (x[ijk+2*kS]-x[ijk-kS]) ) \
      ) \
      + 0.25*FV4_TWELFTH*( \
        + (beta_i[ijk   +jS]-beta_i[ijk   -jS]) * (x[ijk-1 +jS]-x[ijk+jS]-x[ijk-1 -jS]+x[ijk-jS]) \
        + (beta_i[ijk   +kS]-beta_i[ijk   -kS]) * (x[ijk-1 +kS]-x[ijk+kS]-x[ijk-1 -kS]+x[ijk-kS]) \
        + (beta_j[ijk   +1 ]-beta_j[ijk   -1 ]) * (x[ijk-jS+1 ]-x[ijk+1 ]-x[ijk-jS-1 ]+x[ijk-1 ]) \
        + (beta_j[ijk   +kS]-beta_j[ijk   -kS]) * (x[ijk-jS+kS]-x[ijk+kS]-x[ijk-jS-kS]+x[ijk-kS]) \
        + (beta_k[ijk   +1 ]-beta_k[ijk   -1 ]) * (x[ijk-kS+1 ]-x[ijk+1 ]-x[ijk-kS-1 ]+x[ijk-1 ]) \
        + (beta_k[ijk   +jS]-beta_k[ijk   -jS]) * (x[ijk-kS+jS]-x[ijk+jS]-x[ijk-kS-jS]+x[ijk-jS]) \
        + (beta_i[ijk+1 +jS]-beta_i[ijk+1 -jS]) * (x[ijk+1 +jS]-x[ijk+jS]-x[ijk+1 -jS]+x[ijk-jS]) \
        + (beta_i[ijk+1 +kS]-beta_i[ijk+1 -kS]) * (x[ijk+1 +kS]-x[ijk+kS]-x[ijk+1 -kS]+x[ijk-kS]) \
        + (beta_j[ijk+jS+1 ]-beta_j[ijk+jS-1 ]) * (x[ijk+jS+1 ]-x[ijk+1 ]-x[ijk+jS-1 ]+x[ijk-1 ]) \
        + (beta_j[ijk+jS+kS]-beta_j[ijk+jS-kS]) * (x[ijk+jS+kS]-x[ijk+kS]-x[ijk+jS-kS]+x[ijk-kS]) \
        + (beta_k[ijk+kS+1 ]-beta_k[ijk+kS-1 ]) * (x[ijk+kS+1 ]-x[ijk+1 ]-x[ijk+kS-1 ]+x[ijk-1 ]) \
        + (beta_k[ijk+kS+jS]-beta_k[ijk+kS-jS]) * (x[ijk+kS+jS]-x[ijk+jS]-x[ijk+kS-jS]+x[ijk-jS]) \
      ) )
#define SUFFIX _fv4_vc_helmholtz
#define USES_ALPHA 1
#define USES_BETA 1
#define APPLY_OP(x) ( a*alpha[ijk]*x[ijk] - b*h2inv*FV4_SUM(x) )
#include "stencil_sweeps.inc"
#define SUFFIX _fv4_vc_poisson
#define USES_ALPHA 0
#define USES_BETA 1
#define APPLY_OP(x) ( -b*h2inv*FV4_SUM(x) )
#include "stencil_sweeps.inc"

enum { K_7PT_VC_HELM = 0, K_7PT_VC_POIS, K_7PT_CC, K_27PT_CC, K_FV4_VC_HELM, K_FV4_VC_POIS, K_UNSUPPORTED };
static int kernel_variant(void) {
  hpgmg_config c;
  hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_7PT || c.op == HPGMG_OP_FV2) {   /* operators.fv2.c uses the 7-pt stencil with finite-volume BCs */
    if (!c.variable_coeff) return K_7PT_CC;
    return c.helmholtz ? K_7PT_VC_HELM : K_7PT_VC_POIS;
  }
  if (c.op == HPGMG_OP_27PT) return K_27PT_CC;
  if (c.op == HPGMG_OP_FV4 && c.variable_coeff) return c.helmholtz ? K_FV4_VC_HELM : K_FV4_VC_POIS;
  fprintf(stderr, "oracle: operator %d not restated yet\n", c.op);
  exit(1);
}
#define DISPATCH(fn, ...) do { switch (kernel_variant()) { \
    case K_7PT_VC_HELM: fn##_7pt_vc_helmholtz(__VA_ARGS__); break; \
    case K_7PT_VC_POIS: fn##_7pt_vc_poisson(__VA_ARGS__); break; \
    case K_27PT_CC:     fn##_27pt_cc(__VA_ARGS__); break; \
    case K_FV4_VC_HELM: fn##_fv4_vc_helmholtz(__VA_ARGS__); break; \
    case K_FV4_VC_POIS: fn##_fv4_vc_poisson(__VA_ARGS__); break; \
    default:            fn##_7pt_cc(__VA_ARGS__); break; } } while (0)

/* ---------------------------------------------------------------- block copies
 * operators/blockCopy.c:6-105 (copy) and :109-156 (increment). */
static void resolve(const level_type *L, int id, const blockCopy_type *blk, int is_write, double **p, int *jS, int *kS) {
  int box = is_write ? blk->write.box : blk->read.box;
  if (box >= 0) {
    *jS = L->my_boxes[box].jStride; *kS = L->my_boxes[box].kStride;
    *p = vec(L, box, id);
  } else {
    *jS = is_write ? blk->write.jStride : blk->read.jStride;
    *kS = is_write ? blk->write.kStride : blk->read.kStride;
    *p = is_write ? blk->write.ptr : blk->read.ptr;
  }
  if (is_write) *p += blk->write.i + blk->write.j * (*jS) + blk->write.k * (*kS);
  else          *p += blk->read.i  + blk->read.j  * (*jS) + blk->read.k  * (*kS);
}
static void copy_block(level_type *L, int id, const blockCopy_type *blk) {
  double *r, *w; int rj, rk, wj, wk, i, j, k;
  resolve(L, id, blk, 0, &r, &rj, &rk);
  resolve(L, id, blk, 1, &w, &wj, &wk);
  for (k = 0; k < blk->dim.k; k++) for (j = 0; j < blk->dim.j; j++) for (i = 0; i < blk->dim.i; i++)
    w[i + j * wj + k * wk] = r[i + j * rj + k * rk];
}
static void increment_block(level_type *L, int id, double prescale, const blockCopy_type *blk) {
  double *r, *w; int rj, rk, wj, wk, i, j, k;
  resolve(L, id, blk, 0, &r, &rj, &rk);
  resolve(L, id, blk, 1, &w, &wj, &wk);
  for (k = 0; k < blk->dim.k; k++) for (j = 0; j < blk->dim.j; j++) for (i = 0; i < blk->dim.i; i++)
    w[i + j * wj + k * wk] = prescale * w[i + j * wj + k * wk] + r[i + j * rj + k * rk];
}

static void transport_phase(const communicator_type *recv_side, const communicator_type *send_side, int tag) {
  const hpgmg_transport *T = hpgmg_get_transport();
  int nr = recv_side ? recv_side->num_recvs : 0, ns = send_side ? send_side->num_sends : 0;
  if (nr + ns == 0) return;
  if (!T) { fprintf(stderr, "oracle: level needs %d messages but no transport is set\n", nr + ns); exit(1); }
  T->sendrecv(T->ctx, nr, nr ? recv_side->recv_buffers : NULL, nr ? recv_side->recv_sizes : NULL, nr ? recv_side->recv_ranks : NULL,
              ns, ns ? send_side->send_buffers : NULL, ns ? send_side->send_sizes : NULL, ns ? send_side->send_ranks : NULL, tag);
}

/* ---------------------------------------------------------------- ghost exchange
 * operators/exchange_boundary.c:12-117: pack, (send/recv), local copies, unpack. */
void exchange_boundary(level_type *L, int id, int shape) {
  const double t0 = now();
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  communicator_type *C = &L->exchange_ghosts[shape];
  int n;
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < C->num_blocks[0]; n++) copy_block(L, id, &C->blocks[0][n]);
  transport_phase(C, C, (L->tag << 4) | shape);
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < C->num_blocks[1]; n++) copy_block(L, id, &C->blocks[1][n]);
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < C->num_blocks[2]; n++) copy_block(L, id, &C->blocks[2][n]);
  L->timers.ghostZone_total += now() - t0;
}

/* ---------------------------------------------------------------- boundary conditions
 * operators/boundary_fd.c:6-90: homogeneous Dirichlet by linear extrapolation
 * through the face: ghost = -x(mirror) on faces, +x on edges, -x on corners,
 * the mirror cell being one step along the inward DOMAIN normal. */
void apply_BCs_p1(level_type *L, int x_id, int shape) {
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  const double t0 = now();
  int n;
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < L->boundary_condition.num_blocks[shape]; n++) {
    const blockCopy_type *blk = &L->boundary_condition.blocks[shape][n];
    const int inward = 26 - blk->subtype;
    const int di = inward % 3 - 1, dj = (inward % 9) / 3 - 1, dk = inward / 9 - 1;
    const int kind = (di != 0) + (dj != 0) + (dk != 0);
    const double scale = (kind == 2) ? 1.0 : -1.0;
    const box_type *B = &L->my_boxes[blk->read.box];
    const int jS = B->jStride, kS = B->kStride, step = di + dj * jS + dk * kS;
    double *x = vec(L, blk->read.box, x_id);
    int i, j, k;
    for (k = 0; k < blk->dim.k; k++) for (j = 0; j < blk->dim.j; j++) for (i = 0; i < blk->dim.i; i++) {
      const int ijk = (i + blk->read.i) + (j + blk->read.j) * jS + (k + blk->read.k) * kS;
      x[ijk] = scale * x[ijk + step];
    }
  }
  L->timers.boundary_conditions += now() - t0;
}
/* operators/boundary_fd.c:93-205: quadratic extrapolation through a zero on the boundary.  With s the
 * inward step(s): face  -2 x(s) + 1/3 x(2s);  edge  4 x(r+s) - 2/3 x(2r+s) - 2/3 x(r+2s) + 1/9 x(2r+2s);
 * corner the 8-term tensor product.  The constants are the reference's decimal literals. */
void apply_BCs_p2(level_type *L, int x_id, int shape) {
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (L->box_dim < 2) { apply_BCs_p1(L, x_id, shape); return; }
  const double t0 = now();
  int n;
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < L->boundary_condition.num_blocks[shape]; n++) {
    const blockCopy_type *blk = &L->boundary_condition.blocks[shape][n];
    const box_type *B = &L->my_boxes[blk->read.box];
    const int jS = B->jStride, kS = B->kStride, inward = 26 - blk->subtype;
    const int di = (inward % 3 - 1) * 1, dj = ((inward % 9) / 3 - 1) * jS, dk = (inward / 9 - 1) * kS;
    const int kind = (di != 0) + (dj != 0) + (dk != 0);
    double *x = vec(L, blk->read.box, x_id);
    int i, j, k;
    for (k = 0; k < blk->dim.k; k++) for (j = 0; j < blk->dim.j; j++) for (i = 0; i < blk->dim.i; i++) {
      const int ijk = (i + blk->read.i) + (j + blk->read.j) * jS + (k + blk->read.k) * kS;
      if (kind == 1) {
        const int s1 = di + dj + dk;
        x[ijk] = -2.0 * x[ijk + s1] + 0.333333333333333333 * x[ijk + 2 * s1];
      } else if (kind == 2) {
        int dr = -1, ds = -1;
        if (di == 0) { dr = dj; ds = dk; }
        if (dj == 0) { dr = di; ds = dk; }
        if (dk == 0) { dr = di; ds = dj; }
        x[ijk] = 4.000000000000000000 * x[ijk + dr + ds] - 0.666666666666666667 * x[ijk + 2 * dr + ds]
               - 0.666666666666666667 * x[ijk + dr + 2 * ds] + 0.111111111111111111 * x[ijk + 2 * dr + 2 * ds];
      } else {
        x[ijk] = -8.000000000000000000 * x[ijk + di + dj + dk]
               + 1.333333333333333333 * x[ijk + 2 * di + dj + dk] + 1.333333333333333333 * x[ijk + di + 2 * dj + dk]
               + 1.333333333333333333 * x[ijk + di + dj + 2 * dk]
               - 0.222222222222222222 * x[ijk + 2 * di + 2 * dj + dk] - 0.222222222222222222 * x[ijk + di + 2 * dj + 2 * dk]
               - 0.222222222222222222 * x[ijk + 2 * di + dj + 2 * dk] + 0.037037037037037037 * x[ijk + 2 * di + 2 * dj + 2 * dk];
      }
    }
  }
  L->timers.boundary_conditions += now() - t0;
}
/* boundary_fv.c:6-90: linear volume-averaged BC = the same one-point formula as apply_BCs_p1 */
void apply_BCs_v1(level_type *L, int x_id, int shape) { apply_BCs_p1(L, x_id, shape); }

/* Geometry shared by the finite-volume BCs: for a block whose DOMAIN normal is `subtype`, the axes with a
 * non-zero normal component (in i<j<k order) sit at ghost index -1 / box_dim and step inward; the other
 * axes run over the block extent (boundary_fv.c:145-152,172-185,214-223). */
typedef struct { int nn, pos[3], step[3], nfree, lo[3], len[3], fstride[3]; } bc_geom;
static bc_geom bc_geometry(const level_type *L, const blockCopy_type *blk, int jS, int kS) {
  bc_geom g; const int strides[3] = {1, jS, kS}, lo[3] = {blk->read.i, blk->read.j, blk->read.k}, len[3] = {blk->dim.i, blk->dim.j, blk->dim.k};
  const int d[3] = {blk->subtype % 3 - 1, (blk->subtype % 9) / 3 - 1, blk->subtype / 9 - 1};
  int ax; g.nn = 0; g.nfree = 0;
  for (ax = 0; ax < 3; ax++) {
    if (d[ax]) { g.pos[g.nn] = (d[ax] < 0 ? -1 : L->box_dim) * strides[ax]; g.step[g.nn] = -d[ax] * strides[ax]; g.nn++; }
    else { g.lo[g.nfree] = lo[ax]; g.len[g.nfree] = len[ax]; g.fstride[g.nfree] = strides[ax]; g.nfree++; }
  }
  for (ax = g.nfree; ax < 3; ax++) { g.lo[ax] = 0; g.len[ax] = 1; g.fstride[ax] = 0; }
  return g;
}

/* boundary_fv.c:101-250: quadratic volume-averaged BC on the first ghost layer; deeper ghost layers are zeroed */
void apply_BCs_v2(level_type *L, int x_id, int shape) {
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (L->box_dim < 2) { apply_BCs_v1(L, x_id, shape); return; }
  const double t0 = now();
  int n;
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < L->boundary_condition.num_blocks[shape]; n++) {
    const blockCopy_type *blk = &L->boundary_condition.blocks[shape][n];
    const box_type *B = &L->my_boxes[blk->read.box];
    const int jS = B->jStride, kS = B->kStride;
    double *x = vec(L, blk->read.box, x_id);
    int i, j, k, r, q;
    if (L->box_ghosts > 1)
      for (k = 0; k < blk->dim.k; k++) for (j = 0; j < blk->dim.j; j++) for (i = 0; i < blk->dim.i; i++)
        x[(i + blk->read.i) + (j + blk->read.j) * jS + (k + blk->read.k) * kS] = 0.0;
    const bc_geom g = bc_geometry(L, blk, jS, kS);
    for (q = 0; q < g.len[1]; q++) for (r = 0; r < g.len[0]; r++) {
      int ijk = (r + g.lo[0]) * g.fstride[0] + (q + g.lo[1]) * g.fstride[1];
      if (g.nn == 1) {
        const int dt = g.step[0]; ijk += g.pos[0];
        x[ijk] = -2.5 * x[ijk + dt] + 0.5 * x[ijk + 2 * dt];
      } else if (g.nn == 2) {
        const int ds = g.step[0], dt = g.step[1]; ijk += g.pos[0] + g.pos[1];
        x[ijk] = 6.25 * x[ijk + ds + dt] - 1.25 * x[ijk + 2 * ds + dt] - 1.25 * x[ijk + ds + 2 * dt] + 0.25 * x[ijk + 2 * ds + 2 * dt];
      } else {
        const int di = g.step[0], dj = g.step[1], dk = g.step[2]; ijk += g.pos[0] + g.pos[1] + g.pos[2];
        x[ijk] = -15.625 * x[ijk + di + dj + dk] + 3.125 * x[ijk + 2 * di + dj + dk] + 3.125 * x[ijk + di + 2 * dj + dk] + 3.125 * x[ijk + di + dj + 2 * dk]
               - 0.625 * x[ijk + 2 * di + 2 * dj + dk] - 0.625 * x[ijk + di + 2 * dj + 2 * dk] - 0.625 * x[ijk + 2 * di + dj + 2 * dk] + 0.125 * x[ijk + 2 * di + 2 * dj + 2 * dk];
      }
    }
  }
  L->timers.boundary_conditions += now() - t0;
}

/* boundary_fv.c:262-569: quartic volume-averaged BC.  One-dimensional rule from the four cells x1..x4
 * next to the boundary: near ghost N = (-77 x1 + 43 x2 - 17 x3 + 3 x4)/12, far ghost
 * F = (-505 x1 + 335 x2 - 145 x3 + 27 x4)/12; edges and corners apply it axis after axis (i, then j, then k). */
static inline double v4_near(double x1, double x2, double x3, double x4) { const double OneTwelfth = 1.0 / 12.0; return OneTwelfth * (-77.0 * x1 + 43.0 * x2 - 17.0 * x3 + 3.0 * x4); }
static inline double v4_far(double x1, double x2, double x3, double x4)  { const double OneTwelfth = 1.0 / 12.0; return OneTwelfth * (-505.0 * x1 + 335.0 * x2 - 145.0 * x3 + 27.0 * x4); }
void apply_BCs_v4(level_type *L, int x_id, int shape) {
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (L->box_ghosts < 2) { fprintf(stderr, "called quartic BC's with only 1 ghost zone!!!\n"); exit(0); }
  if (L->box_dim < 4) { apply_BCs_v2(L, x_id, shape); return; }
  const double t0 = now();
  int n;
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < L->boundary_condition.num_blocks[shape]; n++) {
    const blockCopy_type *blk = &L->boundary_condition.blocks[shape][n];
    const box_type *B = &L->my_boxes[blk->read.box];
    const int jS = B->jStride, kS = B->kStride;
    double *x = vec(L, blk->read.box, x_id);
    int i, j, k, r, q, m, p;
    if (L->box_ghosts > 2)
      for (k = 0; k < blk->dim.k; k++) for (j = 0; j < blk->dim.j; j++) for (i = 0; i < blk->dim.i; i++)
        x[(i + blk->read.i) + (j + blk->read.j) * jS + (k + blk->read.k) * kS] = 0.0;
    const bc_geom g = bc_geometry(L, blk, jS, kS);
    for (q = 0; q < g.len[1]; q++) for (r = 0; r < g.len[0]; r++) {
      int ijk = (r + g.lo[0]) * g.fstride[0] + (q + g.lo[1]) * g.fstride[1];
      if (g.nn == 1) {
        const int dt = g.step[0]; ijk += g.pos[0];
        const double x1 = x[ijk + dt], x2 = x[ijk + 2 * dt], x3 = x[ijk + 3 * dt], x4 = x[ijk + 4 * dt];
        x[ijk] = v4_near(x1, x2, x3, x4);
        x[ijk - dt] = v4_far(x1, x2, x3, x4);
      } else if (g.nn == 2) {
        const int ds = g.step[0], dt = g.step[1]; ijk += g.pos[0] + g.pos[1];
        double nr[5], fr[5];
        for (m = 1; m <= 4; m++) {
          const double a1 = x[ijk + ds + m * dt], a2 = x[ijk + 2 * ds + m * dt], a3 = x[ijk + 3 * ds + m * dt], a4 = x[ijk + 4 * ds + m * dt];
          nr[m] = v4_near(a1, a2, a3, a4); fr[m] = v4_far(a1, a2, a3, a4);
        }
        x[ijk]           = v4_near(nr[1], nr[2], nr[3], nr[4]);
        x[ijk - dt]      = v4_far(nr[1], nr[2], nr[3], nr[4]);
        x[ijk - ds]      = v4_near(fr[1], fr[2], fr[3], fr[4]);
        x[ijk - ds - dt] = v4_far(fr[1], fr[2], fr[3], fr[4]);
      } else {
        const int di = g.step[0], dj = g.step[1], dk = g.step[2]; ijk += g.pos[0] + g.pos[1] + g.pos[2];
        double nj[5][5], fj[5][5], nn_[5], nf_[5], fn_[5], ff_[5];   /* [j][k] after the i pass; [k] after the j pass */
        for (p = 1; p <= 4; p++) for (m = 1; m <= 4; m++) {
          const double a1 = x[ijk + di + m * dj + p * dk], a2 = x[ijk + 2 * di + m * dj + p * dk], a3 = x[ijk + 3 * di + m * dj + p * dk], a4 = x[ijk + 4 * di + m * dj + p * dk];
          nj[m][p] = v4_near(a1, a2, a3, a4); fj[m][p] = v4_far(a1, a2, a3, a4);
        }
        for (p = 1; p <= 4; p++) {
          nn_[p] = v4_near(nj[1][p], nj[2][p], nj[3][p], nj[4][p]); nf_[p] = v4_far(nj[1][p], nj[2][p], nj[3][p], nj[4][p]);
          fn_[p] = v4_near(fj[1][p], fj[2][p], fj[3][p], fj[4][p]); ff_[p] = v4_far(fj[1][p], fj[2][p], fj[3][p], fj[4][p]);
        }
        x[ijk]                = v4_near(nn_[1], nn_[2], nn_[3], nn_[4]);
        x[ijk - dk]           = v4_far(nn_[1], nn_[2], nn_[3], nn_[4]);
        x[ijk - dj]           = v4_near(nf_[1], nf_[2], nf_[3], nf_[4]);
        x[ijk - dj - dk]      = v4_far(nf_[1], nf_[2], nf_[3], nf_[4]);
        x[ijk - di]           = v4_near(fn_[1], fn_[2], fn_[3], fn_[4]);
        x[ijk - di - dk]      = v4_far(fn_[1], fn_[2], fn_[3], fn_[4]);
        x[ijk - di - dj]      = v4_near(ff_[1], ff_[2], ff_[3], ff_[4]);
        x[ijk - di - dj - dk] = v4_far(ff_[1], ff_[2], ff_[3], ff_[4]);
      }
    }
  }
  L->timers.boundary_conditions += now() - t0;
}

/* boundary_fv.c:573-681: fill the ghost cells of the face coefficients by polynomial extrapolation along the
 * direction pointing back into the box (the coefficient normal to a face keeps its face value).  The
 * reference walks each block in k,j,i order IN PLACE, so a deeper ghost cell may read a shallower one that
 * was (high side) or was not yet (low side) updated; the same order is kept here.
 * BLOCKS depend on each other too: the deeper cells of an edge block are formed from the shallower layer of a FACE block of the same box
 * (one step along the edge's diagonal), so their values depend on which block ran first.  The reference threads this loop over the blocks
 * (PRAGMA_THREAD_ACROSS_BLOCKS), i.e. with several threads those cells -- which no stencil reads: apply_op_ijk only takes coefficient
 * differences along the axes -- are a race; with ONE thread the blocks run in list order, which is what this restatement does and what the
 * per-operator fixtures (tests/golden/ops_golden.json, generated with OMP_NUM_THREADS=1) pin. */
void extrapolate_betas(level_type *L) {
  if (L->boundary_condition.type == BC_PERIODIC) return;
  const double t0 = now();
  const int shape = 0;
  int n;
  for (n = 0; n < L->boundary_condition.num_blocks[shape]; n++) {
    const blockCopy_type *blk = &L->boundary_condition.blocks[shape][n];
    const box_type *B = &L->my_boxes[blk->read.box];
    const int jS = B->jStride, kS = B->kStride, ilo = blk->read.i, jlo = blk->read.j, klo = blk->read.k;
    int subtype = 13;
    if (ilo < 0) subtype -= 1;
    if (jlo < 0) subtype -= 3;
    if (klo < 0) subtype -= 9;
    if (ilo >= L->box_dim) subtype += 1;
    if (jlo >= L->box_dim) subtype += 3;
    if (klo >= L->box_dim) subtype += 9;
    const int normal = 26 - subtype, di = normal % 3 - 1, dj = (normal % 9) / 3 - 1, dk = normal / 9 - 1;
    const int bs[3] = { dj * jS + dk * kS, di + dk * kS, di + dj * jS };
    const int skip_lo[3] = {12, 10, 4}, skip_hi[3] = {14, 16, 22};
    double *beta[3] = { vec(L, blk->read.box, VECTOR_BETA_I), vec(L, blk->read.box, VECTOR_BETA_J), vec(L, blk->read.box, VECTOR_BETA_K) };
    int i, j, k, c;
    for (k = 0; k < blk->dim.k; k++) for (j = 0; j < blk->dim.j; j++) for (i = 0; i < blk->dim.i; i++) {
      const int ijk = (i + ilo) + (j + jlo) * jS + (k + klo) * kS;
      for (c = 0; c < 3; c++) {
        if (subtype == skip_lo[c] || subtype == skip_hi[c]) continue;
        double *bb = beta[c]; const int st = bs[c];
        if (L->box_dim >= 5)      bb[ijk] = 5.0 * bb[ijk + st] - 10.0 * bb[ijk + 2 * st] + 10.0 * bb[ijk + 3 * st] - 5.0 * bb[ijk + 4 * st] + bb[ijk + 5 * st];
        else if (L->box_dim >= 4) bb[ijk] = 4.0 * bb[ijk + st] - 6.0 * bb[ijk + 2 * st] + 4.0 * bb[ijk + 3 * st] - bb[ijk + 4 * st];
        else if (L->box_dim >= 2) bb[ijk] = 2.0 * bb[ijk + st] - bb[ijk + 2 * st];
      }
    }
  }
  L->timers.boundary_conditions += now() - t0;
}

void apply_BCs(level_type *L, int x_id, int shape) { /* plugin dispatch, operators.7pt.c:47 */
  hpgmg_config c;
  hpgmg_get_config(&c);
  switch (c.op) {
    case HPGMG_OP_7PT:  apply_BCs_p1(L, x_id, shape); break;
    case HPGMG_OP_27PT: apply_BCs_p2(L, x_id, shape); break;
    case HPGMG_OP_FV2:  apply_BCs_v2(L, x_id, shape); break;
    default:            apply_BCs_v4(L, x_id, shape); break;
  }
}

/* ---------------------------------------------------------------- smoothers */
static void cheby_coefficients(const level_type *L, int degree, double *c1, double *c2) {
  /* operators/chebyshev.c:22-40 */
  double beta = 1.000 * L->dominant_eigenvalue_of_DinvA;
  double alpha = 0.125000 * beta;
  double theta = 0.5 * (beta + alpha);
  double delta = 0.5 * (beta - alpha);
  double sigma = theta / delta;
  double rho_n = 1 / sigma;
  int s;
  c1[0] = 0.0;
  c2[0] = 1 / theta;
  for (s = 1; s < degree; s++) {
    double rho_nm1 = rho_n;
    rho_n = 1.0 / (2.0 * sigma - rho_nm1);
    c1[s] = rho_n * rho_nm1;
    c2[s] = rho_n * 2.0 / delta;
  }
}

void smooth(level_type *L, int x_id, int rhs_id, double a, double b) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  const int sweeps = hpgmg_smooth_sweeps(), shape = stencil_get_shape();
  int s;
  if (cfg.smoother == HPGMG_SMOOTH_CHEBY) {               /* operators/chebyshev.c:8-100 */
    double c1[16], c2[16];
    if (L->dominant_eigenvalue_of_DinvA <= 0.0 && L->my_rank == 0) fprintf(stderr, "dominant_eigenvalue_of_DinvA <= 0.0 !\n");
    cheby_coefficients(L, sweeps, c1, c2);
    for (s = 0; s < sweeps; s++) {
      const int src = (s & 1) ? VECTOR_TEMP : x_id, dst = (s & 1) ? x_id : VECTOR_TEMP;
      exchange_boundary(L, src, shape);
      apply_BCs(L, src, shape);
      const double t0 = now();
      DISPATCH(cheby_sweep, L, src, dst, rhs_id, a, b, c1[s % sweeps], c2[s % sweeps]);
      L->timers.smooth += now() - t0;
    }
  } else if (cfg.smoother == HPGMG_SMOOTH_GSRB) {         /* operators/gsrb.c:24-132 */
    const int oop = hpgmg_gsrb_out_of_place();
    for (s = 0; s < sweeps; s++) {
      const int src = (oop && (s & 1)) ? VECTOR_TEMP : x_id, dst = oop ? ((s & 1) ? x_id : VECTOR_TEMP) : x_id;
      exchange_boundary(L, src, shape);
      apply_BCs(L, src, shape);
      const double t0 = now();
      DISPATCH(gsrb_sweep, L, src, dst, rhs_id, a, b, s, oop);
      L->timers.smooth += now() - t0;
    }
  } else {                                                /* operators/jacobi.c:8-65 */
    const double weight = 2.0 / 3.0;
    for (s = 0; s < sweeps; s++) {
      const int src = (s & 1) ? VECTOR_TEMP : x_id, dst = (s & 1) ? x_id : VECTOR_TEMP;
      exchange_boundary(L, src, shape);
      apply_BCs(L, src, shape);
      const double t0 = now();
      DISPATCH(jacobi_sweep, L, src, dst, rhs_id, a, b, weight);
      L->timers.smooth += now() - t0;
    }
  }
}

void residual(level_type *L, int res_id, int x_id, int rhs_id, double a, double b) { /* operators/residual.c:9-51 */
  exchange_boundary(L, x_id, stencil_get_shape());
  apply_BCs(L, x_id, stencil_get_shape());
  const double t0 = now();
  DISPATCH(residual_sweep, L, res_id, x_id, rhs_id, a, b);
  L->timers.residual += now() - t0;
}
void apply_op(level_type *L, int Ax_id, int x_id, double a, double b) {             /* operators/apply_op.c:9-48 */
  exchange_boundary(L, x_id, stencil_get_shape());
  apply_BCs(L, x_id, stencil_get_shape());
  const double t0 = now();
  DISPATCH(residual_sweep, L, Ax_id, x_id, -1, a, b);
  L->timers.apply_op += now() - t0;
}

/* ---------------------------------------------------------------- restriction
 * operators/restriction.c:6-94 (block kernel) and :104-212 (pack/local/unpack). */
static void restrict_block(level_type *Lc, int id_c, level_type *Lf, int id_f, const blockCopy_type *blk, int type) {
  double *r, *w; int rj, rk, wj, wk, i, j, k;
  resolve(Lf, id_f, blk, 0, &r, &rj, &rk);
  resolve(Lc, id_c, blk, 1, &w, &wj, &wk);
  for (k = 0; k < blk->dim.k; k++) for (j = 0; j < blk->dim.j; j++) for (i = 0; i < blk->dim.i; i++) {
    const double *f = r + 2 * i + 2 * j * rj + 2 * k * rk;
    double v;
    switch (type) {
      case RESTRICT_CELL:   v = (f[0] + f[1] + f[rj] + f[1 + rj] + f[rk] + f[1 + rk] + f[rj + rk] + f[1 + rj + rk]) * 0.125; break;
      case RESTRICT_FACE_I: v = (f[0] + f[rj] + f[rk] + f[rj + rk]) * 0.25; break;
      case RESTRICT_FACE_J: v = (f[0] + f[1] + f[rk] + f[1 + rk]) * 0.25; break;
      default:              v = (f[0] + f[1] + f[rj] + f[1 + rj]) * 0.25; break;
    }
    w[i + j * wj + k * wk] = v;
  }
}
void restriction(level_type *Lc, int id_c, level_type *Lf, int id_f, int type) {
  const double t0 = now();
  communicator_type *S = &Lf->restriction[type], *R = &Lc->restriction[type];
  int n;
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < S->num_blocks[0]; n++) restrict_block(Lc, id_c, Lf, id_f, &S->blocks[0][n], type);
  transport_phase(R, S, (Lf->tag << 4) | 0x5);
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < S->num_blocks[1]; n++) restrict_block(Lc, id_c, Lf, id_f, &S->blocks[1][n], type);
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < R->num_blocks[2]; n++) copy_block(Lc, id_c, &R->blocks[2][n]);
  Lf->timers.restriction_total += now() - t0;
}

#define HPGMG_V4_C1 (22.0 / 128.0)   /* checked against interpolation_v4.c */
#define HPGMG_V4_C2 (-3.0 / 128.0)
/* ---------------------------------------------------------------- interpolation
 * operators/interpolation_p0.c:6-46 (piecewise constant) and
 * operators/interpolation_p1.c:8-65 (trilinear; an even fine cell leans on the
 * coarse neighbour behind it, an odd one on the neighbour ahead). */
/* 1-D rules of the tensor-product interpolations; v[] = coarse line, centre at v[R]; parity of the fine child.
 *   order 2 = p2  interpolation_p2.c:90-92,150-205 : even w1*c1 + w0*c0 + w2*c2, odd w1*c1 + w2*c0 + w0*c2
 *   order 3 = v2  interpolation_v2.c:111-113       : c1 +- (1/8)*(c0 - c2)
 *   order 4 = v4  interpolation_v4.c (c1,c2 there) : c2 +- C1*(c1 - c3) +- C2*(c0 - c4)                        */
static inline double interp_rule(int order, int odd, const double *v) {
  if (order == 2) {
    const double w0 = 5.0 / 32.0, w1 = 30.0 / 32.0, w2 = -3.0 / 32.0;
    return odd ? (w1 * v[1] + w2 * v[0] + w0 * v[2]) : (w1 * v[1] + w0 * v[0] + w2 * v[2]);
  } else if (order == 3) {
    const double c1 = 1.0 / 8.0;
    return odd ? (v[1] - c1 * (v[0] - v[2])) : (v[1] + c1 * (v[0] - v[2]));
  } else {
    const double c1 = HPGMG_V4_C1, c2 = HPGMG_V4_C2;
    return odd ? (v[2] - c1 * (v[1] - v[3]) - c2 * (v[0] - v[4])) : (v[2] + c1 * (v[1] - v[3]) + c2 * (v[0] - v[4]));
  }
}
/* dimension by dimension, i then j then k, exactly as the reference forms f?c??, f??c?, f??? */
static inline double interp_tensor(int order, const double *c, int jS, int kS, int oi, int oj, int ok) {
  const int R = (order == 4) ? 2 : 1, W = 2 * R + 1;
  double line[5], tj[5], tk[5];
  int ii, jj, kk;
  for (kk = 0; kk < W; kk++) {
    for (jj = 0; jj < W; jj++) {
      for (ii = 0; ii < W; ii++) line[ii] = c[(ii - R) + (jj - R) * jS + (kk - R) * kS];
      tj[jj] = interp_rule(order, oi, line);
    }
    tk[kk] = interp_rule(order, oj, tj);
  }
  return interp_rule(order, ok, tk);
}

static void interp_block(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c, const blockCopy_type *blk, int order) {
  double *r, *w; int rj, rk, wj, wk, i, j, k;
  resolve(Lc, id_c, blk, 0, &r, &rj, &rk);
  resolve(Lf, id_f, blk, 1, &w, &wj, &wk);
  for (k = 0; k < 2 * blk->dim.k; k++) for (j = 0; j < 2 * blk->dim.j; j++) for (i = 0; i < 2 * blk->dim.i; i++) {
    double *fw = w + i + j * wj + k * wk;
    const double *c = r + (i >> 1) + (j >> 1) * rj + (k >> 1) * rk;
    if (order == 0) {
      *fw = prescale * (*fw) + c[0];
    } else if (order >= 2) {
      *fw = prescale * (*fw) + interp_tensor(order, c, rj, rk, i & 1, j & 1, k & 1);
    } else {
      const int di = (i & 1) ? 1 : -1, dj = (j & 1) ? rj : -rj, dk = (k & 1) ? rk : -rk;
      *fw = prescale * (*fw) + 0.421875 * c[0] + 0.140625 * c[dk] + 0.140625 * c[dj] + 0.046875 * c[dj + dk]
          + 0.140625 * c[di] + 0.046875 * c[di + dk] + 0.046875 * c[di + dj] + 0.015625 * c[di + dj + dk];
    }
  }
}
static void interpolation_generic(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c, int order, int tagbits) {
  const double t0 = now();
  communicator_type *S = &Lc->interpolation, *R = &Lf->interpolation;
  int n;
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < S->num_blocks[0]; n++) interp_block(Lf, id_f, 0.0, Lc, id_c, &S->blocks[0][n], order);
  transport_phase(R, S, (Lf->tag << 4) | tagbits);
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < S->num_blocks[1]; n++) interp_block(Lf, id_f, prescale, Lc, id_c, &S->blocks[1][n], order);
  _Pragma("omp parallel for schedule(static,1)")
  for (n = 0; n < R->num_blocks[2]; n++) increment_block(Lf, id_f, prescale, &R->blocks[2][n]);
  Lf->timers.interpolation_total += now() - t0;
}
static void interpolation_p0(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  interpolation_generic(Lf, id_f, prescale, Lc, id_c, 0, 0x6);      /* interpolation_p0.c:52-159 */
}
static void interpolation_p1(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  exchange_boundary(Lc, id_c, STENCIL_SHAPE_BOX);                    /* interpolation_p1.c:70-72 */
  apply_BCs_p1(Lc, id_c, STENCIL_SHAPE_BOX);
  interpolation_generic(Lf, id_f, prescale, Lc, id_c, 1, 0x7);
}
static void interpolation_p2(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  exchange_boundary(Lc, id_c, STENCIL_SHAPE_BOX);                    /* interpolation_p2.c:228-230 */
  apply_BCs_p2(Lc, id_c, STENCIL_SHAPE_BOX);
  interpolation_generic(Lf, id_f, prescale, Lc, id_c, 2, 0x7);
}
static void interpolation_v2(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  exchange_boundary(Lc, id_c, STENCIL_SHAPE_BOX);                    /* interpolation_v2.c:210-212 */
  apply_BCs_v2(Lc, id_c, STENCIL_SHAPE_BOX);
  interpolation_generic(Lf, id_f, prescale, Lc, id_c, 3, 0x7);
}
static void interpolation_v4(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  exchange_boundary(Lc, id_c, STENCIL_SHAPE_BOX);                    /* interpolation_v4.c:276-278 */
  apply_BCs_v4(Lc, id_c, STENCIL_SHAPE_BOX);
  interpolation_generic(Lf, id_f, prescale, Lc, id_c, 4, 0x7);
}
/* which interpolation each plugin wires up: operators.7pt.c:278-279, .27pt.c:150-151, .fv2.c:151-152, .fv4.c:201-202 */
void interpolation_vcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  hpgmg_config c; hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_7PT) interpolation_p0(Lf, id_f, prescale, Lc, id_c);
  else if (c.op == HPGMG_OP_27PT) interpolation_p2(Lf, id_f, prescale, Lc, id_c);
  else interpolation_v2(Lf, id_f, prescale, Lc, id_c);
}
void interpolation_fcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  hpgmg_config c; hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_7PT) interpolation_p1(Lf, id_f, prescale, Lc, id_c);
  else if (c.op == HPGMG_OP_27PT) interpolation_p2(Lf, id_f, prescale, Lc, id_c);
  else if (c.op == HPGMG_OP_FV2) interpolation_v2(Lf, id_f, prescale, Lc, id_c);
  else interpolation_v4(Lf, id_f, prescale, Lc, id_c);
}

/* ---------------------------------------------------------------- BLAS-1 (operators/misc.c) */
#define FOR_TILES(L, ...) do { int t_; \
  _Pragma("omp parallel for schedule(static,1)") \
  for (t_ = 0; t_ < (L)->num_my_blocks; t_++) { \
    const blockCopy_type *T = &(L)->my_blocks[t_]; const int box = T->read.box; \
    const box_type *B = &(L)->my_boxes[box]; const int jS = B->jStride, kS = B->kStride; \
    int ilo = T->read.i, jlo = T->read.j, klo = T->read.k, ihi = ilo + T->dim.i, jhi = jlo + T->dim.j, khi = klo + T->dim.k; \
    int i, j, k; (void)i; (void)j; (void)k; (void)jS; (void)kS; (void)ilo; (void)jlo; (void)klo; (void)ihi; (void)jhi; (void)khi; \
    __VA_ARGS__ } } while (0)
#define FOR_CELLS for (k = klo; k < khi; k++) for (j = jlo; j < jhi; j++) for (i = ilo; i < ihi; i++)

static void fill_with_ghosts(level_type *L, int id, double inside) { /* misc.c:6-44 zero_vector, :48-89 init_vector */
  const double t0 = now();
  FOR_TILES(L, {
    const int g = B->ghosts, dim = B->dim;
    double *v = vec(L, box, id);
    if (ilo <= 0) ilo -= g;   if (jlo <= 0) jlo -= g;   if (klo <= 0) klo -= g;
    if (ihi >= dim) ihi += g; if (jhi >= dim) jhi += g; if (khi >= dim) khi += g;
    FOR_CELLS {
      int ghost = (i < 0) || (j < 0) || (k < 0) || (i >= dim) || (j >= dim) || (k >= dim);
      v[i + j * jS + k * kS] = ghost ? 0.0 : inside;
    }
  });
  L->timers.blas1 += now() - t0;
}
void zero_vector(level_type *L, int id) { fill_with_ghosts(L, id, 0.0); }
void init_vector(level_type *L, int id, double s) { fill_with_ghosts(L, id, s); }

void add_vectors(level_type *L, int id_c, double sa, int id_a, double sb, int id_b) { /* misc.c:94-126 */
  const double t0 = now();
  FOR_TILES(L, { double *c = vec(L, box, id_c); const double *pa = vec(L, box, id_a), *pb = vec(L, box, id_b);
    FOR_CELLS { const int ijk = i + j * jS + k * kS; c[ijk] = sa * pa[ijk] + sb * pb[ijk]; } });
  L->timers.blas1 += now() - t0;
}
void mul_vectors(level_type *L, int id_c, double s, int id_a, int id_b) {             /* misc.c:131-163 */
  const double t0 = now();
  FOR_TILES(L, { double *c = vec(L, box, id_c); const double *pa = vec(L, box, id_a), *pb = vec(L, box, id_b);
    FOR_CELLS { const int ijk = i + j * jS + k * kS; c[ijk] = s * pa[ijk] * pb[ijk]; } });
  L->timers.blas1 += now() - t0;
}
void invert_vector(level_type *L, int id_c, double s, int id_a) {                     /* misc.c:168-199 */
  const double t0 = now();
  FOR_TILES(L, { double *c = vec(L, box, id_c); const double *pa = vec(L, box, id_a);
    FOR_CELLS { const int ijk = i + j * jS + k * kS; c[ijk] = s / pa[ijk]; } });
  L->timers.blas1 += now() - t0;
}
void scale_vector(level_type *L, int id_c, double s, int id_a) {                      /* misc.c:204-234 */
  const double t0 = now();
  FOR_TILES(L, { double *c = vec(L, box, id_c); const double *pa = vec(L, box, id_a);
    FOR_CELLS { const int ijk = i + j * jS + k * kS; c[ijk] = s * pa[ijk]; } });
  L->timers.blas1 += now() - t0;
}
void shift_vector(level_type *L, int id_c, int id_a, double shift) {                  /* misc.c:386-415 */
  const double t0 = now();
  FOR_TILES(L, { double *c = vec(L, box, id_c); const double *pa = vec(L, box, id_a);
    FOR_CELLS { const int ijk = i + j * jS + k * kS; c[ijk] = pa[ijk] + shift; } });
  L->timers.blas1 += now() - t0;
}

static double allreduce_scalar(level_type *L, double v, int op) {
  const hpgmg_transport *T = hpgmg_get_transport();
  if (T && T->size > 1) {
    hpgmg_level_ext *X = hpgmg_level_ext_get(L);
    if (X->num_active_ranks > 1) T->allreduce(T->ctx, &v, 1, op, X->active_ranks, X->num_active_ranks);
  }
  return v;
}

/* Sums: one partial per tile in k,j,i order, partials added in tile-list order
 * (= the reference run with OMP_NUM_THREADS=1; misc.c:239-282, :336-378). */
double dot(level_type *L, int id_a, int id_b) {
  const double t0 = now();
  double *partial = (double *)calloc((size_t)L->num_my_blocks + 1, sizeof(double)), sum = 0.0;
  int n;
  FOR_TILES(L, { const double *pa = vec(L, box, id_a), *pb = vec(L, box, id_b); double acc = 0.0;
    FOR_CELLS { const int ijk = i + j * jS + k * kS; acc += pa[ijk] * pb[ijk]; } partial[t_] = acc; });
  for (n = 0; n < L->num_my_blocks; n++) sum += partial[n];
  free(partial);
  L->timers.blas1 += now() - t0;
  return allreduce_scalar(L, sum, HPGMG_REDUCE_SUM);
}
double mean(level_type *L, int id_a) {
  const double t0 = now();
  double *partial = (double *)calloc((size_t)L->num_my_blocks + 1, sizeof(double)), sum = 0.0;
  int n;
  FOR_TILES(L, { const double *pa = vec(L, box, id_a); double acc = 0.0;
    FOR_CELLS { acc += pa[i + j * jS + k * kS]; } partial[t_] = acc; });
  for (n = 0; n < L->num_my_blocks; n++) sum += partial[n];
  free(partial);
  L->timers.blas1 += now() - t0;
  sum = allreduce_scalar(L, sum, HPGMG_REDUCE_SUM);
  return sum / (double)((double)L->dim.i * (double)L->dim.j * (double)L->dim.k);
}
double norm(level_type *L, int id_a) {                                                /* misc.c:287-329: max norm */
  const double t0 = now();
  double *partial = (double *)calloc((size_t)L->num_my_blocks + 1, sizeof(double)), mx = 0.0;
  int n;
  FOR_TILES(L, { const double *pa = vec(L, box, id_a); double acc = 0.0;
    FOR_CELLS { double f = fabs(pa[i + j * jS + k * kS]); if (f > acc) acc = f; } partial[t_] = acc; });
  for (n = 0; n < L->num_my_blocks; n++) if (partial[n] > mx) mx = partial[n];
  free(partial);
  L->timers.blas1 += now() - t0;
  return allreduce_scalar(L, mx, HPGMG_REDUCE_MAX);
}
double error(level_type *L, int id_a, int id_b) {                                     /* misc.c:420-426 */
  add_vectors(L, VECTOR_TEMP, 1.0, id_a, -1.0, id_b);
  return norm(L, VECTOR_TEMP);
}
void color_vector(level_type *L, int id, int colors, int ic, int jc, int kc) {        /* misc.c:441-473 */
  const double t0 = now();
  FOR_TILES(L, { double *v = vec(L, box, id);
    FOR_CELLS {
      double si = ((i + B->low.i + ic) % colors == 0) ? 1.0 : 0.0;
      double sj = ((j + B->low.j + jc) % colors == 0) ? 1.0 : 0.0;
      double sk = ((k + B->low.k + kc) % colors == 0) ? 1.0 : 0.0;
      v[i + j * jS + k * kS] = si * sj * sk; } });
  L->timers.blas1 += now() - t0;
}
void random_vector(level_type *L, int id) {                                           /* misc.c:478-505 (literal) */
  const double t0 = now();
  FOR_TILES(L, { double *v = vec(L, box, id);
    FOR_CELLS { v[i + j * jS + k * kS] = -1.000 + 2.0 * (i ^ j ^ k ^ 0x1); } });
  L->timers.blas1 += now() - t0;
}

/* ---------------------------------------------------------------- problem setup
 * operators/problem.p6.c:6-35 (beta), :38-76 (u = X(x)X(y)X(z)), :79-135. */
static void evaluate_beta(double x, double y, double z, double *B, double *Bx, double *By, double *Bz) {
  double Bmin = 1.0, Bmax = 10.0;
  double c2 = (Bmax - Bmin) / 2, c1 = (Bmax + Bmin) / 2, c3 = 10.0;
  double xc = 0.50, yc = 0.50, zc = 0.50;
  double r2 = pow((x - xc), 2) + pow((y - yc), 2) + pow((z - zc), 2);
  double r2x = 2.0 * (x - xc), r2y = 2.0 * (y - yc), r2z = 2.0 * (z - zc);
  double r = pow(r2, 0.5);
  double rx = 0.5 * r2x * pow(r2, -0.5), ry = 0.5 * r2y * pow(r2, -0.5), rz = 0.5 * r2z * pow(r2, -0.5);
  *B  = c1 + c2 * tanh(c3 * (r - 0.25));
  *Bx = c2 * c3 * rx * (1 - pow(tanh(c3 * (r - 0.25)), 2));
  *By = c2 * c3 * ry * (1 - pow(tanh(c3 * (r - 0.25)), 2));
  *Bz = c2 * c3 * rz * (1 - pow(tanh(c3 * (r - 0.25)), 2));
}
static void evaluate_u(double x, double y, double z, double *U, double *Ux, double *Uy, double *Uz,
                       double *Uxx, double *Uyy, double *Uzz, int periodic) {
  double shift = periodic ? 1.0 / 21.0 : 0.0;
  double X   =  2.0 * pow(x, 6) -   6.0 * pow(x, 5) +  5.0 * pow(x, 4) - 1.0 * pow(x, 2) + shift;
  double Y   =  2.0 * pow(y, 6) -   6.0 * pow(y, 5) +  5.0 * pow(y, 4) - 1.0 * pow(y, 2) + shift;
  double Z   =  2.0 * pow(z, 6) -   6.0 * pow(z, 5) +  5.0 * pow(z, 4) - 1.0 * pow(z, 2) + shift;
  double Xx  = 12.0 * pow(x, 5) -  30.0 * pow(x, 4) + 20.0 * pow(x, 3) - 2.0 * x;
  double Yy  = 12.0 * pow(y, 5) -  30.0 * pow(y, 4) + 20.0 * pow(y, 3) - 2.0 * y;
  double Zz  = 12.0 * pow(z, 5) -  30.0 * pow(z, 4) + 20.0 * pow(z, 3) - 2.0 * z;
  double Xxx = 60.0 * pow(x, 4) - 120.0 * pow(x, 3) + 60.0 * pow(x, 2) - 2.0;
  double Yyy = 60.0 * pow(y, 4) - 120.0 * pow(y, 3) + 60.0 * pow(y, 2) - 2.0;
  double Zzz = 60.0 * pow(z, 4) - 120.0 * pow(z, 3) + 60.0 * pow(z, 2) - 2.0;
  *U = X * Y * Z;
  *Ux = Xx * Y * Z;  *Uy = X * Yy * Z;  *Uz = X * Y * Zz;
  *Uxx = Xxx * Y * Z; *Uyy = X * Yyy * Z; *Uzz = X * Y * Zzz;
}
/* operators/problem.fv.c:9-28 (beta), :71-87 (F), :90-140: cell/face AVERAGES to 4th order: point value
 * plus h^2/24 times the second derivatives in the averaged directions */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
static double fv_beta(double x, double y, double z, double h, int add_Bxx, int add_Byy, int add_Bzz) {
  double b = 0.25;
  double a = 2.0 * M_PI;
  double B   = 1.0 + b * sin(a * x) * sin(a * y) * sin(a * z);
  double Bxx = -a * a * b * sin(a * x) * sin(a * y) * sin(a * z);
  double Byy = -a * a * b * sin(a * x) * sin(a * y) * sin(a * z);
  double Bzz = -a * a * b * sin(a * x) * sin(a * y) * sin(a * z);
  if (add_Bxx) B += (h * h / 24.0) * Bxx;
  if (add_Byy) B += (h * h / 24.0) * Byy;
  if (add_Bzz) B += (h * h / 24.0) * Bzz;
  return B;
}
static double fv_F(double x, double y, double z, double h, int add_Fxx, int add_Fyy, int add_Fzz) {
  double a = 2.0 * M_PI;
  double p = 7.0;
  double F   = pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p);
  double Fxx = -a * a * p * pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p) + a * a * p * (p - 1) * pow(sin(a * x), p - 2) * pow(sin(a * y), p) * pow(sin(a * z), p) * pow(cos(a * x), 2);
  double Fyy = -a * a * p * pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p) + a * a * p * (p - 1) * pow(sin(a * x), p) * pow(sin(a * y), p - 2) * pow(sin(a * z), p) * pow(cos(a * y), 2);
  double Fzz = -a * a * p * pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p) + a * a * p * (p - 1) * pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p - 2) * pow(cos(a * z), 2);
  if (add_Fxx) F += (h * h / 24.0) * Fxx;
  if (add_Fyy) F += (h * h / 24.0) * Fyy;
  if (add_Fzz) F += (h * h / 24.0) * Fzz;
  return F;
}
static void initialize_problem_fv(level_type *L, double h, const hpgmg_config *cfg) {
  int box;
  L->h = h;
  for (box = 0; box < L->num_my_boxes; box++) {
    const box_type *B = &L->my_boxes[box];
    const int jS = B->jStride, kS = B->kStride, g = B->ghosts, dim = B->dim;
    int i, j, k;
    _Pragma("omp parallel for private(k,j,i) collapse(3)")
    for (k = 0; k <= dim; k++) for (j = 0; j <= dim; j++) for (i = 0; i <= dim; i++) {
      const int ijk = (i + g) + (j + g) * jS + (k + g) * kS;
      double x = h * ((double)(i + B->low.i) + 0.5), y = h * ((double)(j + B->low.j) + 0.5), z = h * ((double)(k + B->low.k) + 0.5);
      double A = 1.0, Bi = 1.0, Bj = 1.0, Bk = 1.0;
      if (cfg->variable_coeff) {
        Bi = fv_beta(x - h * 0.5, y, z, h, 0, 1, 1);
        Bj = fv_beta(x, y - h * 0.5, z, h, 1, 0, 1);
        Bk = fv_beta(x, y, z - h * 0.5, h, 1, 1, 0);
      }
      double F = fv_F(x, y, z, h, 1, 1, 1);
      if (cfg->helmholtz) B->vectors[VECTOR_ALPHA][ijk] = A;
      B->vectors[VECTOR_BETA_I][ijk] = Bi;
      B->vectors[VECTOR_BETA_J][ijk] = Bj;
      B->vectors[VECTOR_BETA_K][ijk] = Bk;
      B->vectors[VECTOR_F][ijk] = F;
    }
  }
}

void initialize_problem(level_type *L, double h, double a, double b) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (cfg.op == HPGMG_OP_FV2 || cfg.op == HPGMG_OP_FV4) { initialize_problem_fv(L, h, &cfg); return; }
  L->h = h;
  int box;
  for (box = 0; box < L->num_my_boxes; box++) {
    const box_type *B = &L->my_boxes[box];
    const int jS = B->jStride, kS = B->kStride, g = B->ghosts, dim = B->dim;
    int i, j, k;
    _Pragma("omp parallel for private(k,j,i) collapse(3)")
    for (k = 0; k <= dim; k++) for (j = 0; j <= dim; j++) for (i = 0; i <= dim; i++) { /* <= : the high faces too */
      const int ijk = (i + g) + (j + g) * jS + (k + g) * kS;
      double x = h * ((double)(i + B->low.i) + 0.5), y = h * ((double)(j + B->low.j) + 0.5), z = h * ((double)(k + B->low.k) + 0.5);
      double A = 1.0, Bc = 1.0, Bx = 0.0, By = 0.0, Bz = 0.0, Bi = 1.0, Bj = 1.0, Bk = 1.0;
      double U, Ux, Uy, Uz, Uxx, Uyy, Uzz;
      if (cfg.variable_coeff) {
        evaluate_beta(x - h * 0.5, y, z, &Bi, &Bx, &By, &Bz);
        evaluate_beta(x, y - h * 0.5, z, &Bj, &Bx, &By, &Bz);
        evaluate_beta(x, y, z - h * 0.5, &Bk, &Bx, &By, &Bz);
        evaluate_beta(x, y, z, &Bc, &Bx, &By, &Bz);
      }
      evaluate_u(x, y, z, &U, &Ux, &Uy, &Uz, &Uxx, &Uyy, &Uzz, L->boundary_condition.type == BC_PERIODIC);
      double F = a * A * U - b * ((Bx * Ux + By * Uy + Bz * Uz) + Bc * (Uxx + Uyy + Uzz));
      B->vectors[VECTOR_BETA_I][ijk] = Bi;
      B->vectors[VECTOR_BETA_J][ijk] = Bj;
      B->vectors[VECTOR_BETA_K][ijk] = Bk;
      if (cfg.helmholtz) B->vectors[VECTOR_ALPHA][ijk] = A;
      B->vectors[VECTOR_F][ijk] = F;
    }
  }
}

/* ---------------------------------------------------------------- black-box rebuild
 * operators/rebuild.c:47-208: probe the operator (boundary conditions included) with colors^3 0/1
 * colourings; Aii and sum|Aij| accumulate in VECTOR_DINV and VECTOR_L1INV (or VECTOR_E when there is
 * no L1INV); then Dinv, L1inv and the Gershgorin bound. */
void rebuild_operator_blackbox(level_type *L, double a, double b, int colors) {
  if (L->dim.i < colors) colors = L->dim.i;
  if (L->dim.j < colors) colors = L->dim.j;
  if (L->dim.k < colors) colors = L->dim.k;
  if (L->my_rank == 0 && hpgmg_verbose) { fprintf(stdout, "  calculating D^{-1} exactly for level h=%e using %3d colors...  ", L->h, colors * colors * colors); fflush(stdout); }
  const int x_id = VECTOR_TEMP, Aii_id = VECTOR_DINV, sum_id = (hpgmg_vectors_reserved() > VECTOR_L1INV) ? VECTOR_L1INV : VECTOR_E;
  int ic, jc, kc;
  zero_vector(L, Aii_id);
  zero_vector(L, sum_id);
  for (kc = 0; kc < colors; kc++) for (jc = 0; jc < colors; jc++) for (ic = 0; ic < colors; ic++) {
    color_vector(L, x_id, colors, ic, jc, kc);
    exchange_boundary(L, x_id, stencil_get_shape());
    apply_BCs(L, x_id, stencil_get_shape());
    DISPATCH(blackbox_sweep, L, x_id, Aii_id, sum_id, a, b);
  }
  double *tile_max = (double *)malloc(((size_t)L->num_my_blocks + 1) * sizeof(double));
  FOR_TILES(L, {
    const double h2inv = 1.0 / (L->h * L->h);
    double *Aii = vec(L, box, Aii_id); double *sumAbsAij = vec(L, box, sum_id);
    double best = -1e9;
    FOR_CELLS {
      const int ijk = i + j * jS + k * kS;
      if (Aii[ijk] == 0.0) { printf("Aii[%d,%d,%d]==0.0 !!!\n", i + B->low.i, j + B->low.j, k + B->low.k); Aii[ijk] = a + b * h2inv; }
      double Di = (Aii[ijk] + sumAbsAij[ijk]) / Aii[ijk];
      if (Di > best) best = Di;
      if (Aii[ijk] >= 1.5 * sumAbsAij[ijk]) sumAbsAij[ijk] = 1.0 / (Aii[ijk]); else sumAbsAij[ijk] = 1.0 / (Aii[ijk] + 0.5 * sumAbsAij[ijk]);
      Aii[ijk] = 1.0 / Aii[ijk];
    }
    tile_max[t_] = best;
  });
  double lambda = -1e9;
  { int n; for (n = 0; n < L->num_my_blocks; n++) if (tile_max[n] > lambda) lambda = tile_max[n]; }
  free(tile_max);
  if (L->my_rank == 0 && hpgmg_verbose) fprintf(stdout, "done\n");
  { const hpgmg_transport *T = hpgmg_get_transport();
    if (T && T->size > 1) { int r, *all = (int *)malloc((size_t)T->size * sizeof(int)); for (r = 0; r < T->size; r++) all[r] = r;
      T->allreduce(T->ctx, &lambda, 1, HPGMG_REDUCE_MAX, all, T->size); free(all); } }
  { hpgmg_config cfg; hpgmg_get_config(&cfg);
    if (cfg.smoother == HPGMG_SMOOTH_CHEBY && L->my_rank == 0 && hpgmg_verbose) { fprintf(stdout, "  estimating  lambda_max... <%1.15e\n", lambda); fflush(stdout); } }
  L->dominant_eigenvalue_of_DinvA = lambda;
}

/* rebuild_operator of the plugins that use the black box: operators.27pt.c:96-121 (2 colours),
 * operators.fv2.c:98-124 (2), operators.fv4.c:145-172 (4, after extrapolate_betas) */
static void rebuild_operator_via_blackbox(level_type *L, level_type *from, double a, double b) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (from) {
    if (cfg.helmholtz) restriction(L, VECTOR_ALPHA, from, VECTOR_ALPHA, RESTRICT_CELL);
    restriction(L, VECTOR_BETA_I, from, VECTOR_BETA_I, RESTRICT_FACE_I);
    restriction(L, VECTOR_BETA_J, from, VECTOR_BETA_J, RESTRICT_FACE_J);
    restriction(L, VECTOR_BETA_K, from, VECTOR_BETA_K, RESTRICT_FACE_K);
  }
  if (cfg.op == HPGMG_OP_FV4) extrapolate_betas(L);
  if (cfg.helmholtz) exchange_boundary(L, VECTOR_ALPHA, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_I, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_J, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_K, STENCIL_SHAPE_BOX);
  rebuild_operator_blackbox(L, a, b, cfg.op == HPGMG_OP_FV4 ? 4 : 2);
  exchange_boundary(L, VECTOR_DINV, STENCIL_SHAPE_BOX);
}

/* ---------------------------------------------------------------- operator rebuild
 * operators.7pt.c:95-252: coarsen coefficients, fill their ghosts, then the
 * diagonal and a Gershgorin bound on lambda_max(D^-1 A) with Dirichlet faces
 * folded in through 0/1 validity masks. */
void rebuild_operator(level_type *L, level_type *from, double a, double b) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (cfg.op != HPGMG_OP_7PT) { rebuild_operator_via_blackbox(L, from, a, b); return; }
  if (L->my_rank == 0 && hpgmg_verbose) { fprintf(stdout, "  rebuilding operator for level...  h=%e  ", L->h); fflush(stdout); }
  if (from) {
    if (cfg.helmholtz) restriction(L, VECTOR_ALPHA, from, VECTOR_ALPHA, RESTRICT_CELL);
    restriction(L, VECTOR_BETA_I, from, VECTOR_BETA_I, RESTRICT_FACE_I);
    restriction(L, VECTOR_BETA_J, from, VECTOR_BETA_J, RESTRICT_FACE_J);
    restriction(L, VECTOR_BETA_K, from, VECTOR_BETA_K, RESTRICT_FACE_K);
  }
  if (cfg.helmholtz) exchange_boundary(L, VECTOR_ALPHA, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_I, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_J, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_K, STENCIL_SHAPE_BOX);

  const double t0 = now();
  double *tile_max = (double *)malloc(((size_t)L->num_my_blocks + 1) * sizeof(double));
  const int periodic = (L->boundary_condition.type == BC_PERIODIC);
  FOR_TILES(L, {
    const double h2inv = 1.0 / (L->h * L->h);
    const double *alpha = cfg.helmholtz ? vec(L, box, VECTOR_ALPHA) : NULL;
    const double *beta_i = vec(L, box, VECTOR_BETA_I), *beta_j = vec(L, box, VECTOR_BETA_J), *beta_k = vec(L, box, VECTOR_BETA_K);
    double *Dinv = vec(L, box, VECTOR_DINV), *L1inv = cfg.helmholtz ? vec(L, box, VECTOR_L1INV) : NULL;
    double best = -1e9;
    FOR_CELLS {
      const int ijk = i + j * jS + k * kS;
      double ilo_ok = 1.0, ihi_ok = 1.0, jlo_ok = 1.0, jhi_ok = 1.0, klo_ok = 1.0, khi_ok = 1.0;
      if (!periodic) {
        if (B->low.i + i - 1 < 0) ilo_ok = 0.0;          if (B->low.j + j - 1 < 0) jlo_ok = 0.0;          if (B->low.k + k - 1 < 0) klo_ok = 0.0;
        if (B->low.i + i + 1 >= L->dim.i) ihi_ok = 0.0;  if (B->low.j + j + 1 >= L->dim.j) jhi_ok = 0.0;  if (B->low.k + k + 1 >= L->dim.k) khi_ok = 0.0;
      }
      double sumAbsAij, Aii;
      if (cfg.variable_coeff) {
        sumAbsAij = fabs(b * h2inv) * ( fabs(beta_i[ijk] * ilo_ok) + fabs(beta_j[ijk] * jlo_ok) + fabs(beta_k[ijk] * klo_ok)
                                      + fabs(beta_i[ijk + 1] * ihi_ok) + fabs(beta_j[ijk + jS] * jhi_ok) + fabs(beta_k[ijk + kS] * khi_ok) );
        Aii = -b * h2inv * ( beta_i[ijk] * (ilo_ok - 2.0) + beta_j[ijk] * (jlo_ok - 2.0) + beta_k[ijk] * (klo_ok - 2.0)
                           + beta_i[ijk + 1] * (ihi_ok - 2.0) + beta_j[ijk + jS] * (jhi_ok - 2.0) + beta_k[ijk + kS] * (khi_ok - 2.0) );
        if (alpha) Aii += a * alpha[ijk];
      } else {
        sumAbsAij = fabs(b * h2inv) * (ilo_ok + jlo_ok + klo_ok + ihi_ok + jhi_ok + khi_ok);
        Aii = a - b * h2inv * (ilo_ok + jlo_ok + klo_ok + ihi_ok + jhi_ok + khi_ok - 12.0);
      }
      Dinv[ijk] = 1.0 / Aii;
      double Di = (Aii + sumAbsAij) / Aii;
      if (Di > best) best = Di;
      if (L1inv) { if (Aii >= 1.5 * sumAbsAij) L1inv[ijk] = 1.0 / (Aii); else L1inv[ijk] = 1.0 / (Aii + 0.5 * sumAbsAij); }
    }
    tile_max[t_] = best;
  });
  double lambda = -1e9;
  { int n; for (n = 0; n < L->num_my_blocks; n++) if (tile_max[n] > lambda) lambda = tile_max[n]; }
  free(tile_max);
  L->timers.blas1 += now() - t0;
  { const hpgmg_transport *T = hpgmg_get_transport();   /* MPI_COMM_WORLD max, operators.7pt.c:237 */
    if (T && T->size > 1) { int r, *all = (int *)malloc((size_t)T->size * sizeof(int)); for (r = 0; r < T->size; r++) all[r] = r;
      T->allreduce(T->ctx, &lambda, 1, HPGMG_REDUCE_MAX, all, T->size); free(all); } }
  if (L->my_rank == 0 && hpgmg_verbose) fprintf(stdout, "eigenvalue_max<%e\n", lambda);
  L->dominant_eigenvalue_of_DinvA = lambda;
  exchange_boundary(L, VECTOR_DINV, STENCIL_SHAPE_BOX);
  if (cfg.helmholtz) exchange_boundary(L, VECTOR_L1INV, STENCIL_SHAPE_BOX);
}

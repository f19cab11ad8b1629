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
void hpgmg_segment_begin(long long key) { (void)key; }
void hpgmg_segment_end(void) {}
int hpgmg_vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg) {
  (void)levels; (void)n; (void)e_id; (void)R_id; (void)a; (void)b; (void)leg; return 0; }

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

enum { K_7PT_VC_HELM = 0, K_7PT_VC_POIS, K_7PT_CC, K_UNSUPPORTED };
static int kernel_variant(void) {
  hpgmg_config c;
  hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_7PT) {
    if (!c.variable_coeff) return K_7PT_CC;
    return c.helmholtz ? K_7PT_VC_HELM : K_7PT_VC_POIS;
  }
  fprintf(stderr, "oracle: operator %d not restated yet\n", c.op);
  exit(1);
}
#define DISPATCH(fn, ...) do { switch (kernel_variant()) { \
    case K_7PT_VC_HELM: fn##_7pt_vc_helmholtz(__VA_ARGS__); break; \
    case K_7PT_VC_POIS: fn##_7pt_vc_poisson(__VA_ARGS__); break; \
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
static void not_yet(const char *what) { fprintf(stderr, "oracle: %s not restated yet\n", what); exit(1); }
void apply_BCs_p2(level_type *L, int x_id, int shape) { (void)L; (void)x_id; (void)shape; not_yet("apply_BCs_p2"); }
void apply_BCs_v1(level_type *L, int x_id, int shape) { (void)L; (void)x_id; (void)shape; not_yet("apply_BCs_v1"); }
void apply_BCs_v2(level_type *L, int x_id, int shape) { (void)L; (void)x_id; (void)shape; not_yet("apply_BCs_v2"); }
void apply_BCs_v4(level_type *L, int x_id, int shape) { (void)L; (void)x_id; (void)shape; not_yet("apply_BCs_v4"); }
void extrapolate_betas(level_type *L) { (void)L; not_yet("extrapolate_betas"); }
void rebuild_operator_blackbox(level_type *L, double a, double b, int c) { (void)L; (void)a; (void)b; (void)c; not_yet("rebuild_operator_blackbox"); }

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

/* ---------------------------------------------------------------- interpolation
 * operators/interpolation_p0.c:6-46 (piecewise constant) and
 * operators/interpolation_p1.c:8-65 (trilinear; an even fine cell leans on the
 * coarse neighbour behind it, an odd one on the neighbour ahead). */
static void interp_block(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c, const blockCopy_type *blk, int order) {
  double *r, *w; int rj, rk, wj, wk, i, j, k;
  resolve(Lc, id_c, blk, 0, &r, &rj, &rk);
  resolve(Lf, id_f, blk, 1, &w, &wj, &wk);
  for (k = 0; k < 2 * blk->dim.k; k++) for (j = 0; j < 2 * blk->dim.j; j++) for (i = 0; i < 2 * blk->dim.i; i++) {
    double *fw = w + i + j * wj + k * wk;
    const double *c = r + (i >> 1) + (j >> 1) * rj + (k >> 1) * rk;
    if (order == 0) {
      *fw = prescale * (*fw) + c[0];
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
void interpolation_vcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  hpgmg_config c; hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_7PT) interpolation_p0(Lf, id_f, prescale, Lc, id_c); else not_yet("interpolation_vcycle for this operator");
}
void interpolation_fcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  hpgmg_config c; hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_7PT) interpolation_p1(Lf, id_f, prescale, Lc, id_c); else not_yet("interpolation_fcycle for this operator");
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
void initialize_problem(level_type *L, double h, double a, double b) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (cfg.op != HPGMG_OP_7PT && cfg.op != HPGMG_OP_27PT) not_yet("initialize_problem (problem.fv)");
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

/* ---------------------------------------------------------------- operator rebuild
 * operators.7pt.c:95-252: coarsen coefficients, fill their ghosts, then the
 * diagonal and a Gershgorin bound on lambda_max(D^-1 A) with Dirichlet faces
 * folded in through 0/1 validity masks. */
void rebuild_operator(level_type *L, level_type *from, double a, double b) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (cfg.op != HPGMG_OP_7PT) not_yet("rebuild_operator for this operator");
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

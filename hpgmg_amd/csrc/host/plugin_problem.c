/*
 * plugin_problem.c -- initialize_problem (operators/problem.p6.c, problem.fv.c) and rebuild_operator (operators.7pt.c:95-252).
 * Part of the operator plugin (see operators_hip.c); no arithmetic on vector data happens here.
 */
#include "plugin_internal.h"

/* ---------------------------------------------------------------- problem.p6.c:79-135
 * Analytic coefficients and right-hand side are evaluated on the host with the
 * same libm calls as the reference (pow, tanh) and staged into device memory box
 * by box; this is untimed setup and keeps beta/F bit-identical to the reference. */
static void eval_beta(double x, double y, double z, double *B, double *Bx, double *By, double *Bz) {
  const double Bmin = 1.0, Bmax = 10.0, c2 = (Bmax - Bmin) / 2, c1 = (Bmax + Bmin) / 2, c3 = 10.0;
  const double xc = 0.50, yc = 0.50, zc = 0.50;
  double r2 = pow((x - xc), 2) + pow((y - yc), 2) + pow((z - zc), 2);
  double r2x = 2.0 * (x - xc), r2y = 2.0 * (y - yc), r2z = 2.0 * (z - zc);
  double r = pow(r2, 0.5);
  double rx = 0.5 * r2x * pow(r2, -0.5), ry = 0.5 * r2y * pow(r2, -0.5), rz = 0.5 * r2z * pow(r2, -0.5);
  *B  = c1 + c2 * tanh(c3 * (r - 0.25));
  *Bx = c2 * c3 * rx * (1 - pow(tanh(c3 * (r - 0.25)), 2));
  *By = c2 * c3 * ry * (1 - pow(tanh(c3 * (r - 0.25)), 2));
  *Bz = c2 * c3 * rz * (1 - pow(tanh(c3 * (r - 0.25)), 2));
}
static void eval_poly(double t, double shift, double *P, double *Pt, double *Ptt) {
  *P   =  2.0 * pow(t, 6) -   6.0 * pow(t, 5) +  5.0 * pow(t, 4) - 1.0 * pow(t, 2) + shift;
  *Pt  = 12.0 * pow(t, 5) -  30.0 * pow(t, 4) + 20.0 * pow(t, 3) - 2.0 * t;
  *Ptt = 60.0 * pow(t, 4) - 120.0 * pow(t, 3) + 60.0 * pow(t, 2) - 2.0;
}
/* problem.fv.c:9-28,71-87,90-140: 4th-order cell/face averages = point value + h^2/24 * second derivatives */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
static double fv_beta(double x, double y, double z, double h, int add_Bxx, int add_Byy, int add_Bzz) {
  const double b = 0.25, a = 2.0 * M_PI;
  double B   = 1.0 + b * sin(a * x) * sin(a * y) * sin(a * z);
  double Bxx = -a * a * b * sin(a * x) * sin(a * y) * sin(a * z);
  double Byy = -a * a * b * sin(a * x) * sin(a * y) * sin(a * z);
  double Bzz = -a * a * b * sin(a * x) * sin(a * y) * sin(a * z);
  if (add_Bxx) B += (h * h / 24.0) * Bxx;
  if (add_Byy) B += (h * h / 24.0) * Byy;
  if (add_Bzz) B += (h * h / 24.0) * Bzz;
  return B;
}
static double fv_F(double x, double y, double z, double h) {
  const double a = 2.0 * M_PI, p = 7.0;
  double F   = pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p);
  double Fxx = -a * a * p * pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p) + a * a * p * (p - 1) * pow(sin(a * x), p - 2) * pow(sin(a * y), p) * pow(sin(a * z), p) * pow(cos(a * x), 2);
  double Fyy = -a * a * p * pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p) + a * a * p * (p - 1) * pow(sin(a * x), p) * pow(sin(a * y), p - 2) * pow(sin(a * z), p) * pow(cos(a * y), 2);
  double Fzz = -a * a * p * pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p) + a * a * p * (p - 1) * pow(sin(a * x), p) * pow(sin(a * y), p) * pow(sin(a * z), p - 2) * pow(cos(a * z), 2);
  F += (h * h / 24.0) * Fxx;
  F += (h * h / 24.0) * Fyy;
  F += (h * h / 24.0) * Fzz;
  return F;
}
static void initialize_problem_fv(level_type *L, double h, const hpgmg_config *cfg) {
  L->h = h;
  const int jS = L->box_jStride, kS = L->box_kStride, g = L->box_ghosts, dim = L->box_dim;
  const size_t vol = (size_t)L->box_volume;
  double *stage = (double *)calloc(5 * vol, sizeof(double));
  int box, i, j, k;
  for (box = 0; box < L->num_my_boxes; box++) {
    const box_type *B = &L->my_boxes[box];
    memset(stage, 0, 5 * vol * sizeof(double));
    #pragma omp parallel for private(k, j, i) collapse(2)
    for (k = 0; k <= dim; k++) for (j = 0; j <= dim; j++) for (i = 0; i <= dim; i++) {
      const size_t ijk = (size_t)(i + g) + (size_t)(j + g) * jS + (size_t)(k + g) * kS;
      const double x = h * ((double)(i + B->low.i) + 0.5), y = h * ((double)(j + B->low.j) + 0.5), z = h * ((double)(k + B->low.k) + 0.5);
      double Bi = 1.0, Bj = 1.0, Bk = 1.0;
      if (cfg->variable_coeff) {
        Bi = fv_beta(x - h * 0.5, y, z, h, 0, 1, 1);
        Bj = fv_beta(x, y - h * 0.5, z, h, 1, 0, 1);
        Bk = fv_beta(x, y, z - h * 0.5, h, 1, 1, 0);
      }
      stage[0 * vol + ijk] = Bi; stage[1 * vol + ijk] = Bj; stage[2 * vol + ijk] = Bk; stage[3 * vol + ijk] = 1.0;
      stage[4 * vol + ijk] = fv_F(x, y, z, h);
    }
    hpgmg_vector_upload(B->vectors[VECTOR_BETA_I], stage + 0 * vol, vol);
    hpgmg_vector_upload(B->vectors[VECTOR_BETA_J], stage + 1 * vol, vol);
    hpgmg_vector_upload(B->vectors[VECTOR_BETA_K], stage + 2 * vol, vol);
    if (cfg->helmholtz) hpgmg_vector_upload(B->vectors[VECTOR_ALPHA], stage + 3 * vol, vol);
    hpgmg_vector_upload(B->vectors[VECTOR_F], stage + 4 * vol, vol);
  }
  free(stage);
}

void initialize_problem(level_type *L, double h, double a, double b) {
  hp_coef32_invalidate(L);
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (cfg.op == HPGMG_OP_FV2 || cfg.op == HPGMG_OP_FV4) { initialize_problem_fv(L, h, &cfg); return; }
  L->h = h;
  const int jS = L->box_jStride, kS = L->box_kStride, g = L->box_ghosts, dim = L->box_dim;
  const size_t vol = (size_t)L->box_volume;
  const double shift = (L->boundary_condition.type == BC_PERIODIC) ? 1.0 / 21.0 : 0.0;
  double *stage = (double *)calloc(5 * vol, sizeof(double)); /* beta_i, beta_j, beta_k, alpha, F */
  int box, i, j, k;
  for (box = 0; box < L->num_my_boxes; box++) {
    const box_type *B = &L->my_boxes[box];
    memset(stage, 0, 5 * vol * sizeof(double));
    #pragma omp parallel for private(k, j, i) collapse(2)
    for (k = 0; k <= dim; k++) for (j = 0; j <= dim; j++) for (i = 0; i <= dim; i++) {   /* <= : high faces too */
      const size_t ijk = (size_t)(i + g) + (size_t)(j + g) * jS + (size_t)(k + g) * kS;
      const double x = h * ((double)(i + B->low.i) + 0.5), y = h * ((double)(j + B->low.j) + 0.5), z = h * ((double)(k + B->low.k) + 0.5);
      double A = 1.0, Bc = 1.0, Bx = 0.0, By = 0.0, Bz = 0.0, Bi = 1.0, Bj = 1.0, Bk = 1.0;
      if (cfg.variable_coeff) {
        eval_beta(x - h * 0.5, y, z, &Bi, &Bx, &By, &Bz);
        eval_beta(x, y - h * 0.5, z, &Bj, &Bx, &By, &Bz);
        eval_beta(x, y, z - h * 0.5, &Bk, &Bx, &By, &Bz);
        eval_beta(x, y, z, &Bc, &Bx, &By, &Bz);
      }
      double X, Xx, Xxx, Y, Yy, Yyy, Z, Zz, Zzz;
      eval_poly(x, shift, &X, &Xx, &Xxx); eval_poly(y, shift, &Y, &Yy, &Yyy); eval_poly(z, shift, &Z, &Zz, &Zzz);
      const double U = X * Y * Z, Ux = Xx * Y * Z, Uy = X * Yy * Z, Uz = X * Y * Zz, Uxx = Xxx * Y * Z, Uyy = X * Yyy * Z, Uzz = X * Y * Zzz;
      stage[0 * vol + ijk] = Bi; stage[1 * vol + ijk] = Bj; stage[2 * vol + ijk] = Bk; stage[3 * vol + ijk] = A;
      stage[4 * vol + ijk] = a * A * U - b * ((Bx * Ux + By * Uy + Bz * Uz) + Bc * (Uxx + Uyy + Uzz));
    }
    hpgmg_vector_upload(B->vectors[VECTOR_BETA_I], stage + 0 * vol, vol);
    hpgmg_vector_upload(B->vectors[VECTOR_BETA_J], stage + 1 * vol, vol);
    hpgmg_vector_upload(B->vectors[VECTOR_BETA_K], stage + 2 * vol, vol);
    if (cfg.helmholtz) hpgmg_vector_upload(B->vectors[VECTOR_ALPHA], stage + 3 * vol, vol);
    hpgmg_vector_upload(B->vectors[VECTOR_F], stage + 4 * vol, vol);
  }
  free(stage);
}

/* ---------------------------------------------------------------- operators.7pt.c:95-252 */
void rebuild_operator(level_type *L, level_type *from, double a, double b) {
  hp_coef32_invalidate(L);
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (cfg.op != HPGMG_OP_7PT) {                                 /* operators.27pt.c:96-121, .fv2.c:98-124, .fv4.c:145-172 */
    if (from) {
      if (cfg.helmholtz) hp_do_restriction(L, VECTOR_ALPHA, from, VECTOR_ALPHA, RESTRICT_CELL);
      hp_do_restriction(L, VECTOR_BETA_I, from, VECTOR_BETA_I, RESTRICT_FACE_I);
      hp_do_restriction(L, VECTOR_BETA_J, from, VECTOR_BETA_J, RESTRICT_FACE_J);
      hp_do_restriction(L, VECTOR_BETA_K, from, VECTOR_BETA_K, RESTRICT_FACE_K);
    }
    if (cfg.op == HPGMG_OP_FV4) extrapolate_betas(L);           /* mixed-derivative terms read beta in the ghost zone */
    if (cfg.helmholtz) exchange_boundary(L, VECTOR_ALPHA, STENCIL_SHAPE_BOX);
    exchange_boundary(L, VECTOR_BETA_I, STENCIL_SHAPE_BOX);
    exchange_boundary(L, VECTOR_BETA_J, STENCIL_SHAPE_BOX);
    exchange_boundary(L, VECTOR_BETA_K, STENCIL_SHAPE_BOX);
    rebuild_operator_blackbox(L, a, b, cfg.op == HPGMG_OP_FV4 ? 4 : 2);
    exchange_boundary(L, VECTOR_DINV, STENCIL_SHAPE_BOX);
    return;
  }
  if (cfg.op != HPGMG_OP_7PT) hp_no_kernel("rebuild_operator for this operator");
  if (L->my_rank == 0 && hpgmg_verbose) { fprintf(stdout, "  rebuilding operator for level...  h=%e  ", L->h); fflush(stdout); }
  if (from) {
    if (cfg.helmholtz) hp_do_restriction(L, VECTOR_ALPHA, from, VECTOR_ALPHA, RESTRICT_CELL);
    hp_do_restriction(L, VECTOR_BETA_I, from, VECTOR_BETA_I, RESTRICT_FACE_I);
    hp_do_restriction(L, VECTOR_BETA_J, from, VECTOR_BETA_J, RESTRICT_FACE_J);
    hp_do_restriction(L, VECTOR_BETA_K, from, VECTOR_BETA_K, RESTRICT_FACE_K);
  }
  if (cfg.helmholtz) exchange_boundary(L, VECTOR_ALPHA, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_I, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_J, STENCIL_SHAPE_BOX);
  exchange_boundary(L, VECTOR_BETA_K, STENCIL_SHAPE_BOX);

  double lambda = -1e9;
  BLAS1(hpgmg_hip_rebuild_7pt(&hp_backend_of(L)->dev, cfg.variable_coeff, cfg.helmholtz ? VECTOR_ALPHA : -1,
                              cfg.helmholtz ? VECTOR_L1INV : -1, a, b, 1.0 / (L->h * L->h), &lambda));
  { const hpgmg_transport *T = hpgmg_get_transport();
    if (T && T->size > 1) { int r, *all = (int *)malloc((size_t)T->size * sizeof(int)); for (r = 0; r < T->size; r++) all[r] = r;
      T->allreduce(T->ctx, &lambda, 1, HPGMG_REDUCE_MAX, all, T->size); free(all); } }
  if (L->my_rank == 0 && hpgmg_verbose) fprintf(stdout, "eigenvalue_max<%e\n", lambda);
  L->dominant_eigenvalue_of_DinvA = lambda;
  exchange_boundary(L, VECTOR_DINV, STENCIL_SHAPE_BOX);
  if (cfg.helmholtz) exchange_boundary(L, VECTOR_L1INV, STENCIL_SHAPE_BOX);
}

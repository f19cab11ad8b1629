/*
 * solvers.c -- bottom (coarsest level) solver: diagonally preconditioned
 * BiCGStab (default) or CG, driven from the host through operators.h only.
 *
 * Behavioural reference: finite-volume/source/solvers.c:27-95 and
 * solvers/bicgstab.c:14-97 (Saad, Iterative Methods, Alg. 7.7 with a right
 * preconditioner M = D).  The sequence of operator calls and the break-down
 * tests are the same, because the number of bottom iterations per solve and
 * the coarse correction they produce feed the pinned F-cycle norms.
 * north_star: "BiCGStab on the coarse bottom stays on host" -- the control flow
 * and every scalar live here; vectors stay wherever the plugin keeps them.
 */
#include <math.h>
#include "hpgmg_level.h"
#include "hpgmg_operators.h"
#include "hpgmg_mg.h"

int IterativeSolver_NumVectors(void) { return hpgmg_get_bottom_solver() == HPGMG_BOTTOM_CG ? 5 : 8; } /* BiCGStab: r0, r, p, q, s, t, Ap, As; CG: r0, r, p, Ap, z (solvers.c:92-104) */

static void remove_mean(level_type *L, int id) {
  if (L->must_subtract_mean == 1) {
    double m = mean(L, id);
    shift_vector(L, id, id, -m);
  }
}

static void bicgstab(level_type *L, int x_id, int R_id, double a, double b, double want) {
  const int base = hpgmg_vectors_reserved();
  const int r0 = base + 0, r = base + 1, p = base + 2, q = base + 3, s = base + 4, t = base + 5, Ap = base + 6, As = base + 7;
  const int max_iters = 200;
  int it = 0;

  residual(L, r0, x_id, R_id, a, b);
  remove_mean(L, r0);
  scale_vector(L, r, 1.0, r0);
  scale_vector(L, p, 1.0, r0);
  double rho = dot(L, r, r0);
  const double r0_norm = norm(L, r);
  if (rho == 0.0 || r0_norm == 0.0) return; /* entered with the exact solution */

  while (it < max_iters) {
    it++;
    L->Krylov_iterations++;
    mul_vectors(L, q, 1.0, VECTOR_DINV, p);              /* q = M^-1 p */
    apply_op(L, Ap, q, a, b);
    double Ap_r0 = dot(L, Ap, r0);
    if (Ap_r0 == 0.0) break;                              /* pivot breakdown */
    double alpha = rho / Ap_r0;
    if (isinf(alpha)) break;
    add_vectors(L, x_id, 1.0, x_id, alpha, q);
    add_vectors(L, s, 1.0, r, -alpha, Ap);                /* s = r - alpha A q */
    remove_mean(L, s);
    double s_norm = norm(L, s);
    if (s_norm == 0.0 || s_norm < want * r0_norm) break;  /* converged on the half step */
    mul_vectors(L, t, 1.0, VECTOR_DINV, s);              /* t = M^-1 s */
    apply_op(L, As, t, a, b);
    double As_As = dot(L, As, As);
    double As_s  = dot(L, As, s);
    if (As_As == 0.0) break;
    double omega = As_s / As_As;
    if (omega == 0.0 || isinf(omega)) break;              /* stabilisation breakdown */
    add_vectors(L, x_id, 1.0, x_id, omega, t);
    add_vectors(L, r, 1.0, s, -omega, As);
    remove_mean(L, r);
    double r_norm = norm(L, r);
    if (r_norm == 0.0 || r_norm < want * r0_norm) break;
    double rho_new = dot(L, r, r0);
    if (rho_new == 0.0) break;                            /* Lanczos breakdown */
    double beta = (rho_new / rho) * (alpha / omega);
    if (isinf(beta)) break;
    add_vectors(L, VECTOR_TEMP, 1.0, p, -omega, Ap);
    add_vectors(L, p, 1.0, r, beta, VECTOR_TEMP);         /* p = r + beta (p - omega Ap) */
    rho = rho_new;
  }
}

/* The reference's other host-driven choice, -DUSE_CG (solvers/cg.c:14-77; Saad, algorithm 9.1 with the diagonal as preconditioner): same calls,
 * same order, same break-down tests -- the iteration count and the correction feed the pinned norms exactly as BiCGStab's do. */
static void cg(level_type *L, int x_id, int R_id, double a, double b, double want) {
  const int base = hpgmg_vectors_reserved();
  const int r0 = base + 0, r = base + 1, p = base + 2, Ap = base + 3, z = base + 4;
  const int max_iters = 200;
  int it = 0;
  residual(L, r0, x_id, R_id, a, b);
  remove_mean(L, r0);
  scale_vector(L, r, 1.0, r0);
  mul_vectors(L, z, 1.0, VECTOR_DINV, r0);               /* z = D^-1 r0 */
  scale_vector(L, p, 1.0, z);
  const double r0_norm = norm(L, r);
  if (r0_norm == 0.0) return;                             /* entered with the exact solution */
  double r_dot_z = dot(L, r, z);
  while (it < max_iters) {
    it++;
    L->Krylov_iterations++;
    apply_op(L, Ap, p, a, b);
    const double Ap_p = dot(L, Ap, p);
    if (Ap_p == 0.0) break;                               /* pivot breakdown */
    const double alpha = r_dot_z / Ap_p;
    if (isinf(alpha)) break;
    add_vectors(L, x_id, 1.0, x_id, alpha, p);
    add_vectors(L, r, 1.0, r, -alpha, Ap);
    remove_mean(L, r);
    const double r_norm = norm(L, r);
    if (r_norm == 0.0 || r_norm < want * r0_norm) break;
    mul_vectors(L, z, 1.0, VECTOR_DINV, r);
    const double r_dot_z_new = dot(L, r, z);
    if (r_dot_z_new == 0.0) break;                        /* Lanczos breakdown */
    const double beta = r_dot_z_new / r_dot_z;
    if (isinf(beta)) break;
    add_vectors(L, p, 1.0, z, beta, p);
    r_dot_z = r_dot_z_new;
  }
}

void IterativeSolver(level_type *L, int u_id, int f_id, double a, double b, double desired_reduction_in_norm) {
  if (!L->active) return;
  if (L->must_subtract_mean == -1) {
    int alpha_is_zero = 1;
    L->must_subtract_mean = 0;
    if (hpgmg_vectors_reserved() > VECTOR_ALPHA) alpha_is_zero = (dot(L, VECTOR_ALPHA, VECTOR_ALPHA) == 0.0);
    if (L->boundary_condition.type == BC_PERIODIC && (a == 0 || alpha_is_zero)) L->must_subtract_mean = 1;
  }
  if (hpgmg_get_bottom_solver() == HPGMG_BOTTOM_CG) { cg(L, u_id, f_id, a, b, desired_reduction_in_norm); return; }
  if (L->must_subtract_mean != 1 && hpgmg_bottom_solve_fused(L, u_id, f_id, a, b, desired_reduction_in_norm)) return;   /* the same solver as one device launch */
  bicgstab(L, u_id, f_id, a, b, desired_reduction_in_norm);
}

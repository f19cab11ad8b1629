/*
 * hpgmg_operators.h -- the operator plugin surface (THE drop-in boundary).
 *
 * Every prototype below has the same name, argument order and meaning as the
 * reference's finite-volume/source/operators.h:14-50.  The reference selects
 * one plugin (operators.7pt.c, .27pt.c, .fv4.c ...) and one smoother at compile
 * time with -D flags; here the same choices are a process-wide runtime
 * configuration (hpgmg_configure) because the reference's own
 * stencil_get_radius()/stencil_get_shape() take no level argument.
 *
 * Two implementations of this header exist in the repository:
 *   hpgmg_amd/csrc/host/operators_hip.c  -- product: forwards to the HIP kernels
 *                                           behind include/hpgmg_hip.h
 *   oracle/operators_cpu.c               -- test oracle: plain C restatement
 * They are never linked into the same binary.
 */
#ifndef HPGMG_OPERATORS_H
#define HPGMG_OPERATORS_H

#include "hpgmg_level.h"

#ifdef __cplusplus
extern "C" {
#endif

/* vector ids (reference defines.h:12-39).  ALPHA/L1INV exist only when the
 * configuration is Helmholtz, exactly like -DUSE_HELMHOLTZ in the reference. */
#define VECTOR_TEMP    0
#define VECTOR_U       1
#define VECTOR_F       2
#define VECTOR_E       3
#define VECTOR_R       4
#define VECTOR_DINV    5
#define VECTOR_BETA_I  6
#define VECTOR_BETA_J  7
#define VECTOR_BETA_K  8
#define VECTOR_ALPHA   9
#define VECTOR_L1INV  10

/* ---- runtime replacement of the reference's compile-time -D switches ---- */
enum { HPGMG_OP_7PT = 0, HPGMG_OP_27PT = 1, HPGMG_OP_FV4 = 2, HPGMG_OP_FV2 = 3 };
enum { HPGMG_SMOOTH_CHEBY = 0, HPGMG_SMOOTH_GSRB = 1, HPGMG_SMOOTH_JACOBI = 2 };

typedef struct {
  int op;                 /* HPGMG_OP_*          (which operators.X.c)             */
  int smoother;           /* HPGMG_SMOOTH_*      (-DUSE_CHEBY / -DUSE_GSRB / ...)   */
  int helmholtz;          /* 1 = -DUSE_HELMHOLTZ (adds VECTOR_ALPHA, VECTOR_L1INV) */
  int variable_coeff;     /* 1 = STENCIL_VARIABLE_COEFFICIENT (7pt/fv2/fv4)        */
} hpgmg_config;

/* Returns 0, or -1 for a combination the reference itself rejects with #error
 * (e.g. 27pt + variable coefficients, operators.27pt.c:53-55). */
int  hpgmg_configure(const hpgmg_config *cfg);
void hpgmg_get_config(hpgmg_config *cfg);
int  hpgmg_vectors_reserved(void); /* VECTORS_RESERVED: 9, or 11 for Helmholtz */
enum { HPGMG_BOTTOM_BICGSTAB = 0, HPGMG_BOTTOM_CG = 1 };      /* the reference's -DUSE_BICGSTAB / -DUSE_CG (solvers.c:17-24): host loops over the operators */
void hpgmg_set_bottom_solver(int which);   /* before MGBuild (the solver's work vectors are created there: 8 / 5) */
int  hpgmg_get_bottom_solver(void);

/* ---- operators.h:14-15 ---- */
int stencil_get_radius(void);
int stencil_get_shape(void);
/* ---- operators.h:17-21 ---- */
void apply_op(level_type *level, int Ax_id, int x_id, double a, double b);
void residual(level_type *level, int res_id, int x_id, int rhs_id, double a, double b);
void smooth(level_type *level, int phi_id, int rhs_id, double a, double b);
void rebuild_operator(level_type *level, level_type *fromLevel, double a, double b);
void rebuild_operator_blackbox(level_type *level, double a, double b, int colors_in_each_dim);
/* ---- operators.h:23-25 ---- */
void restriction(level_type *level_c, int id_c, level_type *level_f, int id_f, int restrictionType);
void interpolation_vcycle(level_type *level_f, int id_f, double prescale_f, level_type *level_c, int id_c);
void interpolation_fcycle(level_type *level_f, int id_f, double prescale_f, level_type *level_c, int id_c);
/* ---- operators.h:27-33 ---- */
void exchange_boundary(level_type *level, int id_a, int shape);
void apply_BCs(level_type *level, int x_id, int shape);   /* plugin's dispatch, operators.7pt.c:47 */
void apply_BCs_p1(level_type *level, int x_id, int shape);
void apply_BCs_p2(level_type *level, int x_id, int shape);
void apply_BCs_v1(level_type *level, int x_id, int shape);
void apply_BCs_v2(level_type *level, int x_id, int shape);
void apply_BCs_v4(level_type *level, int x_id, int shape);
void extrapolate_betas(level_type *level);
/* ---- operators.h:35-45 ---- */
double dot(level_type *level, int id_a, int id_b);
double norm(level_type *level, int id_a);
double mean(level_type *level, int id_a);
double error(level_type *level, int id_a, int id_b);
void add_vectors(level_type *level, int id_c, double scale_a, int id_a, double scale_b, int id_b);
void scale_vector(level_type *level, int id_c, double scale_a, int id_a);
void zero_vector(level_type *level, int id_a);
void shift_vector(level_type *level, int id_c, int id_a, double shift_a);
void mul_vectors(level_type *level, int id_c, double scale, int id_a, int id_b);
void invert_vector(level_type *level, int id_c, double scale_a, int id_a);
void init_vector(level_type *level, int id_a, double scalar);
/* ---- operators.h:47-48 ---- */
void color_vector(level_type *level, int id, int colors, int icolor, int jcolor, int kcolor);
void random_vector(level_type *level, int id);
/* ---- operators.h:50 ---- */
void initialize_problem(level_type *level, double hLevel, double a, double b);

/* ---- the one hook outside operators.h: who owns vector storage -----------
 * The reference allocates with MALLOC() in level.c:25-40 and zero-fills on the
 * host (level.c:976).  The plugin supplies these instead, so level.c never
 * touches vector bytes: HIP build -> hipMalloc/hipMemset, oracle -> calloc. */
double *hpgmg_vector_alloc(size_t num_doubles);           /* zero-filled */
void    hpgmg_vector_free(double *p);
void    hpgmg_vector_copy(double *dst, const double *src, size_t num_doubles);
/* host<->plugin staging, used by tests and by initialize_problem */
void    hpgmg_vector_upload(double *dst_plugin, const double *src_host, size_t num_doubles);
void    hpgmg_vector_download(double *dst_host, const double *src_plugin, size_t num_doubles);
/* Launch-bound stretches of a cycle (everything done on levels of <= 64^3 cells between two
 * bottom solves) are bracketed by the cycle driver as a SEGMENT with a key that repeats every
 * solve, so the HIP plugin can capture it once into a hipGraph and replay it.  Plugins without
 * such a mechanism implement these as no-ops.  A reduction (norm/dot/mean) ends an open segment. */
void    hpgmg_segment_begin(long long key);
void    hpgmg_segment_end(void);
/* Optional fused form of MGVCycle (mg.c:1147-1163) over the chain levels[0..n-1] (finest first, levels[n-1] = bottom level).  `leg` names the part
 * (enum hpgmg_leg).  Returns 1 when the plugin executed it (bit-identical to the per-operator sequence), 0 when it cannot -- the driver then issues the
 * operators one by one. */
enum hpgmg_leg {
  HPGMG_LEG_DOWN = 0,              /* smooth, residual, restriction, zero_vector per level going down (the driver runs IterativeSolver on the bottom level afterwards) */
  HPGMG_LEG_UP = 1,                /* interpolation_vcycle, smooth per level going up */
  HPGMG_LEG_VCYCLE = 2,            /* DOWN, the bottom solve (solvers.c:27-95, BiCGStab to MG_DEFAULT_BOTTOM_NORM), UP */
  HPGMG_LEG_BOTTOM = 3,            /* the bottom solve alone (n == 1) */
  HPGMG_LEG_FCYCLE_TAIL = 4,       /* the part of FMGSolve below levels[0] (mg.c:1270-1300): restriction of R down the chain, zero_vector + bottom solve, then per
                                      level upwards interpolation_fcycle and MGVCycle, levels[0] included */
  HPGMG_LEG_FCYCLE_TAIL_ASK = 5,   /* only ask whether HPGMG_LEG_FCYCLE_TAIL would be executed (nothing runs) */
  HPGMG_LEG_FCYCLE_STEP = 6,       /* one step of FMGSolve's climb (mg.c:1289-1293): interpolation_fcycle(levels[0] <- levels[1]) and the V-cycle from levels[0] */
  HPGMG_LEG_ASK = 16               /* added to a leg: only ask whether it would be executed */
};
int     hpgmg_vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg);
/* A plugin whose fused launches can fail AS A WHOLE (the HIP plugin's brick launches, when not all their workgroups get to run: kernels/brick_visit.hip) lets
 * the cycle driver bracket a solve it is able to repeat from its inputs: begin() = "should such a launch fail from here on, do not stop at the next scalar";
 * end() returns 1 when one did fail -- the plugin has then switched that form off, and the driver repeats the solve (host/mg.c FMGSolve).  Outside such a
 * bracket a failure stops the program with a message at the next scalar, as before.  The CPU oracle: no-ops returning 0. */
void    hpgmg_solve_attempt_begin(void);
int     hpgmg_solve_attempt_end(void);
/* The whole bottom solve -- IterativeSolver's BiCGStab (solvers/bicgstab.c:14-97) on level L: e_id = initial guess and solution -- as one device
 * launch where the plugin has one (27-point / fv2 / fv4: a bottom level of one small box, Dirichlet); 0 = not taken, the caller runs the
 * host-driven solver.  Same iterates, same iteration count (folded into L->Krylov_iterations by hpgmg_level_sync_counters). */
int     hpgmg_bottom_solve_fused(level_type *L, int e_id, int R_id, double a, double b, double desired_reduction_in_norm);
/* Optional fused form of  interpolation_vcycle(fine, e, 1.0, coarse, e); smooth(fine, e, R)  (mg.c:1160-1161): returns 1 when the
 * plugin executed both (same iterate; VECTOR_TEMP unspecified, as with hpgmg_smooth_in_cycle), 0 when the driver must call the two operators. */
int     hpgmg_interp_smooth_fused(level_type *fine, int e_id, int R_id, level_type *coarse, double a, double b);
/* Optional fused form of  restriction(coarse, id_c, fine, id_f, RESTRICT_CELL); zero_vector(coarse, zero_id)  (mg.c:1152-1153) */
int     hpgmg_restrict_zero_fused(level_type *coarse, int id_c, level_type *fine, int id_f, int zero_id);
/* Timing hooks for the cycle driver's per-level "Total" rows (mg.c:54-161).  The plugin decides what a tick measures: the
 * CPU oracle reads the host clock; the HIP plugin, whose launches are asynchronous, can record a hipEvent pair instead and
 * add the elapsed DEVICE time to *acc later -- hpgmg_timers_settle() (also done by hpgmg_level_sync_counters) makes every
 * pending tick land in its accumulator.  `what` names the range for profilers (roctx). */
typedef struct { double t0; double *acc; int slot, range; } hpgmg_tick;
hpgmg_tick hpgmg_tick_begin(level_type *level, double *acc_seconds, const char *what);
void    hpgmg_tick_end(hpgmg_tick t);
void    hpgmg_timers_settle(void);
/* 0 host clock around (asynchronous) operator calls, 1 device time per operator (hipEvent pairs), 2 synchronise around every
 * operator; the CPU oracle ignores it.  Environment: HPGMG_TIMERS=host|device|sync. */
void    hpgmg_set_timer_mode(int mode);
int     hpgmg_get_timer_mode(void);
/* Optional fused forms around residual() (return 1 when executed, 0 when the driver must issue the operators one by one):
 *   residual(fine, TEMP, x, rhs); restriction(coarse, id_c, fine, TEMP, RESTRICT_CELL); zero_vector(coarse, zero_id)   (mg.c:1150-1153)
 *     -- same coarse result; the fine level's VECTOR_TEMP is left untouched (the residual is never stored);
 *   residual(level, res, x, rhs); *norm_out = norm(level, res)                                                          (mg.c:1321-1323)
 *     -- res_id < 0: only the norm is wanted (the cycle driver's convergence check: nothing reads the residual afterwards) */
int     hpgmg_residual_restrict_zero_fused(level_type *coarse, int id_c, level_type *fine, int x_id, int rhs_id, double a, double b, int zero_id);
int     hpgmg_residual_norm_fused(level_type *level, int res_id, int x_id, int rhs_id, double a, double b, double *norm_out);
/*   *norm_out = norm(level, F); scale_vector(level, R, 1.0, F); restriction(coarse, R, level, R, RESTRICT_CELL)                        (mg.c:1262-1270) */
int     hpgmg_norm_scale_restrict_fused(level_type *level, int F_id, int R_id, level_type *coarse, double *norm_out);
/*   the same with the norm collected LATER by hpgmg_norm_deferred_fetch(level) (once, before anything else defers): FMGSolve uses norm(F) only in the check at its end */
int     hpgmg_norm_scale_restrict_fused_deferred(level_type *level, int F_id, int R_id, level_type *coarse);
double  hpgmg_norm_deferred_fetch(level_type *level);
/*   zero_vector(fine, id_f); interpolation_fcycle(fine, id_f, 0.0, coarse, id_c) -- the benchmark step's zero_vector(u) (hpgmg-fv.c:77-85) and the F-cycle's
 *   first write of u on that level (mg.c:1295) -- as one launch: same interior; the ghost zones zero_vector would clear may keep their content */
int     hpgmg_zero_interpolation_fcycle_fused(level_type *fine, int id_f, level_type *coarse, int id_c);
/* smooth() for callers to whom VECTOR_TEMP is scratch afterwards (MGVCycle: the operator that follows a smooth() overwrites or ignores
 * it): same iterate in phi_id, VECTOR_TEMP unspecified.  Returns 1 when executed, 0 when the caller must call smooth(). */
int     hpgmg_smooth_in_cycle(level_type *level, int phi_id, int rhs_id, double a, double b);
/* The operators that return nothing may be postponed by a plugin: the HIP plugin records smooth / residual / restriction / zero_vector /
 * interpolation_vcycle while they follow the order MGVCycle issues them in (mg.c:1145-1164) and runs them, fused where it can, at the first
 * call that does not -- so the reference's unmodified driver gets the fused forms too (INTEGRATION.md Route B).  The state every later call
 * sees is exactly the one the separate operators leave.  hpgmg_operators_flush() issues what is pending (tests that count launches use it),
 * hpgmg_set_lazy(0) / HPGMG_LAZY=0 turns the queue off.  The CPU oracle implements both as no-ops. */
void    hpgmg_operators_flush(void);
void    hpgmg_set_lazy(int on);
/* bring level->Krylov_iterations up to date with bottom solves the plugin ran asynchronously */
void    hpgmg_level_sync_counters(level_type *level);
/* called by destroy_level / MGDestroy so the plugin can drop device mirrors */
void    hpgmg_level_release(level_type *level);
const char *hpgmg_backend_name(void);                      /* "hip" or "oracle-cpu" */

#ifdef __cplusplus
}
#endif
#endif

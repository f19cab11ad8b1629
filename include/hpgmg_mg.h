/*
 * hpgmg_mg.h -- multigrid hierarchy + cycle driver + bottom solver + transport.
 *
 * Callers of the operator plugin.  Same entry points as the reference's
 * mg.h:22-45 and solvers.h:8-9; the control flow of FMGSolve/MGVCycle follows
 * reference mg.c:1135-1344 call for call, because the order of operator calls
 * is what the golden residual norms pin.
 */
#ifndef HPGMG_MG_H
#define HPGMG_MG_H

#include "hpgmg_level.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MG_AGGLOMERATION_START 8     /* reference mg.h:15-16 */
#define MG_DEFAULT_BOTTOM_NORM 1e-3  /* reference mg.h:18-19 */

typedef struct {
  int my_rank;
  int num_levels;
  level_type **levels;
  struct { double MGBuild, MGSolve; } timers;
  int MGSolves_performed;
} mg_type;

void MGBuild(mg_type *all_grids, level_type *fine_grid, double a, double b, int minCoarseGridDim);
void MGDestroy(mg_type *all_grids);
void MGVCycle(mg_type *all_grids, int e_id, int R_id, double a, double b, int level);
void MGSolve(mg_type *all_grids, int onLevel, int u_id, int F_id, double a, double b, double rtol);
void FMGSolve(mg_type *all_grids, int onLevel, int u_id, int F_id, double a, double b, double rtol);
void hpgmg_set_ucycles(int on);        /* before MGBuild: 1 = the reference's -DUSE_UCYCLES ladder (boxes halved, never merged; mg.c:878-893) */
void hpgmg_fmg_zero_u_first(void);  /* the NEXT FMGSolve starts from u = 0: it zeroes u itself, where it first touches it (the benchmark step's zero_vector(u), hpgmg-fv.c:77-85) */
void hpgmg_set_fmg_vcycles(int n);     /* V-cycles FMGSolve may add after its F-cycle: 0 (default) or 20 = -DUNLIMIT_FMG_ITERATIONS (mg.c:1239-1247) */
void MGPCG(mg_type *all_grids, int onLevel, int x_id, int F_id, double a, double b, double rtol);   /* mg.c:1500-1605: CG preconditioned with one V-cycle per iteration; grows every level by three vectors */
void MGPrintTiming(mg_type *all_grids, int fromLevel);
void MGResetTimers(mg_type *all_grids);
void richardson_error(mg_type *all_grids, int levelh, int u_id);

/* bottom solver (reference solvers.h / solvers/bicgstab.c) -- host control flow */
void IterativeSolver(level_type *level, int u_id, int f_id, double a, double b, double desired_reduction_in_norm);
int  IterativeSolver_NumVectors(void);

/* last values computed by FMGSolve/MGSolve/richardson_error -- what the
 * reference only prints (mg.c:1328, :1128, :1130); kept so tests can read them. */
typedef struct {
  double norm_of_F, norm_of_residual;   /* after the F-cycle (or last V-cycle) */
  double richardson_error, richardson_order;
  int    vcycles;
} hpgmg_solve_record;
extern hpgmg_solve_record hpgmg_last_solve;
extern int hpgmg_verbose;  /* 1 = print exactly what the reference prints (default), 0 = silent */

/* ---- transport: what replaces MPI (reference SURVEY 2.3) --------------------
 * One process per GPU.  Buffers handed to sendrecv live in plugin memory
 * (device pointers in the HIP build).  A NULL transport means one rank. */
enum { HPGMG_REDUCE_MAX = 0, HPGMG_REDUCE_SUM = 1 };
typedef struct {
  int   rank, size;
  void *ctx;
  /* post every receive and every send of one exchange phase and complete them
   * (MPI_Irecv* / MPI_Isend* / MPI_Waitall of reference exchange_boundary.c:33-97) */
  void (*sendrecv)(void *ctx,
                   int nrecv, double *const *rbuf, const int *rsize, const int *rrank,
                   int nsend, double *const *sbuf, const int *ssize, const int *srank, int tag);
  /* in-place allreduce of n host doubles over the listed ranks (sorted, includes caller) */
  void (*allreduce)(void *ctx, double *vals, int n, int op, const int *ranks, int nranks);
  /* optional (NULL: not wanted).  MGBuild calls it on EVERY rank of the job, in the same order, once per level whose reductions run over a proper
   * subset of the ranks (the reference's MPI_Comm_split per level, mg.c:985-993): a transport that can make a sub-communicator for the set does it
   * here -- building one is collective over the whole job (ncclCommSplit), which allreduce(), called by the members only, cannot be. */
  void (*prepare_subset)(void *ctx, const int *ranks, int nranks);
} hpgmg_transport;
void hpgmg_set_transport(const hpgmg_transport *t);
const hpgmg_transport *hpgmg_get_transport(void);

/* per-level side record (never stored inside level_type, see hpgmg_level.h) */
typedef struct hpgmg_level_ext {
  level_type *level;
  int  *active_ranks;      /* ranks that take part in reductions on this level */
  int   num_active_ranks;
  void *backend;           /* plugin-private (device mirrors of the block lists ...) */
  double *slab;            /* single allocation holding every owned box, or NULL */
  size_t  slab_doubles;
  struct hpgmg_level_ext *next;
} hpgmg_level_ext;
hpgmg_level_ext *hpgmg_level_ext_get(level_type *level);   /* creates on first use */
void             hpgmg_level_ext_drop(level_type *level);

#ifdef __cplusplus
}
#endif
#endif

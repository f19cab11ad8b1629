/*
 * hpgmg_fv.h -- C-ABI of the host driver library (libhpgmg_fv.so / liboracle_fv.so).
 *
 * Plain pointers, ints and doubles only; this is what ctypes (tests, bench.py)
 * and any non-C host binds.  It wraps what the reference does in
 * finite-volume/source/hpgmg-fv.c: main() :103-386 (problem-size selection
 * :152-205, level creation :283-295, MGBuild :308, the timed loop :320-345,
 * Richardson analysis :351-366) and bench_hpgmg() :50-99.
 */
#ifndef HPGMG_FV_H
#define HPGMG_FV_H

#include "hpgmg_level.h"
#include "hpgmg_operators.h"
#include "hpgmg_mg.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hpgmg_solver {
  level_type level_h;   /* finest level */
  mg_type    mg;
  double a, b, h;
  int boxes_in_i, box_dim;
  int my_rank, num_ranks;
} hpgmg_solver;

/* largest boxes_in_i with boxes_in_i^3 <= boxes_per_rank*ranks whose odd part of
 * (box_dim*boxes_in_i) is <= 11 (reference hpgmg-fv.c:181-197); -1 if none */
int hpgmg_choose_boxes_in_i(int log2_box_dim, int target_boxes_per_rank, int num_ranks);

/* create level + problem + operator + hierarchy (hpgmg-fv.c:283-308).
 * Call hpgmg_configure() first.  bc = BC_DIRICHLET or BC_PERIODIC. */
hpgmg_solver *hpgmg_solver_create(int log2_box_dim, int target_boxes_per_rank, int bc, int my_rank, int num_ranks);
/* same, but with an explicit box grid (tests use tiny domains the CLI rejects) */
hpgmg_solver *hpgmg_solver_create_explicit(int boxes_in_i, int box_dim, int bc, int my_rank, int num_ranks);
void hpgmg_solver_destroy(hpgmg_solver *s);

int         hpgmg_solver_num_levels(const hpgmg_solver *s);
level_type *hpgmg_solver_level(hpgmg_solver *s, int l);

/* F(l) = restriction of F(l-1), as the benchmark does before solving on 2h, 4h (hpgmg-fv.c:324) */
void   hpgmg_solver_restrict_rhs(hpgmg_solver *s, int l);
/* zero_vector(U); FMGSolve(...) on level l; returns ||F - A u||_inf (hpgmg-fv.c:79-81) */
double hpgmg_solver_fmg(hpgmg_solver *s, int l);
/* warmup + timed solves (hpgmg-fv.c:50-99); returns average seconds per solve */
double hpgmg_solver_bench(hpgmg_solver *s, int l, int warmup, int solves);
mg_type *hpgmg_solver_mg(hpgmg_solver *s);                               /* the hierarchy, for callers of MGSolve / FMGSolve (hpgmg_mg.h) */
void   hpgmg_solver_coefficients(const hpgmg_solver *s, double ab[2]);   /* the a, b of the test problem */
/* solve on l, l+1, l+2 then richardson_error (hpgmg-fv.c:351-366); out[0]=error out[1]=order */
void   hpgmg_solver_richardson(hpgmg_solver *s, double out[2]);

/* the reference's whole main(): prints the same report; returns 0 */
int hpgmg_fv_main(int argc, char **argv);

/* ---- small accessors so a ctypes caller never needs the struct layouts ---- */
enum { HPGMG_INFO_DIM = 0, HPGMG_INFO_BOX_DIM, HPGMG_INFO_GHOSTS, HPGMG_INFO_JSTRIDE, HPGMG_INFO_KSTRIDE,
       HPGMG_INFO_VOLUME, HPGMG_INFO_NUM_MY_BOXES, HPGMG_INFO_NUM_VECTORS, HPGMG_INFO_BOXES_IN_I,
       HPGMG_INFO_MY_RANK, HPGMG_INFO_NUM_RANKS, HPGMG_INFO_NUM_MY_BLOCKS, HPGMG_INFO_ACTIVE, HPGMG_INFO_COUNT };
void   hpgmg_level_info(const level_type *level, int out[HPGMG_INFO_COUNT]);
double hpgmg_level_h(const level_type *level);
double hpgmg_level_eigenvalue(const level_type *level);
void   hpgmg_level_set_eigenvalue(level_type *level, double dominant_eigenvalue_of_DinvA);   /* what rebuild_operator would have left (tests of a Chebyshev smoother on given coefficients) */
void   hpgmg_level_box_low(const level_type *level, int box, int out[3]);
int    hpgmg_level_list_counts(const level_type *level, int which, int shape_or_type, int out[3]);
/* whole padded box volume of one vector <-> host array of box_volume doubles */
void   hpgmg_level_read_vector(level_type *level, int box, int id, double *host_out);
void   hpgmg_level_write_vector(level_type *level, int box, int id, const double *host_in);
/* standalone level for operator-level tests */
level_type *hpgmg_level_create(int boxes_in_i, int box_dim, int ghosts, int numVectors, int bc, int my_rank, int num_ranks, double h);
void        hpgmg_level_destroy(level_type *level);
/* two-level hierarchy around an existing fine level (restriction/interpolation tests) */
mg_type    *hpgmg_mg_create(level_type *fine, double a, double b, int minCoarseDim);
void        hpgmg_mg_destroy(mg_type *mg);
level_type *hpgmg_mg_level(mg_type *mg, int l);
int         hpgmg_mg_num_levels(const mg_type *mg);
void        hpgmg_set_verbose(int v);
/* HIP build only: install the RCCL transport (id from hpgmg_hip_rccl_unique_id on rank 0) */
int         hpgmg_transport_init_rccl(const char *id128, int rank, int size);
void        hpgmg_transport_finalize_rccl(void);
/* the node-local peer-copy transport (hipIpc memory handles + stream-ordered host functions on shared counters, include/hpgmg_hip.h): `name` = a POSIX shared-memory
 * name ("/...") every rank of the job passes; rank 0 creates the segment */
int         hpgmg_transport_init_ipc(const char *name, int rank, int size);
void        hpgmg_transport_finalize_ipc(void);
/* first contact of a multi-rank job (every rank calls it once the transport is installed, before creating levels): a known pattern to and from every
 * other rank, one maximum, one rank-ordered sum, one sum over a rank subset, all checked; 0, -1 with the failing rank pair / reduction in msg, or -2 when only another rank saw a failure */
int         hpgmg_transport_selftest(char *msg, int msglen);
void        hpgmg_set_sync_timers(int on);
void        hpgmg_print_switches(void);        /* every run-time switch of the plugin (environment variable, value in force, default, meaning) on stderr; HPGMG_SWITCHES=1 prints it at load time */
void        hpgmg_set_small_fused(int mode);   /* 27-pt / fv2 / fv4: 2 (default) smooth() on levels of ONE box as one single-workgroup launch on an image of the box in LDS; 0 off */
void        hpgmg_set_small_vtail(int on);     /* 27-pt / fv2 / fv4: the rest of a V-cycle below a level of one box as ONE launch: 2 (default) on except for 27-pt GSRB, 1 on, 0 off; bit-identical */
void        hpgmg_set_small_ops(int on);       /* 1 (default): BLAS-1 calls / apply_op / residual on a level of one small box wait for the dot product or norm that
                                                  follows and go out with it as one launch (host-driven Krylov solvers); 0: a launch each */
long long   hpgmg_small_ops_groups(void);      /* such launches so far (tests) */
long long   hpgmg_small_ops_prefetched(void);  /* scalars answered from a value the previous launch formed in advance (tests) */
void        hpgmg_set_fused_bottom(int on);    /* 0: the bottom solve driven from the host (BiCGStab of host/solvers.c through the operators; tests) */
void        hpgmg_set_fused_tail(int on);      /* 0: no single-launch V-/F-cycle tails (7-pt: kernels/tail.hip; tests) */
void        hpgmg_set_brick_visits(int on);    /* 0: the 32^3 / 64^3 levels of a 7-pt V-cycle launch by launch instead of one launch per visit (kernels/brick_visit.hip); 1: on; 8 / 16: on, bricks of that side (tests) */
long long   hpgmg_brick_visits(void);          /* level visits done that way so far (tests) */
long long   hpgmg_brick_failures(void);        /* solves repeated launch by launch because a brick launch did not get all its workgroups running (tests) */
long long   hpgmg_brick_capacity_refusals(void);      /* level visits left to the launch-by-launch path because the device does not hold that many bricks at once (tests) */
void        hpgmg_set_brick_wide(int on);      /* 0: the 27-point / fv4 plugins visit their launch-bound levels launch by launch (kernels/brick_wide.hip off; tests) */
void        hpgmg_set_brick_chains(int on);    /* 0: one launch per level visit instead of one per V-cycle leg (tests) */
long long   hpgmg_pair_remote_smooths(void);   /* smooth() calls executed as sweep pairs with faces owned by other ranks (tests) */
long long   hpgmg_fused_residuals_remote(void); /* 7-point: fused residual passes (residual + restriction, residual + norm) run on levels with faces owned by other ranks (tests) */
long long   hpgmg_fv4_rb_smooths(void);        /* fv4: smooth() calls run as one-pass red + black sweeps; hpgmg_rb27_passes(): such passes of the 27-point smoother (tests) */
long long   hpgmg_rb27_passes(void);
long long   hpgmg_image_exchanges(void);       /* 27-point / fv4 across ranks: refreshes of the images of the neighbouring ranks' boxes (tests) */
/* level->timers after settling pending device timers: smooth, residual, apply_op, blas1, boundary_conditions, restriction_total,
 * interpolation_total, ghostZone_total, Total (seconds since MGResetTimers; reference level.h:162-196) */
void        hpgmg_level_timers(level_type *level, double out[9]);
void        hpgmg_set_gather_dim(int dim);  /* multi-rank: levels of <= dim^3 cells live entirely on rank 0 (default 64; 0 = the reference's rank map); call before MGBuild */
long long   hpgmg_overlap_count(void);      /* number of overlapped exchanges performed so far */
void        hpgmg_set_overlap(int on);      /* multi-rank: 1 (default) overlaps the halo exchange with the stencil launch that consumes it */
void        hpgmg_set_smoother_precision(int bits); /* 64 (default, bit-exact) or 32: mixed-precision Chebyshev smoother, BASELINE config 5 */
int         hpgmg_get_smoother_precision(void);
void        hpgmg_set_graphs(int on);       /* 1: replay the launch-bound segments of a cycle as hipGraphs (default 0: eager launches measured faster) */
void        hpgmg_set_fused_sweeps(int on); /* 1 (default): Chebyshev smooth() on boxes of side 128k runs as fused sweep pairs; 0: one launch per sweep */
void        hpgmg_set_pair_min_cells(long long cells); /* the smallest level (cells) that takes the sweep-pair kernel (default 2 000 000, HPGMG_PAIR_MIN_CELLS); tests */
void        hpgmg_set_ghost_free(int on);   /* 1 (default): fused ghost handling in the 7-pt stencil launches; 0: exchange + BC + stencil */
void        hpgmg_set_box_alignment(int jstride, int kstride, int volume, int base_bytes);

#ifdef __cplusplus
}
#endif
#endif

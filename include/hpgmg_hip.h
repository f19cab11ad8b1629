/*
 * hpgmg_hip.h -- C-ABI of libhpgmg_hip.so: the hand-written gfx950 kernels of the
 * HPGMG-FV operator layer plus the device-memory hooks.
 *
 * Plain C types only (device pointers as double*, ints, doubles, streams as
 * void*); every launcher returns 0 on success or a hipError_t value.  Each
 * entry point names the reference routine it replaces (paths relative to
 * finite-volume/source/).  The host-side plugin hpgmg_amd/csrc/host/operators_hip.c
 * implements include/hpgmg_operators.h (= the reference's operators.h) on top of
 * these; tests call them directly through ctypes.
 *
 * Geometry of a level is passed as one POD record.  All boxes of a level have
 * the same dim/ghosts/strides (reference level.c:935-938); box b's vector v
 * starts at box_base[b] + v*volume and cell (i,j,k) of it sits at
 * (i+g) + (j+g)*jStride + (k+g)*kStride.
 */
#ifndef HPGMG_HIP_H
#define HPGMG_HIP_H

#include <stddef.h>
#include "hpgmg_level.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double *const *box_base; /* DEVICE array [num_boxes]: vectors[0] of each owned box        */
  const int *box_low;      /* DEVICE array [3*num_boxes]: global (i,j,k) of first interior cell */
  int num_boxes;
  int dim, ghosts, jStride, kStride, volume;
  int dim_i, dim_j, dim_k; /* global cells per side (Dirichlet masks in rebuild)             */
  int periodic;
  const int *box_nbr;      /* DEVICE array [6*num_boxes] or NULL: local index of the box across the -i,+i,-j,+j,-k,+k
                              face; -1 = homogeneous-Dirichlet domain face; -2 = box on another rank (use the ghost zone) */
  int flags;               /* bit 0: the first interior cell of every box and vector is 16-byte aligned (enables 16-B loads) */
  long long box_stride;    /* doubles between the bases of consecutive boxes when that distance is the same for all boxes (one slab), else 0 */
} hpgmg_hip_level;

/* stencil variants of apply_op_ijk (operators.7pt.c:49-89, operators.27pt.c:60-91, operators.fv4.c:55-134) */
enum { HPGMG_HIP_7PT_VC_HELMHOLTZ = 0, HPGMG_HIP_7PT_VC_POISSON = 1, HPGMG_HIP_7PT_CC = 2,
       HPGMG_HIP_27PT_CC = 3, HPGMG_HIP_FV4_VC_HELMHOLTZ = 4, HPGMG_HIP_FV4_VC_POISSON = 5 };

/* ---- runtime / memory (replaces MALLOC of level.c:25-40 for vector data) ---- */
int    hpgmg_hip_device_count(void);
int    hpgmg_hip_set_device(int dev);
void   hpgmg_hip_set_stream(void *hip_stream);     /* stream every launcher below enqueues on (default: null stream) */
void  *hpgmg_hip_get_stream(void);
int    hpgmg_hip_sync(void);                        /* hipStreamSynchronize on that stream */
void  *hpgmg_hip_malloc(size_t bytes);              /* zero-filled device memory, NULL on failure */
void   hpgmg_hip_free(void *p);
void  *hpgmg_hip_host_malloc(size_t bytes);         /* zero-filled pinned host memory the device can write, NULL on failure */
void   hpgmg_hip_host_free(void *p);
int    hpgmg_hip_memcpy_h2d(void *dst, const void *src, size_t bytes);
int    hpgmg_hip_memcpy_d2h(void *dst, const void *src, size_t bytes);
int    hpgmg_hip_memcpy_d2d(void *dst, const void *src, size_t bytes);
int    hpgmg_hip_memset0(void *dst, size_t bytes);
const char *hpgmg_hip_last_error(void);
/* event timing on the launch stream (bench.py: average kernel time over the timed region) */
void  *hpgmg_hip_event_create(void);
void   hpgmg_hip_event_destroy(void *ev);
int    hpgmg_hip_event_record(void *ev);
void  *hpgmg_hip_stream_create(void);               /* a second (non-blocking) stream, e.g. for the halo exchange */
void   hpgmg_hip_stream_destroy(void *stream);
int    hpgmg_hip_stream_wait_event(void *ev);       /* the CURRENT launch stream waits for ev */
double hpgmg_hip_event_elapsed_ms(void *start, void *stop); /* synchronises on stop */
/* Device-time attribution for the per-level timing table (reference level.h:162-196, mg.c:54-161): a hipEvent pair on the
 * launch stream around an operator.  begin() returns a slot (-1: not recorded, e.g. inside a graph capture); the elapsed
 * device time is ADDED to *acc_seconds by flush() (called by the plugin before the table is printed / reset; also when the
 * pool of 8192 pairs is full).  forget(lo, hi): accumulators in [lo, hi) are about to be freed -- settle them now. */
int    hpgmg_hip_timer_begin(double *acc_seconds);
void   hpgmg_hip_timer_end(int slot);
int    hpgmg_hip_timer_flush(void);
void   hpgmg_hip_timer_forget(const void *lo, const void *hi);
/* roctx ranges around operators (rocprofv3 --marker-trace); active only with HPGMG_ROCTX=1 (library resolved with dlopen) */
int    hpgmg_hip_range_enabled(void);
void   hpgmg_hip_range_push(const char *name);
void   hpgmg_hip_range_pop(void);
/* accumulate the GPU time of every smoother-kernel launch between begin/end (hipEvents around each launch) */
void   hpgmg_hip_profile_smoother(int enable);
void   hpgmg_hip_profile_smoother_min_cells(long long min_cells); /* time only launches over >= this many cells */
/* time only every stride-th eligible launch (an event pair idles the GPU ~10 us per timed launch); 1 = all */
void hpgmg_hip_profile_smoother_stride(int stride);
int    hpgmg_hip_profile_smoother_read(double *total_ms, long long *launches, long long *cells);

/* Ghost-free mode of the 7-pt stencil launchers below.  When `on` and L->box_nbr is given, a face
 * neighbour that lies in another local box is read from that box directly and a Dirichlet face is
 * evaluated as -x(centre) (= what apply_BCs_p1 would have stored, boundary_fd.c:35-65), so the
 * caller may skip exchange_boundary()'s local copies and apply_BCs() for STAR-shaped stencils:
 * one launch per sweep instead of three.  Interior results are bit-identical; ghost cells of the
 * operand are then simply not refreshed (every consumer refreshes or bypasses them itself). */
void hpgmg_hip_set_ghost_free(int on);
int  hpgmg_hip_get_ghost_free(void);

/* ---- smoothers: operators/chebyshev.c:43-99, operators/gsrb.c:24-132, operators/jacobi.c:8-65 ----
 * One sweep over every owned box.  x_n/x_np1/rhs are vector ids.  Chebyshev and
 * Jacobi read x_n and write x_np1 (x_np1 doubles as x_{n-1} for Chebyshev);
 * GSRB updates the cells whose global parity (i+j+k+sweep) is even, in place
 * when xn_id == xnp1_id, otherwise copying the other colour. */
/* interpolation_vcycle (piecewise constant, prescale 1.0; interpolation_p0.c:43) folded into the first two SINGLE Chebyshev sweeps of the smooth() that follows it
 * (mg.c:1160-1161) on levels the sweep-pair kernel does not take: the NEXT hpgmg_hip_smooth_cheby launch reads x_n (which = 1, sweep 0) resp. x_{n-1} (which = 2, sweep 1)
 * as stored + the coarse value above the cell; map = per fine box the coarse box and the coarse cell under its first cell (device memory).  Ghost-free path, every box local. */
int  hpgmg_hip_smooth_cheby_fold_supported(const hpgmg_hip_level *L, int variant);
void hpgmg_hip_stencil_fold_interpolation(const hpgmg_hip_level *Lc, int coarse_id, const int *map, int which);
int hpgmg_hip_smooth_cheby(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                           double a, double b, double h2inv, double c1, double c2);
int hpgmg_hip_smooth_gsrb(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                          double a, double b, double h2inv, int sweep);
int hpgmg_hip_smooth_jacobi(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                            double a, double b, double h2inv, double weight);
/* operators/residual.c:9-51 (rhs_id >= 0: res = rhs - A x) and operators/apply_op.c:9-48 (rhs_id < 0: res = A x) */
int hpgmg_hip_residual(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id,
                       double a, double b, double h2inv);
/* Fused forms of residual() for the bandwidth-bound fine level (ghost-free 7-point path, every face neighbour local, box side a
 * multiple of 128; _supported() says whether a level qualifies):
 *   _restrict: res = rhs - A x is not stored but restricted (restriction.c:54-57, same summation order) into vector coarse_id of Lc,
 *              and, when zero_id >= 0, zero_vector(Lc, zero_id) runs in the same launch -- residual + restriction + zero_vector of
 *              MGVCycle's down leg (mg.c:1150-1153) as one pass: 58 instead of 74 B per fine cell.  map = DEVICE array, 4 ints per fine
 *              box: coarse box index and the coarse (i, j, k) under the fine box's first cell.
 *   _norm:     res is stored AND max |res| is returned (residual + norm of the convergence check, mg.c:1321-1323). */
int hpgmg_hip_residual_fused_supported(const hpgmg_hip_level *L, int variant);
/* the restriction form alone (hpgmg_hip_residual_restrict / _store) also takes the launch-bound 7-point levels of boxes of an even side <= 32 */
int  hpgmg_hip_residual_restrict_supported(const hpgmg_hip_level *L, int variant);
int hpgmg_hip_residual_restrict(const hpgmg_hip_level *L, int variant, int x_id, int rhs_id, double a, double b, double h2inv,
                                const hpgmg_hip_level *Lc, int coarse_id, const int *map, int zero_id);
/* the same, and the residual is ALSO stored to vector res_id (>= 0; 7-point kernels): exactly the state residual() + restriction() + zero_vector() leave */
int hpgmg_hip_residual_restrict_store(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id, double a, double b, double h2inv,
                                      const hpgmg_hip_level *Lc, int coarse_id, const int *map, int zero_id);
int hpgmg_hip_residual_norm(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id, double a, double b, double h2inv, double *norm_out);
/* Small levels (<= hpgmg_hip_small_level_max_cells() cells, every exchange copy local) of the 27-point / fv2 / fv4 plugins: a whole
 * smooth() -- per sweep exchange_boundary (local copy list), apply_BCs (list + kind: 1 p1, 2 p2, 3 v2, 4 v4; zero_first = clear the
 * regions before extrapolating, boundary_fv.c:133-140) and the stencil -- or a residual() / apply_op() in ONE single-workgroup launch.
 * mode: 0 Chebyshev (c1, c2 per sweep), 1 GSRB (in place, or via VECTOR_TEMP when out_of_place), 2 Jacobi (c2 = weight), 3 residual
 * (res_id = rhs - A x), 4 apply_op (res_id = A x).  Same entry routines and per-cell expressions as the streaming kernels. */
int hpgmg_hip_small_level_max_cells(void);
int hpgmg_hip_small_level_op(const hpgmg_hip_level *L, int variant, int mode, int sweeps, int x_id, int rhs_id, int res_id, int out_of_place,
                             double a, double b, double h2inv, const double *c1, const double *c2,
                             const blockCopy_type *copy_list, int n_copy, const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first);
/* Overlap of the halo exchange with the stencil launches below (ghost-free 7-point path only): mode 1 = the next
 * launches leave the cells next to a face owned by another rank (box_nbr == -2) untouched, mode 2 = the next launches
 * compute exactly those cells (one lane per cell), mode 0 = whole boxes. */
/* Bottom solve of the 27-point / fv2 / fv4 plugins: diagonally preconditioned BiCGStab (solvers/bicgstab.c:14-97, as host/solvers.c drives
 * it) on a bottom level of ONE box of at most max_cells() cells, Dirichlet, in one single-workgroup launch: x_id = initial guess and
 * solution, eight work vectors from krylov_base, boundary list / kind / zero_first as for hpgmg_hip_small_level_op, krylov_iterations =
 * device-visible host counter the kernel adds its iteration count to (or NULL).  Bit-identical to the host-driven solve. */
int hpgmg_hip_bottom_bicgstab_max_cells(void);
int hpgmg_hip_bottom_bicgstab(const hpgmg_hip_level *L, int variant, int x_id, int rhs_id, int krylov_base, double a, double b, double h2inv, double want,
                              const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first, int *krylov_iterations);
/* The rest of a V-cycle (mg.c:1133-1166 MGVCycle) of the 27-point / fv2 / fv4 plugins below a level of ONE box, as one single-workgroup
 * launch: per level smooth, residual, restriction (restriction.c:49-60), zero_vector of the coarse correction; the BiCGStab bottom solve
 * (as hpgmg_hip_bottom_bicgstab); then per level upwards interpolation_vcycle (interpolation_p2.c / _v2.c: apply_BCs of the coarse
 * correction over STENCIL_SHAPE_BOX, then fine += tensor rule) and smooth.  lv[0] is the finest level of the chain, lv[n-1] the bottom.  Every
 * level is one box whose nine vectors fit the LDS (lds_doubles() tells how much a chain needs); Dirichlet.  mode: 0 Chebyshev, 1 GSRB,
 * 2 Jacobi (as hpgmg_hip_small_level_op).  The same entry routines and per-cell expressions as the per-operator kernels: bit-identical. */
/* A queue of BLAS-1 / operator calls on a level of ONE box of side <= 8 (Dirichlet) as one single-workgroup launch: what a host-driven Krylov
 * solver issues between two scalars it needs (solvers/bicgstab.c through operators.h).  kinds[q]: 1 add_vectors c = sa*a + sb*b, 2 mul_vectors
 * c = sa*a*b, 3 scale_vector c = sa*a, 4 apply_op c = A a, 5 residual c = b - A a (operators' a, b, h2inv; boundary list / kind / zero_first as for
 * hpgmg_hip_small_level_op), 6 dot(a, b), 7 norm(a) -- the last two only as the last entry or the last TWO entries (the second: a value the caller expects to be asked for
 * next), their values go to *value_out / *value2_out (non-NULL exactly then; the call waits for them).  At most hpgmg_hip_small_ops_max() entries.  Same expressions and summation order as the per-operator kernels. */
int hpgmg_hip_small_ops_max(void);
int hpgmg_hip_small_ops(const hpgmg_hip_level *L, int variant, int n, const int *kinds, const int *c, const int *a, const int *b, const double *sa, const double *sb,
                        const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first, double op_a, double op_b, double h2inv, double *value_out, double *value2_out);
long long hpgmg_hip_small_ops_launch_count(void);   /* launches so far (tests) */
#define HPGMG_HIP_SMALL_TAIL_MAX_LEVELS 4
typedef struct {
  hpgmg_hip_level L;
  const blockCopy_type *bc_list, *ibc_list;   /* DEVICE lists: boundary blocks of the operator's stencil shape / of STENCIL_SHAPE_BOX (interpolation FROM this level) */
  int n_bc, bc_kind, zero_first, n_ibc, ibc_kind, ibc_zero_first;   /* kinds: 1 p1, 2 p2, 3 v2, 4 v4; zero_first: clear the blocks before the condition fills its layers */
  double h2inv, c1[8], c2[8];                  /* Chebyshev coefficients per sweep (Jacobi: c2 = the weight) */
} hpgmg_hip_small_tail_level;
typedef struct {
  int n, mode, sweeps, out_of_place, e_id, R_id, krylov_base, legs;   /* legs: bit 0 the way down, bit 1 the bottom solve, bit 2 the way up (7 = the whole tail) */
  double a, b, want;
  int *krylov_iterations;                      /* device-visible host counter the bottom solve adds its iteration count to, or NULL */
  hpgmg_hip_small_tail_level lv[HPGMG_HIP_SMALL_TAIL_MAX_LEVELS];
} hpgmg_hip_small_tail_args;
long long hpgmg_hip_small_vtail_lds_doubles(const hpgmg_hip_small_tail_args *args);   /* > what a workgroup has: the chain does not qualify */
long long hpgmg_hip_small_vtail_lds_limit(void);
int hpgmg_hip_small_vtail(const hpgmg_hip_small_tail_args *args, int variant);
long long hpgmg_hip_small_vtail_launch_count(void);   /* launches so far (tests) */
void hpgmg_hip_set_defer_mode(int mode);
/* Two-part launches of the tiled kernels that read neighbouring ranks' cells from images (boxes listed in box_base / box_low / box_nbr
 * BEHIND the num_boxes own ones, reached through box_nbr like any neighbour): part 1 = the tiles whose halo reaches no image (for the fv4
 * red + black pass also no domain wall: nothing its pre-pass forms), part 2 = the others, 0 = whole launches.  Lets the caller run the
 * exchange that refreshes the images on a second stream under part 1 (north_star: "ghost-zone exchange ... overlapped with interior
 * smoothing"; reference operators/exchange_boundary.c:81-90 only overlaps the local copies).  Applies to hpgmg_hip_smooth_gsrb_fv4_rb,
 * hpgmg_hip_smooth_gsrb27_rb, the sweep-pair launches with remote faces, and the LDS-tiled 27-point / fv4 kernels behind hpgmg_hip_smooth_* /
 * hpgmg_hip_residual (plain forms; the fused residual forms always run whole). */
void hpgmg_hip_set_tile_part(int part);
/* The LDS-tiled 27-point and fv4 kernels (boxes whose side is a multiple of 64, out of place) can read x outside a box from the
 * neighbouring box itself when every box of the level is local: the caller then runs only apply_BCs before the launch, not
 * exchange_boundary.  applies() tells whether the next smooth / residual / apply_op launch of `variant` would be such a kernel. */
void hpgmg_hip_set_tile_ghost_free(int on);
void hpgmg_hip_set_27pt_tile32(int on);
/* Both coloured half sweeps (sweep, sweep + 1; sweep even) of one out-of-place GSRB sweep of the 27-point operator in one pass
 * (gsrb.c:24-132 twice; kernels/stencil27_rb.hpp): x_id -> out_id, the intermediate vector and its boundary conditions live in LDS.
 * Boxes of side 64 m, every box local; the caller has run apply_BCs_p2 on x_id (no exchange_boundary needed). */
int  hpgmg_hip_smooth_gsrb27_rb_supported(const hpgmg_hip_level *L);
int  hpgmg_hip_smooth_gsrb27_rb(const hpgmg_hip_level *L, int x_id, int out_id, int rhs_id, double a, double b, double h2inv, int sweep);
/* The same on small levels (boxes of 2^3 ... 32^3), one workgroup per cube of at most 8^3 cells held in LDS (kernels/stencil27_rb_box.hpp):
 * forms the domain-boundary ghost cells of x_id itself -- the caller runs neither exchange_boundary nor apply_BCs_p2.  Every box local. */
int  hpgmg_hip_smooth_gsrb27_rb_box_supported(const hpgmg_hip_level *L);
void hpgmg_hip_set_27pt_rb_box_maxdim(int dim);   /* largest box side taken (default 8; 16 and 32 work as cubes of 8^3 but are not faster) */
int  hpgmg_hip_smooth_gsrb27_rb_box(const hpgmg_hip_level *L, int x_id, int out_id, int rhs_id, double a, double b, double h2inv, int sweep);
long long hpgmg_hip_rb27_launch_count(void);   /* launches of either kernel so far (tests) */   /* tiled 27-point kernel also for boxes of 32^3 (off by default: slower than the register kernel there; HPGMG_TUNE_27PT_TILE32=1) */
int  hpgmg_hip_tile_kernel_applies(const hpgmg_hip_level *L, int variant, int out_of_place);

/* ---- fused forms of smooth() for bandwidth-bound levels (kernels/cheby_pair.hpp) ---- */
/* Two consecutive Chebyshev sweeps (chebyshev.c:43-99 twice) in one pass: x1 = S(x0, xm1; c1a, c2a),
 * x2 = S(x1, x0; c1b, c2b); bit-identical to two hpgmg_hip_smooth_cheby calls with the Dirichlet ghost rule between
 * them.  Each vector is (scratch?, id): scratch ids 0/1 are the two extra vectors per box behind scr_base[box]
 * (same padded layout as the level's vectors).  out1/out2 must differ from x0/xm1.  Needs every face neighbour
 * local, Dirichlet, rows of 128 cells (box side a multiple of 128, or a divisor of 128 with the boxes in one slab), boxes
 * numbered lexicographically (_supported() checks the rest). */
int hpgmg_hip_smooth_cheby_pair_supported(const hpgmg_hip_level *L, int variant);
int hpgmg_hip_smooth_cheby_pair(const hpgmg_hip_level *L, int variant, double *const *scr_base, const float *const *c32_base,
                                int x0_scr, int x0_id, int xm1_scr, int xm1_id, int out1_scr, int out1_id, int out2_scr, int out2_id,
                                int rhs_id, double a, double b, double h2inv, double c1a, double c2a, double c1b, double c2b);
/* Sweep pairs across rank boundaries (boxes over the GPUs of a node, halo exchange over RCCL).  This rank's boxes form a
 * brick of brick_boxes[0..2] boxes (numbered lexicographically inside it); remote_face[f] (f = -i,+i,-j,+j,-k,+k) is 1 where the
 * brick face belongs to another rank, 0 where it is the (Dirichlet) domain boundary.  Before the launch the caller must have
 * brought, across every remote face: x0 two cells deep (ghost zone + `deep`), xm1, the right-hand side and every coefficient
 * vector one cell deep (ghost zones; x0 also on the brick's edges), and `deep_beta` = the normal beta component at index dim+1 on
 * the high faces.  The kernel then forms x1 on the ghost layer itself -- the same expression the owning rank evaluates -- so ONE
 * exchange serves TWO sweeps (reference: one exchange_boundary per sweep, chebyshev.c:45-46).  Consumed by the next
 * hpgmg_hip_smooth_cheby_pair / _gsrb_pair launch.  Layouts: deep[box][face 0..5][v][u], deep_beta[box][0..2 = +i,+j,+k][v][u],
 * planes of dim x dim values, (u, v) = the two in-face axes in i < j < k order. */
int  hpgmg_hip_smooth_cheby_pair_supported_brick(const hpgmg_hip_level *L, int variant, int nbi, int nbj, int nbk);
void hpgmg_hip_pair_set_halo(const int brick_boxes[3], const int remote_face[6], const double *deep, const double *deep_beta);
/* The NEXT hpgmg_hip_smooth_cheby_pair launch need not store x1: its out1 vector is scratch to the caller (the cycle driver's
 * smooth(): nothing reads VECTOR_TEMP after it).  Saves one of the launch's ten streams; x2 (out2) is unaffected. */
void hpgmg_hip_pair_discard_x1(void);
/* The sweep-pair launches keep a packed copy of the coefficient values their pre-pass reads (eight per cell of the columns next to interior tile
 * edges; built on the first launch of a level): _invalidate(L) after its coefficient vectors changed (rebuild_operator / initialize_problem),
 * _forget(L) before the level's storage is freed; L = NULL: every level. */
void hpgmg_hip_pair_packed_invalidate(const hpgmg_hip_level *L);
void hpgmg_hip_pair_packed_forget(const hpgmg_hip_level *L);
void hpgmg_hip_pair_launch_counts(long long out[2]);      /* sweep-pair launches so far: all, and those with remote faces (tests) */
/* One region of a sweep-pair halo message.  vec: 0 = the pair's x0, 1 = its xm1, 2 = its right-hand side, 16 + id = level vector id.
 * Pack copies the region (i fastest) to sendbuf + off; unpack copies recvbuf + off into the region (deep = -1: ghost cells at
 * (i,j,k)), into deep plane `deep` (0..5) or into deep_beta plane deep - 8 (8..10) of the box. */
typedef struct { int box, vec, i, j, k, ni, nj, nk, deep, pad_; long long off; } hpgmg_hip_halo_entry;
/* the NEXT hpgmg_hip_pair_halo_pack adds the coarse parent (interpolation_p0.c:43) to the x0 values it packs: across ranks the folded interpolation of
 * hpgmg_hip_pair_fold_interpolation is applied by each cell's OWNER, so ghost zones and deep planes arrive interpolated */
void hpgmg_hip_pair_halo_fold_interpolation(const hpgmg_hip_level *Lc, int coarse_id, double prescale);
int  hpgmg_hip_pair_halo_pack(const hpgmg_hip_level *L, double *const *scr_base, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id,
                              const hpgmg_hip_halo_entry *entries, int n, double *sendbuf);
int  hpgmg_hip_pair_halo_unpack(const hpgmg_hip_level *L, double *const *scr_base, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id,
                                const hpgmg_hip_halo_entry *entries, int n, double *recvbuf, double *deep, double *deep_beta);
/* Two consecutive in-place GSRB half sweeps (gsrb.c:24-132, colours `sweep` and `sweep + 1`) in one pass: reads x0, writes the
 * result to out2 (which must differ from x0); scratch vector edge_scr_id (0/1 behind scr_base) is clobbered. */
int hpgmg_hip_smooth_gsrb_pair(const hpgmg_hip_level *L, int variant, double *const *scr_base, const float *const *c32_base,
                               int x0_scr, int x0_id, int edge_scr_id, int out2_scr, int out2_id, int rhs_id,
                               double a, double b, double h2inv, int sweep);
/* Fold interpolation_vcycle (interpolation_p0.c:43, f = prescale*f + parent) into the NEXT hpgmg_hip_smooth_cheby_pair /
 * _gsrb_pair launch: that launch reads its x0 as prescale*x0 + coarse parent (vector coarse_id of Lc, same box numbering,
 * boxes half the size) instead of x0 as stored, which is never materialised.  Consumed by that one launch.
 * Lc must stay valid until the launch has been issued. */
void hpgmg_hip_pair_fold_interpolation(const hpgmg_hip_level *Lc, int coarse_id, double prescale);
/* Mixed-precision smoother (BASELINE config 5): c32_base[box] = 5 x volume floats holding fp32 copies of Dinv, alpha,
 * beta_i, beta_j, beta_k (whole padded vectors, same indexing).  hpgmg_hip_coef32_refresh fills them from the level's
 * vectors; passing them to hpgmg_hip_smooth_cheby_pair makes the sweep pair read 4-byte coefficients (iterate, right-hand
 * side and all arithmetic stay fp64).  c32_base == NULL is the bit-exact fp64 smoother. */
int hpgmg_hip_coef32_refresh(const hpgmg_hip_level *L, float *const *c32_base, int num_vectors /* of the level: absent coefficient vectors are skipped */);

/* ---- block lists.  `blocks` is a DEVICE copy of a host blockCopy_type array ---- */
/* operators/blockCopy.c:6-105 CopyBlock over a list (exchange_boundary pack/local/unpack, restriction unpack) */
int hpgmg_hip_copy_blocks(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int num_blocks);
/* operators/blockCopy.c:109-156 IncrementBlock (interpolation unpack): w = prescale*w + r */
int hpgmg_hip_increment_blocks(const hpgmg_hip_level *L, int id, double prescale, const blockCopy_type *blocks, int num_blocks);
/* operators/boundary_fd.c:6-90 apply_BCs_p1 */
int hpgmg_hip_apply_bc_p1(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int num_blocks);
/* operators/boundary_fd.c:93-205 apply_BCs_p2 */
int hpgmg_hip_apply_bc_p2(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int num_blocks);
/* operators/boundary_fv.c:101-250 apply_BCs_v2, :262-569 apply_BCs_v4, :573-681 extrapolate_betas (BOX list) */
int hpgmg_hip_apply_bc_v2(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int num_blocks);
int hpgmg_hip_apply_bc_v4(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int num_blocks);
int hpgmg_hip_extrapolate_betas(const hpgmg_hip_level *L, const blockCopy_type *blocks, int num_blocks);
/* apply_BCs_v2 / apply_BCs_v4 / apply_BCs_p2 from entries whose geometry the host has worked out (one per boundary_condition block, any order):
 * cell (r, q) of the entry is x[base + r fs0 + q fs1] in vector `id` of box `box`, 0 <= r < len0, 0 <= q < len1; `nn` axes (1 face, 2 edge,
 * 3 corner) leave the domain there and step[0..nn-1] lead back into it.  The values the condition READS are taken at the same offsets
 * from (src_box, src_base): normally the entry's own box; for a block that runs along another box's face (its in-face coordinates lie in
 * this box's ghost zone) the host may name that neighbouring box instead -- after an exchange the values are the same, but read there the
 * condition does not depend on the exchange: hpgmg_hip_exchange_and_bc() uses the sources and runs both as one launch, while
 * hpgmg_hip_apply_bc_fv() (= apply_BCs on its own) reads the entry's own box like the reference.
 * Short kernels: on the small levels a launch lasts as long as its instruction fetch.  order = 2 (v2) | 4 (v4) | 12 (p2).  A ghost zone
 * deeper than the condition fills (v4: 2 layers, v2: 1) is cleared first, block by block, as the reference does; p2 needs ghosts == 1. */
typedef struct { int box, nn, base, len0, len1, fs0, fs1, step[3], src_box, src_base;
                 int zbase, zi, zj, zk;   /* the whole block (offset of its first cell, extents): cleared first when the ghost zone is deeper than the condition fills */
} hpgmg_hip_bc_entry;
int hpgmg_hip_apply_bc_fv(const hpgmg_hip_level *L, int id, const hpgmg_hip_bc_entry *entries, int num_entries, int order);
/* exchange_boundary's box-to-box copies (a `copy` list, operators/exchange_boundary.c:81-90) and the boundary conditions in ONE launch:
 * only for entries whose sources do not depend on the copies (see above) and levels without messages */
int hpgmg_hip_exchange_and_bc(const hpgmg_hip_level *L, int id, const blockCopy_type *copies, int num_copies,
                              const hpgmg_hip_bc_entry *entries, int num_entries, int order);

/* Both coloured half sweeps of one out-of-place GSRB sweep of the 4th-order operator (operators.fv4.c:55-134 with GSRB_OOP,
 * gsrb.c:24-132 twice, apply_BCs_v4 boundary_fv.c:262-569 in between) in one pass: kernels/fv4_rb.hpp.  Vectors are (scratch, id) pairs;
 * scratch = 1 addresses the plugin-private vectors behind scr_base (same box layout as the level's).  The intermediate vector lives in
 * LDS; its ghost planes below / above the domain are read from the k ghost zone of scratch vector tg_id, which _prepass fills first
 * (entries_k: the geometry of the boundary blocks whose domain normal has a k component; special_cells: the cells on internal box faces that
 * are next to a domain wall -- their intermediate value is formed by the pre-pass with the owning box's coefficients and read by the main
 * kernel, which recomputes every other cell next to its tiles itself).  sweep = number of the first (even) half sweep. */
int  hpgmg_hip_smooth_gsrb_fv4_rb_supported(const hpgmg_hip_level *L, int variant);
int  hpgmg_hip_fv4_rb_prepass(const hpgmg_hip_level *L, int variant, double *const *scr_base, int x_scratch, int x_id, int tg_id, int rhs_id,
                              double a, double b, double h2inv, int sweep, const hpgmg_hip_bc_entry *entries_k, int n_k,
                              const int *special_cells /* DEVICE: box, i, j, k per cell */, int n_special);
int  hpgmg_hip_smooth_gsrb_fv4_rb(const hpgmg_hip_level *L, int variant, double *const *scr_base, int x_scratch, int x_id, int out_scratch, int out_id,
                                  int tg_id, int rhs_id, double a, double b, double h2inv, int sweep);
long long hpgmg_hip_rb_fv4_launch_count(void);   /* launches of the one-pass kernel so far (tests) */
/* operators/restriction.c:6-94 restriction_pc_block over a list; type = RESTRICT_* */
int hpgmg_hip_restrict_blocks(const hpgmg_hip_level *Lc, int id_c, const hpgmg_hip_level *Lf, int id_f,
                              const blockCopy_type *blocks, int num_blocks, int type);
/* restriction (cell) of a LOCAL list and zero_vector(coarse, zero_id) -- misc.c:6-44, whole padded boxes -- in one launch */
int hpgmg_hip_restrict_cell_and_zero(const hpgmg_hip_level *Lc, int id_c, const hpgmg_hip_level *Lf, int id_f,
                                     const blockCopy_type *blocks, int num_blocks, int zero_id);
/* operators/interpolation_p0.c:6-46 (order 0), interpolation_p1.c:8-65 (order 1), and the tensor-product
 * interpolation_p2.c (order 2), interpolation_v2.c (order 3), interpolation_v4.c (order 4) over a list */
int hpgmg_hip_interpolate_blocks(const hpgmg_hip_level *Lf, int id_f, double prescale, const hpgmg_hip_level *Lc, int id_c,
                                 const blockCopy_type *blocks, int num_blocks, int order);

/* ---- BLAS-1: operators/misc.c ---- */
int hpgmg_hip_fill(const hpgmg_hip_level *L, int id, double interior_value);          /* zero_vector :6 / init_vector :48 (ghosts := 0) */
int hpgmg_hip_axpby(const hpgmg_hip_level *L, int id_c, double sa, int id_a, double sb, int id_b); /* add_vectors :94 */
int hpgmg_hip_mul(const hpgmg_hip_level *L, int id_c, double s, int id_a, int id_b);  /* mul_vectors :131 */
int hpgmg_hip_invert(const hpgmg_hip_level *L, int id_c, double s, int id_a);         /* invert_vector :168 */
int hpgmg_hip_scale(const hpgmg_hip_level *L, int id_c, double s, int id_a);          /* scale_vector :204 */
int hpgmg_hip_shift(const hpgmg_hip_level *L, int id_c, int id_a, double shift);      /* shift_vector :386 */
int hpgmg_hip_color(const hpgmg_hip_level *L, int id, int colors, int ic, int jc, int kc); /* color_vector :441 */
int hpgmg_hip_random(const hpgmg_hip_level *L, int id);                               /* random_vector :478 */
/* reductions over interior cells of this rank's boxes; synchronise and return through *out.
 * max: exact under any order.  sums: one partial per dim x 8 x 8 tile in k,j,i order,
 * partials added in tile order -- the order of the reference run with one thread. */
int hpgmg_hip_norm_max(const hpgmg_hip_level *L, int id, double *out);                /* norm :287 */
int hpgmg_hip_dot(const hpgmg_hip_level *L, int id_a, int id_b, double *out);         /* dot :239 */
/* norm(F), scale_vector(R, 1.0, F) and restriction(coarse rc_id <- R, RESTRICT_CELL) -- the opening of FMGSolve, mg.c:1262-1270 -- in one pass
 * over F (even box side, 16-byte aligned rows; map as for hpgmg_hip_residual_restrict): 25 instead of 41 B per fine cell */
int hpgmg_hip_norm_copy_restrict(const hpgmg_hip_level *L, int f_id, int r_id, const hpgmg_hip_level *Lc, int rc_id, const int *map, double *norm_out);
/* the same pass, but the maximum is NOT waited for: it lands in a slot of its own and hpgmg_hip_deferred_fetch() collects it when the caller needs it
 * (FMGSolve uses norm(F) only in the convergence check at its end, mg.c:1262,1323: the host keeps launching behind the pass) */
int  hpgmg_hip_norm_copy_restrict_deferred(const hpgmg_hip_level *L, int f_id, int r_id, const hpgmg_hip_level *Lc, int rc_id, const int *restrict_map);
int  hpgmg_hip_deferred_fetch(double *out);
int hpgmg_hip_sum(const hpgmg_hip_level *L, int id, double *out);                     /* mean :336 (before the divide) */

/* ---- operators/rebuild.c:47-208 black-box rebuild: accumulate one colouring (x = 0/1 pattern, ghosts
 *      already exchanged / BCs applied) into Aii and sum|Aij|, then turn them into Dinv, L1inv, lambda_max ---- */
int hpgmg_hip_blackbox_accumulate(const hpgmg_hip_level *L, int variant, int x_id, int Aii_id, int sumAbs_id, double a, double b, double h2inv);
int hpgmg_hip_blackbox_finalize(const hpgmg_hip_level *L, int Aii_id, int sumAbs_id, double a, double b, double h2inv, double *lambda_max_out);

/* ---- operators.7pt.c:158-227: Dinv (+L1inv when l1inv_id >= 0) and the Gershgorin bound of lambda_max(D^-1 A) ---- */
int hpgmg_hip_rebuild_7pt(const hpgmg_hip_level *L, int variable_coeff, int alpha_id, int l1inv_id,
                          double a, double b, double h2inv, double *lambda_max_out);

/* ---- tail.hip: the part of a V-cycle below a small level -- a chain of levels of <= hpgmg_hip_tail_max_cells()
 *      cells each, all face neighbours local -- as ONE single-workgroup launch (or one per leg): the operator
 *      sequence of mg.c:1147-1163 (smooth, residual, restriction, zero_vector | interpolation_vcycle, smooth) with
 *      barriers where the driver has kernel boundaries, and optionally the bottom solve.  levels[n-1] is the bottom level. ---- */
int hpgmg_hip_tail_max_levels(void);
int hpgmg_hip_tail_max_cells(void);          /* per smoothed level of the chain */
int hpgmg_hip_tail_bottom_max_cells(void);   /* bottom level, when its BiCGStab solve runs on the device (leg 2, 3) */
/* leg 0 / 1: down / up legs around a host-driven bottom solve;  leg 2: down legs, the bottom solve
 * (diagonally preconditioned BiCGStab, solvers/bicgstab.c:14-97, work vectors krylov_base..+7, stop at
 * bottom_norm relative reduction), up legs -- one launch;  leg 3: the bottom solve alone (n >= 1);
 * leg 4: the F-cycle below levels[0] (FMGSolve, mg.c:1270-1300): restriction of the right-hand side R_id down the chain, zero_vector
 * + bottom solve, then for every level upwards interpolation_fcycle (interpolation_p1.c, coarse ghosts by apply_BCs_p1) and a V-cycle.
 * krylov_iterations: device-visible host counter the kernel adds its iteration count to, or NULL. */
int hpgmg_hip_vcycle_tail(int n, const hpgmg_hip_level *const *levels, const double *h2inv,
                          const double *c1, const double *c2, int sweeps,
                          int variant, int smoother, int e_id, int R_id, double a, double b, int leg,
                          int krylov_base, double bottom_norm, int *krylov_iterations);

/* ---- brick_visit.hip: the visits of the launch-bound levels ABOVE the tail's reach (16^3 .. 64^3 cells) -- each level as bricks of 8^3 or 16^3 cells, a
 *      workgroup each, iterate in LDS, coefficients in registers; between sweeps the bricks trade faces through memory inside the launch ({value,
 *      sequence number} records, written through and polled: one memory hop, 1.6 us), and so do the levels of a chain (restricted residuals on the
 *      way down, corrections on the way up).  Replaces, per level visit, the 5 + 4 launches of mg.c:1147-1153 (smooth, residual, restriction,
 *      zero_vector) and mg.c:1160-1161 (interpolation_vcycle, smooth).  Every face neighbour local, Dirichlet, cubic domain, boxes in lexicographic
 *      order (as for the tail).  Not capturable in a hipGraph. ---- */
typedef struct { hpgmg_hip_level L; double h2inv, c1[8], c2[8]; } hpgmg_hip_brick_level;      /* c1 / c2: the level's Chebyshev coefficients per sweep */
int hpgmg_hip_brick_visit_supported(const hpgmg_hip_level *L, int brick);      /* 1: dim_i^3 cells = 2^3 or 4^3 bricks of 16^3 cells (brick = 16), 2^3 .. 8^3 bricks of 8^3 (brick = 8) */
int hpgmg_hip_brick_visit_max_sweeps(void);
int hpgmg_hip_brick_chain_max_levels(void);
/* ONE launch for the chain levels[0 .. n-1] (finest first; `below` = the level under levels[n-1]).
 * dir 0 (down): per level smooth, residual -> VECTOR_TEMP, restriction(next.R_id <- TEMP); levels[1..] start from a zero correction (mg.c:1153) and get
 *   their right-hand side from the level above inside the launch.  top_e_zero: levels[0].e counts as +0.0 too and is not read (its zero_vector is completed
 *   here: ghost zone and padding; the interior is stored at the end anyway); below_zero: zero_vector(below.e) at the end.
 * dir 1 (up): coarsest level first: e += P next.e (piecewise constant), smooth; levels[n-1] reads `below`, every finer level waits for the one under it.
 * dir 2 (n == 1): interpolation_fcycle (e = 0.0 e + P1 below.e, interpolation_p1.c:40-70, the ghost cells of apply_BCs_p1 formed on the fly), then dir 0 --
 *   every brick reads below.e, so below_zero must be 0 and the launch that visits `below` clears it (top_e_zero). */
int hpgmg_hip_brick_chain(int n, const hpgmg_hip_brick_level *levels, const hpgmg_hip_level *below, int sweeps, int variant, int smoother,
                          int e_id, int R_id, double a, double b, int dir, int brick, int top_e_zero, int below_zero);
/* ---- brick_wide.hip: the same for the 27-point and 4th-order operators (variants HPGMG_HIP_27PT_CC, HPGMG_HIP_FV4_*): bricks of 8^3 cells with a halo of the stencil's
 *      radius (faces + edges + corners / faces + edges), every shell cell published once per exchange and fetched by whichever neighbour needs it, apply_BCs_p2 / _v4
 *      formed on the LDS image after the exchange, interpolation_p2 / _v2 on the way up from an image of the brick's parents; smoother 0 Chebyshev, 1 GSRB (out of
 *      place, operators.27pt.c:126 / operators.fv4.c:178).  dir 0 / 1 and the two flags as hpgmg_hip_brick_chain.  Replaces per level visit the 13 + 2 + 1 (fv4 GSRB) launches of
 *      mg.c:1147-1153 with operators/gsrb.c:24-132, boundary_fv.c:262-569, residual.c, restriction.c and the 2 + 12 of mg.c:1160-1161 with interpolation_v2.c. ---- */
int hpgmg_hip_brick_wide_supported(const hpgmg_hip_level *L, int variant);
int hpgmg_hip_brick_wide_capacity(int variant, int smoother);
int hpgmg_hip_brick_wide_chain(int n, const hpgmg_hip_brick_level *levels, const hpgmg_hip_level *below, int sweeps, int variant, int smoother,
                               int e_id, int R_id, double a, double b, int dir, int top_e_zero, int below_zero);
long long hpgmg_hip_brick_visits(void);      /* level visits so far (tests) */
int hpgmg_hip_brick_visit_error(void);       /* 1: a poll of an earlier launch gave up after 2 s (a workgroup of the launch was not running): results are void */
int hpgmg_hip_brick_visit_error_clear(void); /* the host has dealt with it (synchronises) */
/* workgroups of the (variant, smoother) brick kernels this device holds at once (occupancy x CUs) less an eighth: a level with more bricks than this must not be
 * launched as bricks -- its workgroups wait for each other (HPGMG_TEST_BRICK_CAPACITY=<n> forces the raw figure: tests) */
int hpgmg_hip_brick_chain_capacity(int variant, int smoother, int brick);

/* ---- hipGraph segments (graph.hip): capture/replay of the launch-bound small-level part of a cycle.
 *      begin(key): first use of a key runs eagerly, second is captured, later ones are replayed
 *      (launchers return immediately while a replay segment is open; end() launches the graph).
 *      Host-synchronising entry points flush an open segment themselves. ---- */
void hpgmg_hip_graph_enable(int on);
int  hpgmg_hip_graph_enabled(void);
int  hpgmg_hip_graph_begin(long long key);
int  hpgmg_hip_graph_end(void);
int  hpgmg_hip_graph_is_open(void);
int  hpgmg_hip_graph_flush(void);
void hpgmg_hip_graph_reset(void);
void hpgmg_hip_graph_stats(long long out[3]);   /* eager, captured, replayed segment counts */

/* ---- transport over RCCL / xGMI (replaces the MPI calls of exchange_boundary.c:33-97,
 *      restriction.c:128-192, interpolation_p*.c:74-139 and the MPI_Allreduce of misc.c:276,324,373).
 *      One process per GPU; rank 0 makes the id, every rank calls init with it.  The two
 *      functions after init have exactly the hpgmg_transport callback signatures. ---- */
int  hpgmg_hip_rccl_unique_id(char *out128);
int  hpgmg_hip_rccl_init(const char *id128, int rank, int size);
void hpgmg_hip_rccl_finalize(void);
void hpgmg_hip_rccl_sendrecv(void *ctx, int nrecv, double *const *rbuf, const int *rsize, const int *rrank,
                             int nsend, double *const *sbuf, const int *ssize, const int *srank, int tag);
void hpgmg_hip_rccl_allreduce(void *ctx, double *vals, int n, int op, const int *ranks, int nranks);
/* hpgmg_transport.prepare_subset: EVERY rank of the job, same sets in the same order (MGBuild): a sub-communicator for the set by ncclCommSplit (the reference's
 * MPI_Comm_split per level, mg.c:985-993); reductions over the set are then ONE collective on it.  HPGMG_RCCL_SUBCOMM=0 (on every rank), a failed split on any
 * rank, or a set never announced: the all-to-all of 8-byte messages. */
void hpgmg_hip_rccl_prepare_subset(void *ctx, const int *ranks, int nranks);
long long hpgmg_hip_rccl_subset_collectives(void);      /* subset reductions done as one collective ... */
long long hpgmg_hip_rccl_subset_alltoalls(void);        /* ... and as an all-to-all (tests, bench.py) */
/* MAX of n (<= job size) host doubles over every rank of the communicator, in place (one ncclAllReduce; what the transport's allreduce uses for a
 * whole-job maximum, misc.c:324) */
int  hpgmg_hip_rccl_allreduce_max_world(double *vals, int n);
/* SUM (op 1) or MAX (op 0) of n (<= 16) host doubles over every rank, in place, as ONE ncclAllGather of the partials followed by a reduction in
 * rank order on the host: the association of misc.c:276,373's MPI_Allreduce(MPI_SUM) made deterministic (what the transport's allreduce uses
 * for a whole-job sum); _allgather_count = how many have run (tests) */
int  hpgmg_hip_rccl_allreduce_ordered_world(double *vals, int n, int op);
long long hpgmg_hip_rccl_allgather_count(void);

/* ---- a second transport for ONE node: peer copies between the ranks' device buffers through hipIpc memory handles, ordered against both ranks'
 *      streams by host functions on counters in a small POSIX shared-memory segment (no collective library, nothing staged through the host, no
 *      stream synchronisation); scalars through the same segment, reduced in rank order.  `name` = a shared-memory name ("/...") the ranks of a job agree on; rank 0 creates the segment.  Same callback signatures as
 *      the RCCL transport.  Works between GPUs of a node (xGMI) and between processes that share a GPU. ---- */
int  hpgmg_hip_ipc_init(const char *name, int rank, int size);
void hpgmg_hip_ipc_finalize(void);
void hpgmg_hip_ipc_sendrecv(void *ctx, int nrecv, double *const *rbuf, const int *rsize, const int *rrank,
                            int nsend, double *const *sbuf, const int *ssize, const int *srank, int tag);
void hpgmg_hip_ipc_allreduce(void *ctx, double *vals, int n, int op, const int *ranks, int nranks);
long long hpgmg_hip_ipc_message_count(void);

#ifdef __cplusplus
}
#endif
#endif

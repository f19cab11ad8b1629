/*
 * plugin_internal.h -- what the translation units of the operator plugin (operators_hip.c, halo_images.c) share: the per-level device
 * record, the timer / launch-check macros and the helpers that are not part of include/hpgmg_operators.h.  Not installed, not part of the
 * drop-in boundary; every symbol declared HP_INTERNAL is hidden in libhpgmg_fv.so.
 */
#ifndef HPGMG_PLUGIN_INTERNAL_H
#define HPGMG_PLUGIN_INTERNAL_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <stdint.h>
#include "hpgmg_level.h"
#include "hpgmg_operators.h"
#include "hpgmg_mg.h"
#include "hpgmg_hip.h"

int hpgmg_smooth_sweeps(void);
int hpgmg_gsrb_out_of_place(void);

#define HP_INTERNAL __attribute__((visibility("hidden")))

/* Timers of level->timers (reference level.h:162-196; printed by MGPrintTiming, mg.c:54-161).  Launches are asynchronous, so
 * there are three ways to fill them:
 *   HPGMG_TIMERS=host   (default for library users / bench.py) host clock around the launch calls: costs nothing, but the rows only
 *                       say where the HOST thread spent its time;
 *   HPGMG_TIMERS=device (default of the hpgmg-fv executable, hpgmg_set_timer_mode(1)) a hipEvent pair on the launch stream around
 *                       every operator: device time per operator class and level, settled when the table is printed or reset;
 *   HPGMG_TIMERS=sync   (or HPGMG_SYNC_TIMERS=1) synchronise around every operator: exact wall time, serialises host and device.
 * With HPGMG_ROCTX=1 every timed operator is also a roctx range "<dim>^3 <operator>" (rocprofv3 --marker-trace). */
enum { TIMERS_HOST = 0, TIMERS_DEVICE = 1, TIMERS_SYNC = 2 };
#define TICK(L, FIELD, WHAT) const hpgmg_tick tick_ = hpgmg_tick_begin((L), &(L)->timers.FIELD, WHAT)
#define TOCK() hpgmg_tick_end(tick_)
/* every device call of the plugin goes through HIP_OK: the operators still waiting in the lazy queue (operators_hip.c) are issued first, so
 * whatever runs next sees the state they leave */
HP_INTERNAL void hp_lazy_flush(void);
#define HIP_OK(call) do { hp_lazy_flush(); int e_ = (call); if (e_) { fprintf(stderr, "hpgmg: %s failed (%d): %s\n", #call, e_, hpgmg_hip_last_error()); abort(); } } while (0)

/* ---------------------------------------------------------------- per-level device record */
#define MAX_LISTS 32
typedef struct {
  hpgmg_hip_level dev;         /* what the kernels receive */
  double **d_box_base;  int *d_box_low;  int *d_box_nbr;  int all_faces_local;
  double  *seen_v0;     int seen_nv, seen_boxes;  /* detects create_vectors() re-allocation */
  struct { const blockCopy_type *host; int n; blockCopy_type *dev; } lists[MAX_LISTS];
  int num_lists;
  hpgmg_hip_bc_entry *d_bc[STENCIL_MAX_SHAPES]; int n_bc[STENCIL_MAX_SHAPES];   /* boundary-condition blocks with their geometry worked out */
  int *d_fv4_special; int n_fv4_special;       /* fv4 red + black: cells on internal box faces next to a domain wall (box, i, j, k); n < 0: not built */
  hpgmg_hip_bc_entry *d_bc_k; int n_bc_k, bc_k_local;      /* the blocks of the stencil's shape whose domain normal has a k component (fv4 red + black pre-pass); n_bc_k < 0: not built */
  int bc_sources_local[STENCIL_MAX_SHAPES];    /* 1: every entry reads cells of local boxes' interiors only (no exchange needed before the conditions) */
  int *krylov_pinned;          /* iterations of device-side bottom solves not yet folded into level->Krylov_iterations */
  double *pair_scratch; double **d_pair_base;   /* two private vectors per box for the fused Chebyshev sweep pairs */
  float *coef32; float **d_coef32_base; int coef32_valid;   /* fp32 copies of Dinv, alpha, beta_* for the mixed-precision smoother */
  int lexicographic;           /* -1 unknown, else whether local box b sits at (b % nb, (b / nb) % nb, b / nb^2) and all boxes are local */
  int *d_restrict_map;         /* fused residual + restriction: per fine box the coarse box and the coarse cell under its first cell (device) */
  struct pair_halo *halo;      /* sweep pairs across rank boundaries: brick shape, message plans, deep halos (NULL: not built / not applicable) */
  int halo_state;              /* 0 not examined, 1 usable, -1 this level cannot use it */
  struct halo_images *img;     /* 27-point / fv4 across rank boundaries: images of the neighbouring ranks' boxes (halo_images.c; NULL: not built) */
  int img_state;               /* 0 not examined, 1 usable, -1 this level cannot use them */
  int img_active;              /* the stencil launch being prepared reads neighbouring ranks' cells from the images: it gets hp_stencil_dev() */
} backend_t;

/* ---- halo of a sweep pair (kernels/cheby_pair.hpp, REMOTE variants) ----------------------------------------------------------
 * One message per neighbouring rank per sweep PAIR instead of one per sweep (reference: exchange_boundary before every sweep,
 * chebyshev.c:45-46).  Plan FIRST (first pair of a smooth()): x0 two cells deep on faces + one cell on the brick's edges, xm1 and the
 * right-hand side one cell deep; plan NEXT: the same without the right-hand side; plan COEF (once per operator rebuild): the normal
 * beta component one index beyond the ghost zone on the high faces.  Inside a message regions are ordered by the sender's global box
 * id, then the direction seen from the sender, then the item -- both sides derive that order independently, like level.c does. */
enum { HALO_FIRST = 0, HALO_NEXT = 1, HALO_COEF = 2, HALO_PLANS = 3 };
typedef struct {
  int n_send, n_recv;                           /* regions */
  hpgmg_hip_halo_entry *d_send, *d_recv;        /* device copies of the region lists */
  int n_sp, n_rp;                               /* messages: peers this rank sends to / receives from */
  int *sp_rank, *rp_rank, *sp_size, *rp_size;   /* doubles per message */
  double **sp_ptr, **rp_ptr;                    /* start of each message inside the rank's send / receive buffer */
} halo_plan;
typedef struct pair_halo {
  int brick[3], rem[6];
  halo_plan plan[HALO_PLANS];
  double *sendbuf, *recvbuf, *deep, *deep_beta;
  int coef_valid;
} pair_halo;

/* ---- images of the neighbouring ranks' boxes (halo_images.c) -------------------------------------------------------------------
 * The LDS-tiled kernels of the 27-point and fv4 operators, their one-pass red + black forms and the fused residual passes read a cell
 * outside a box where it LIVES (kernels/common.hpp gf_column: box_nbr leads to the neighbouring box).  With boxes on other ranks that
 * stops working -- unless the neighbouring rank's box is there: an IMAGE is a box of this level's layout (same strides, numVectors + 2
 * private vectors) that stands in the kernels' box table behind the rank's own boxes and holds, of the box it stands for, exactly the
 * cells within `depth` cells of this rank's brick.  One message per neighbouring rank and stencil launch refreshes them (x `depth` cells
 * deep; the right-hand side one cell deep once per smooth(); the coefficient vectors with their ghost zones -- the OWNER's view, which
 * differs outside the domain because extrapolate_betas works with box-relative normals, boundary_fv.c:573-681 -- once per operator
 * rebuild).  The kernels run unchanged, on the table with the images; what the owner would compute on the cells next to the face -- the
 * intermediate vector of a red + black pass -- is recomputed here from the same inputs with the same expression, so the results stay bit
 * identical while a GSRB sweep needs ONE exchange instead of the reference's two (gsrb.c:30-33). */
enum { IMG_PLAN_COEF = 6, IMG_PLANS = 7 };      /* plans 0..2: x 1..3 cells deep; 3..5: the same + the right-hand side one cell deep */
typedef struct {
  int n_send, n_recv;
  hpgmg_hip_halo_entry *d_send, *d_recv;
  int n_sp, n_rp;
  int *sp_rank, *rp_rank, *sp_size, *rp_size;
  long long *sp_off, *rp_off;                   /* start of each message in the send / receive buffer (doubles) */
  int built;
} image_plan;
typedef struct halo_images {
  int n_real, n_img, n_all;                     /* own boxes, images, both: the kernels' box table lists the own boxes first */
  int *gid;                                     /* [n_all] global box ids */
  int *h_nbr;                                   /* [6 n_all] host copy of the face-neighbour table */
  int lo[3], n[3];                              /* this rank's brick, in boxes */
  int *brick_lo, *brick_n;                      /* [3 ranks] every rank's brick (n = 0: the rank owns no box of this level) */
  int depth_max;                                /* images hold the cells within this distance of the brick (stencil radius + 1) */
  double *storage, *scratch;                    /* the images' level vectors and private vectors */
  double **d_box_base, **d_pair_base; int *d_box_low, *d_box_nbr;     /* DEVICE tables over n_all boxes */
  hpgmg_hip_level dev, dev_all;                 /* kernels: num_boxes = n_real (work on the own boxes) / n_all (the fv4 pre-pass also works on images) */
  image_plan plan[IMG_PLANS];
  double *sendbuf, *recvbuf; size_t send_cap, recv_cap;
  double **ptr_tmp; int ptr_cap;
  hpgmg_hip_bc_entry *d_bc; int n_bc, n_bc_own; /* boundary entries of the stencil's shape: own boxes (sources in the table), then images */
  hpgmg_hip_bc_entry *d_bc_k; int n_bc_k;       /* those whose domain normal has a k component */
  int *d_special; int n_special;                /* fv4 red + black: cells of images (and own boxes) on internal faces next to a wall */
  int coef_valid, shape;
  double *seen_v0; int seen_nv;
} halo_images;

/* ---- run-time switches: one table (plugin_switches.c) ---- */
typedef enum { SW_GHOST_FREE = 0, SW_ONE_LAUNCH_GHOSTS, SW_OVERLAP, SW_PAIR_REMOTE, SW_IMAGES, SW_FUSED_SWEEPS, SW_PAIR_MIN_CELLS, SW_FUSED_RESIDUAL, SW_FUSED_TAIL,
               SW_FUSED_FTAIL, SW_FUSED_BOTTOM, SW_SMALL_FUSED, SW_SMALL_VTAIL, SW_SMALL_27PT_GSRB, SW_SMALL_OPS, SW_LAZY, SW_LAZY_REPORT, SW_TEMP_SCRATCH,
               SW_FV4_NO_EXACT_RB, SW_GRAPH, SW_SMOOTHER_PRECISION, SW_DEFER_NORM, SW_BRICK_VISITS, SW_BRICK_SIZE, SW_BRICK_MIN_DIM, SW_BRICK_FSTEP, SW_BRICK_CHAIN, SW_BRICK_WIDE, SW_BRICK_WIDE_MAX_DIM, SW_BRICK_WIDE_TAIL_DIM, SW_FTAIL_MAX_DIM, SW_COUNT } hp_switch_id;
HP_INTERNAL long long hp_switch(hp_switch_id id);                 /* the value in force: the setter's, else the environment's, else the default */
HP_INTERNAL void hp_switch_set(hp_switch_id id, long long value);

HP_INTERNAL backend_t *hp_backend_of(level_type *L);
HP_INTERNAL const blockCopy_type *hp_mirror(level_type *owner, const blockCopy_type *host, int n);
HP_INTERNAL int  hp_variant(void);
HP_INTERNAL int  hp_ghost_free_mode(void);
HP_INTERNAL int  hp_box_rank_at(const level_type *L, int bi, int bj, int bk);
HP_INTERNAL void hp_ensure_pair_scratch(level_type *L, backend_t *B);
/* one boundary entry from a block of a box that sits at position bpos (in boxes) and index `box` of the kernels' table; find(ctx, gid) = the
 * table index of a box or -1.  Returns 0 when the block runs along the face of a box that is not in the table (it then reads the own ghost zone). */
HP_INTERNAL int  hp_bc_entry_from_block(const level_type *L, int box, const int bpos[3], const int lo[3], const int len[3], int subtype,
                                        int (*find)(void *, int), void *ctx, hpgmg_hip_bc_entry *o);
HP_INTERNAL int  hp_images_ready(level_type *L, backend_t *B);
HP_INTERNAL void hp_images_refresh(level_type *L, backend_t *B, int scr, int id, int depth, int rhs_id, int bc_order);
HP_INTERNAL int  hp_images_refresh_begin(level_type *L, backend_t *B, int scr, int id, int depth, int rhs_id, int bc_order);
HP_INTERNAL void hp_images_refresh_end(void);
HP_INTERNAL int  hp_overlap_enabled(void);
HP_INTERNAL void hp_overlap_counted(void);
HP_INTERNAL void hp_images_bcs(level_type *L, backend_t *B, int scr, int id, int bc_order, int part);
HP_INTERNAL const int *hp_images_fv4_special(level_type *L, backend_t *B, int *n_out);
HP_INTERNAL const hpgmg_hip_bc_entry *hp_images_bc_k(level_type *L, backend_t *B, int *n_out);
HP_INTERNAL void hp_images_release(backend_t *B);
HP_INTERNAL void hp_images_invalidate_coefficients(backend_t *B);
extern HP_INTERNAL long long hp_images_exchanges;      /* image refreshes so far (tests) */
static inline const hpgmg_hip_level *hp_stencil_dev(backend_t *B) { return (B->img && B->img_active) ? &B->img->dev : &B->dev; }


/* ---- what the translation units of the plugin share (operators_hip.c was one file until round 4) ---- */
extern HP_INTERNAL void *hp_comm_stream, *hp_ev_packed, *hp_ev_landed;
extern HP_INTERNAL long long hp_overlap_count;
HP_INTERNAL double hp_now(void);
HP_INTERNAL void hp_transport_phase(const communicator_type *recv_side, const communicator_type *send_side, int tag);
HP_INTERNAL int hp_exchange_and_bcs_one_launch(level_type *L, int id, int shape, int order, int with_copies);
HP_INTERNAL int hp_ghosts_for_stencil(level_type *L, int id, int out_id);
HP_INTERNAL int hp_overlap_begin(level_type *L, int id);
HP_INTERNAL void hp_overlap_end(void);
HP_INTERNAL void hp_no_kernel(const char *what);
HP_INTERNAL const hpgmg_hip_bc_entry *hp_bc_entries(level_type *L, int shape, int *n_out);
HP_INTERNAL const hpgmg_hip_bc_entry *hp_bc_entries_k(level_type *L, int *n_out, int *all_local_out);
HP_INTERNAL int hp_vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg);
HP_INTERNAL void hp_coef32_invalidate(level_type *L);
HP_INTERNAL int hp_interp_smooth_fused(level_type *Lf, int e_id, int R_id, level_type *Lc, double a, double b, int exact_state);
HP_INTERNAL void hp_do_smooth(level_type *L, int x_id, int rhs_id, double a, double b, int temp_dead);
HP_INTERNAL void hp_do_residual(level_type *L, int res_id, int x_id, int rhs_id, double a, double b);
HP_INTERNAL void hp_do_apply_op(level_type *L, int Ax_id, int x_id, double a, double b);
HP_INTERNAL int hp_pair_halo_ready(level_type *L, backend_t *B);
HP_INTERNAL int hp_pair_halo_begin(level_type *L, backend_t *B, int first, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id);
HP_INTERNAL void hp_do_restriction(level_type *Lc, int id_c, level_type *Lf, int id_f, int type);
HP_INTERNAL int hp_residual_restrict_zero_fused(level_type *Lc, int id_c, level_type *Lf, int res_id, int x_id, int rhs_id, double a, double b, int zero_id);
HP_INTERNAL int hp_norm_scale_restrict_fused(level_type *L, int F_id, int R_id, level_type *Lc, double *norm_out);
HP_INTERNAL void hp_do_interpolation_vcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c);
HP_INTERNAL void hp_do_zero_vector(level_type *L, int id);
HP_INTERNAL void hp_do_add_vectors(level_type *L, int c, double sa, int a, double sb, int b);
HP_INTERNAL void hp_do_mul_vectors(level_type *L, int c, double s, int a, int b);
HP_INTERNAL void hp_do_scale_vector(level_type *L, int c, double s, int a);
HP_INTERNAL double hp_allreduce_scalar(level_type *L, double v, int op);
HP_INTERNAL double hp_do_dot(level_type *L, int a, int b);
HP_INTERNAL double hp_do_norm(level_type *L, int a);
HP_INTERNAL void hp_small_ops_forget(void);
HP_INTERNAL const int *hp_restrict_map_of(level_type *Lf, backend_t *Bf);      /* device table: per fine box the coarse box and the coarse cell under its first cell; NULL when a parent is not local */

#define BLAS1(call) do { TICK(L, blas1, "BLAS1"); HIP_OK(call); TOCK(); } while (0)
#define STENCIL_WITH_GHOSTS(L, id, out_id, TIMER, CALL) do {                                                     \
    hp_backend_of(L)->img_active = 0;                                                                    \
    if (hp_overlap_begin(L, id)) {                                                                          \
      TICK(L, TIMER, #TIMER " (overlapped with the halo exchange)");                                     \
      hpgmg_hip_set_defer_mode(1); HIP_OK(CALL);                                                         \
      hp_overlap_end();                                                                                     \
      hpgmg_hip_set_defer_mode(2); HIP_OK(CALL); hpgmg_hip_set_defer_mode(0);                            \
      TOCK();                                                                                            \
    } else {                                                                                             \
      const int two_parts_ = hp_ghosts_for_stencil(L, id, out_id);   /* 1: the images' refresh runs on the exchange stream */ \
      TICK(L, TIMER, #TIMER);                                                                            \
      if (two_parts_) {                                                                                  \
        hpgmg_hip_set_tile_part(1); HIP_OK(CALL);                                                        \
        hp_images_refresh_end();                                                                         \
        hpgmg_hip_set_tile_part(2); HIP_OK(CALL); hpgmg_hip_set_tile_part(0);                            \
      } else HIP_OK(CALL);                                                                               \
      TOCK();                                                                                            \
    } } while (0)
#define PAIR_REMOTE_LAUNCH(OVERLAPPED, DISCARD_X1, CALL) do {                                                   \
    pair_halo *H_ = B->halo;                                                                                   \
    if (OVERLAPPED) {                                                                                          \
      hpgmg_hip_pair_set_halo(H_->brick, H_->rem, H_->deep, H_->deep_beta); hpgmg_hip_set_tile_part(1);        \
      if (DISCARD_X1) hpgmg_hip_pair_discard_x1();                                                             \
      HIP_OK(CALL);                                                                                            \
      hp_overlap_end();                                                                                           \
      hpgmg_hip_pair_set_halo(H_->brick, H_->rem, H_->deep, H_->deep_beta); hpgmg_hip_set_tile_part(2);        \
      if (DISCARD_X1) hpgmg_hip_pair_discard_x1();                                                             \
      HIP_OK(CALL);                                                                                            \
      hpgmg_hip_set_tile_part(0);                                                                              \
    } else {                                                                                                   \
      hpgmg_hip_pair_set_halo(H_->brick, H_->rem, H_->deep, H_->deep_beta);                                    \
      if (DISCARD_X1) hpgmg_hip_pair_discard_x1();                                                             \
      HIP_OK(CALL);                                                                                            \
    } } while (0)

#endif

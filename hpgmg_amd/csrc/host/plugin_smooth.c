/*
 * plugin_smooth.c -- smooth() (operators/chebyshev.c, gsrb.c, jacobi.c) in all its forms: single sweeps, sweep pairs, red + black passes, the single-launch legs of the small levels; residual() / apply_op().
 * Part of the operator plugin (see operators_hip.c); no arithmetic on vector data happens here.
 */
#include "plugin_internal.h"

/* ---------------------------------------------------------------- smoothers */
static void cheby_coefficients(const level_type *L, int degree, double *c1, double *c2) { /* chebyshev.c:22-40 */
  double beta = 1.000 * L->dominant_eigenvalue_of_DinvA, alpha = 0.125000 * beta;
  double theta = 0.5 * (beta + alpha), delta = 0.5 * (beta - alpha), sigma = theta / delta, rho_n = 1 / sigma;
  int s;
  c1[0] = 0.0; c2[0] = 1 / theta;
  for (s = 1; s < degree; s++) { double rho_nm1 = rho_n; rho_n = 1.0 / (2.0 * sigma - rho_nm1); c1[s] = rho_n * rho_nm1; c2[s] = rho_n * 2.0 / delta; }
}

/* Both legs of a V-cycle over a chain of tiny levels in one launch each (kernels/tail.hip). */
/* fold the iteration counts of device-side bottom solves into level->Krylov_iterations (mg.c:156 prints it) */
void hpgmg_level_sync_counters(level_type *L) {
  hpgmg_hip_timer_flush();                       /* pending device timers land in level->timers before they are read or reset */
  hpgmg_level_ext *X = hpgmg_level_ext_get(L);
  backend_t *B = (backend_t *)X->backend;
  if (!B || !B->krylov_pinned) return;
  HIP_OK(hpgmg_hip_sync());
  L->Krylov_iterations += *B->krylov_pinned;
  *B->krylov_pinned = 0;
}

/* `leg`: enum hpgmg_leg (include/hpgmg_operators.h) -- DOWN 0, UP 1, VCYCLE 2, BOTTOM 3, FCYCLE_TAIL 4, FCYCLE_TAIL_ASK 5, FCYCLE_STEP 6; + HPGMG_LEG_ASK (16): only answer */
int hpgmg_vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg) { hp_lazy_flush(); return hp_vcycle_legs_fused(levels, n, e_id, R_id, a, b, leg); }
/* The same for the 27-point / fv2 / fv4 plugins, leg 2 only (smooth ... bottom solve ... smooth as one launch): every level of the chain is ONE
 * box whose vectors fit the LDS (kernels/stencil.hip: small_vtail_kernel).  `7 8`: the levels of 8^3, 4^3, 2^3 (and 1^3) cells. */
/* On by default except for the 27-point plugin with GSRB (HPGMG_SMALL_VTAIL=0 / 1, hpgmg_set_small_vtail(); bit-identical, tested both ways).
 * Measured on MI355X, `7 8` F-cycles with it on / off: fv4 GSRB 7.72 / 7.77 ms, fv4 Chebyshev 8.13 / 8.23, fv2 GSRB 6.35 / 6.56, fv2 Chebyshev
 * 6.65 / 6.91, 27-point Chebyshev 4.72 / 4.84 -- and 27-point GSRB 4.04 / 3.94: that plugin's one-launch red + black box kernel beats two half
 * sweeps of the generic form.  The first version (1024 lanes) was slower everywhere: the bottom solve's 240 registers per lane spilled into
 * scratch memory under the 128-register cap; with 512 lanes the launch of 8^3 + 4^3 + 2^3 levels takes 169 instead of 191 us (fv4 GSRB;
 * tools/exp_vtail_timeline.py): 4 x 24 us of smoothing, 28 us of bottom solve, the rest image traffic and interpolation. */
static int bottom_solve_fused_impl(level_type *L, int e_id, int R_id, double a, double b, double want, int ask_only);
static level_type *tail_books_on = NULL;      /* the level whose timers take the tail launch when bricks were visited above it (else its own first level) */
static int tail_follows_bricks = 0;      /* the chain handed to small_vtail_fused is what is left below levels visited as bricks (hp_vcycle_legs_fused) */
static long long small_vtails = 0;
long long hpgmg_small_vtails(void) { return small_vtails; }
void hpgmg_set_small_vtail(int on) { hp_switch_set(SW_SMALL_VTAIL, (on == 2) ? 2 : (on ? 1 : 0)); }      /* 0 off, 1 on for every plugin, 2 the default (not for 27-point GSRB) */
static int small_vtail_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int legs, int ask_only) {
  hpgmg_config cfg;
  hpgmg_hip_small_tail_args T;
  int l;
  const int small_vtail_on = (int)hp_switch(SW_SMALL_VTAIL);      /* 2: the default */
  if (!small_vtail_on) return 0;
  hpgmg_get_config(&cfg);
  if (small_vtail_on == 2 && cfg.op == HPGMG_OP_27PT && cfg.smoother == HPGMG_SMOOTH_GSRB && !tail_follows_bricks) return 0;      /* (below brick launches it is the tail: the legs must stay whole) */
  const int sweeps = hpgmg_smooth_sweeps();
  if (n < 2 || n > HPGMG_HIP_SMALL_TAIL_MAX_LEVELS || sweeps < 1 || sweeps > 8 || (sweeps & 1) || hp_switch(SW_GRAPH)) return 0;      /* (captured segments: the argument block's upload is not capturable) */
  if (cfg.smoother != HPGMG_SMOOTH_CHEBY && cfg.smoother != HPGMG_SMOOTH_GSRB && cfg.smoother != HPGMG_SMOOTH_JACOBI) return 0;
  memset(&T, 0, sizeof T);
  T.n = n; T.legs = legs; T.mode = (cfg.smoother == HPGMG_SMOOTH_CHEBY) ? 0 : (cfg.smoother == HPGMG_SMOOTH_GSRB ? 1 : 2);
  T.sweeps = sweeps; T.out_of_place = (T.mode == 1) ? hpgmg_gsrb_out_of_place() : 0;
  T.e_id = e_id; T.R_id = R_id; T.krylov_base = hpgmg_vectors_reserved(); T.a = a; T.b = b; T.want = MG_DEFAULT_BOTTOM_NORM;
  const int shape = stencil_get_shape();
  for (l = 0; l < n; l++) {
    level_type *L = levels[l];
    if (!L->active || L->num_my_boxes != 1 || L->boxes_in.i * L->boxes_in.j * L->boxes_in.k != 1) return 0;
    if (L->boundary_condition.type != BC_DIRICHLET || L->dim.i != L->dim.j || L->dim.i != L->dim.k) return 0;
    if (l > 0 && 2 * L->dim.i != levels[l - 1]->dim.i) return 0;
    {
      communicator_type *C = &L->exchange_ghosts[shape], *CB = &L->exchange_ghosts[STENCIL_SHAPE_BOX];
      if (C->num_sends + C->num_recvs > 0 || C->num_blocks[0] || C->num_blocks[1] || C->num_blocks[2]) return 0;      /* one box: nothing to exchange */
      if (CB->num_sends + CB->num_recvs > 0 || CB->num_blocks[0] || CB->num_blocks[1] || CB->num_blocks[2]) return 0;
    }
    backend_t *B = hp_backend_of(L);
    hpgmg_hip_small_tail_level *v = &T.lv[l];
    v->L = B->dev;
    v->h2inv = 1.0 / (L->h * L->h);
    v->n_bc = L->boundary_condition.num_blocks[shape];
    v->bc_list = v->n_bc ? hp_mirror(L, L->boundary_condition.blocks[shape], v->n_bc) : NULL;
    if (cfg.op == HPGMG_OP_27PT) v->bc_kind = (L->box_dim < 2) ? 1 : 2;                                   /* as small_level_try / apply_BCs */
    else if (cfg.op == HPGMG_OP_FV2 || L->box_dim < 4) { v->bc_kind = (L->box_dim < 2) ? 1 : 3; v->zero_first = (v->bc_kind == 3 && L->box_ghosts > 1); }
    else { v->bc_kind = 4; v->zero_first = (L->box_ghosts > 2); }
    /* the conditions interpolation_vcycle applies to THIS level's correction before the level above reads it: apply_BCs_p2 (27-point) /
     * apply_BCs_v2 (fv2, fv4) over STENCIL_SHAPE_BOX */
    v->n_ibc = L->boundary_condition.num_blocks[STENCIL_SHAPE_BOX];
    v->ibc_list = v->n_ibc ? hp_mirror(L, L->boundary_condition.blocks[STENCIL_SHAPE_BOX], v->n_ibc) : NULL;
    if (cfg.op == HPGMG_OP_27PT) v->ibc_kind = (L->box_dim < 2) ? 1 : 2;
    else { v->ibc_kind = (L->box_dim < 2) ? 1 : 3; v->ibc_zero_first = (v->ibc_kind == 3 && L->box_ghosts > 1); }
    if (v->n_bc > 32 || v->n_ibc > 32) return 0;
    if (l + 1 < n) {
      if (T.mode == 0) { if (L->dominant_eigenvalue_of_DinvA <= 0.0) return 0; cheby_coefficients(L, sweeps, v->c1, v->c2); }
      if (T.mode == 2) { int q; for (q = 0; q < sweeps; q++) v->c2[q] = 2.0 / 3.0; }
    } else if (legs & 2) {
      /* solvers.c:77-87: the fused solve is the Dirichlet one (no mean to remove); the Krylov vectors must exist */
      if (L->must_subtract_mean != 0) return 0;
      if ((long long)L->dim.i * L->dim.j * L->dim.k > hpgmg_hip_bottom_bicgstab_max_cells()) return 0;
      if (L->numVectors < hpgmg_vectors_reserved() + IterativeSolver_NumVectors()) return 0;
      if (!B->krylov_pinned) { B->krylov_pinned = (int *)hpgmg_hip_host_malloc(64); if (B->krylov_pinned) *B->krylov_pinned = 0; }
      if (!B->krylov_pinned) return 0;
      T.krylov_iterations = B->krylov_pinned;
    }
  }
  if (hpgmg_hip_small_vtail_lds_doubles(&T) > hpgmg_hip_small_vtail_lds_limit()) return 0;
  if (ask_only) return 1;
  TICK(tail_books_on ? tail_books_on : levels[0], smooth, "fused V-cycle tail (levels of one box)");
  HIP_OK(hpgmg_hip_small_vtail(&T, hp_variant()));
  TOCK();
  small_vtails++;
  return 1;
}
void hpgmg_set_fused_tail(int on) { hp_switch_set(SW_FUSED_TAIL, on ? 1 : 0); }      /* tests: 0 = every operator of the small levels as its own launch(es) */
void hpgmg_set_fused_bottom(int on) { hp_switch_set(SW_FUSED_BOTTOM, on ? 1 : 0); }  /* tests: 0 = the bottom solve driven from the host (host/solvers.c BiCGStab through the operators) */
void hpgmg_set_brick_visits(int on) { hp_switch_set(SW_BRICK_VISITS, on ? 1 : 0); if (on == 8 || on == 16) hp_switch_set(SW_BRICK_SIZE, on); }      /* 0 off, 1 on, 8 / 16: on with bricks of that side */
long long hpgmg_brick_visits(void) { return hpgmg_hip_brick_visits(); }
void hpgmg_set_brick_wide(int on) { hp_switch_set(SW_BRICK_WIDE, on ? 1 : 0); }      /* 0: the 27-point / fv4 plugins visit their launch-bound levels launch by launch (tests) */
void hpgmg_set_brick_chains(int on) { hp_switch_set(SW_BRICK_CHAIN, on ? 1 : 0); }      /* 0: one launch per level visit instead of one per V-cycle leg (tests) */
/* what the kernels that address cells by global coordinate need of a level (tail.hip, brick_visit.hip): a cubic Dirichlet domain whose boxes are all
 * here, all faces local, local box b at lexicographic position b */
static int dense_level_ok(level_type *L) {
  backend_t *B = hp_backend_of(L);
  const int nb = L->dim.i / L->box_dim;
  int bx;
  if (!L->active || L->num_my_boxes < 1 || !B->all_faces_local) return 0;
  if (L->boundary_condition.type != BC_DIRICHLET || L->dim.i != L->dim.j || L->dim.i != L->dim.k) return 0;
  if (L->num_my_boxes != nb * nb * nb) return 0;
  for (bx = 0; bx < L->num_my_boxes; bx++) {
    const box_type *X = &L->my_boxes[bx];
    if (X->low.i != (bx % nb) * L->box_dim || X->low.j != ((bx / nb) % nb) * L->box_dim || X->low.k != (bx / (nb * nb)) * L->box_dim) return 0;
  }
  return 1;
}
/* How many leading levels of the chain are visited as bricks of 8^3 / 16^3 cells, one launch per visit (kernels/brick_visit.hip): the levels of 64^3 / 32^3
 * cells above the single-workgroup tail.  0: none. */
static int brick_op_is_wide(const hpgmg_config *cfg) { return cfg->op == HPGMG_OP_27PT || cfg->op == HPGMG_OP_FV4; }
static long long brick_capacity_refusals = 0;
long long hpgmg_brick_capacity_refusals(void) { return brick_capacity_refusals; }      /* level visits left to the launch-by-launch path because the device does not hold that many bricks at once (tests) */
static int brick_prefix(level_type **levels, int n, const hpgmg_config *cfg) {
  const int sweeps = hpgmg_smooth_sweeps();
  int k = 0;
  const int wide = brick_op_is_wide(cfg);      /* 27-point / fv4: kernels/brick_wide.hip (bricks of 8^3 only, Chebyshev or out-of-place GSRB) */
  if (!hp_switch(SW_BRICK_VISITS) || !hp_switch(SW_FUSED_TAIL) || hp_switch(SW_GRAPH)) return 0;
  if (wide ? (!hp_switch(SW_BRICK_WIDE) || (cfg->smoother != HPGMG_SMOOTH_CHEBY && cfg->smoother != HPGMG_SMOOTH_GSRB) || (cfg->smoother == HPGMG_SMOOTH_GSRB && !hpgmg_gsrb_out_of_place()))
           : (!hp_ghost_free_mode() || cfg->op != HPGMG_OP_7PT)) return 0;
  if (sweeps < 1 || sweeps > hpgmg_hip_brick_visit_max_sweeps() || (sweeps & 1)) return 0;
  while (k + 1 < n) {
    level_type *L = levels[k];
    /* where the single-workgroup tail takes over: the 7-point tail holds levels up to 16^3; the tail of the other plugins starts at the first level of ONE box */
    const int fits_tail = wide ? (L->num_my_boxes == 1 && L->boxes_in.i * L->boxes_in.j * L->boxes_in.k == 1 && L->dim.i <= (int)hp_switch(SW_BRICK_WIDE_TAIL_DIM))
                               : ((long long)L->dim.i * L->dim.j * L->dim.k <= hpgmg_hip_tail_max_cells());
    if ((wide || L->dim.i < (int)hp_switch(SW_BRICK_MIN_DIM)) && fits_tail) break;
    if (wide && L->dim.i > (int)hp_switch(SW_BRICK_WIDE_MAX_DIM)) return 0;      /* (a tuning switch: all three launch-bound levels pay, profiles/r06d_ab_wide_max.txt) */
    if (!dense_level_ok(L) || !dense_level_ok(levels[k + 1]) || 2 * levels[k + 1]->dim.i != L->dim.i) return 0;
    const int wide_brick = wide ? hpgmg_hip_brick_wide_supported(&hp_backend_of(L)->dev, hp_variant()) : 0;      /* 8, 4 (the level of 4^3 cells) or 0 */
    if (wide ? !wide_brick
             : !hpgmg_hip_brick_visit_supported(&hp_backend_of(L)->dev, (int)hp_switch(SW_BRICK_SIZE))) { if (fits_tail) break; return 0; }
    { /* every brick of a launch must be running at once: more bricks than the device holds of this kernel = the launch-by-launch path (kernels/brick_visit.hip) */
      const int brick = wide ? wide_brick : (int)hp_switch(SW_BRICK_SIZE), side = L->dim.i / brick;
      const int capacity = wide ? hpgmg_hip_brick_wide_capacity(hp_variant(), cfg->smoother) : hpgmg_hip_brick_chain_capacity(hp_variant(), cfg->smoother, brick);
      if (side * side * side > capacity) { brick_capacity_refusals++; if (fits_tail) break; return 0; }
    }
    if (L->dominant_eigenvalue_of_DinvA <= 0.0 && cfg->smoother == HPGMG_SMOOTH_CHEBY) return 0;
    k++;
  }
  return k;
}
/* One launch of bricks for levels[first .. first + count - 1] (dir 0 down, 1 up, 2 interpolation_fcycle + down), or one per level when chains are off.
 * `top`: the level whose timers take the launches -- the one whose V-cycle this is, like the tail below it (mg.c books the whole fused cycle on that level's Total) */
static void brick_chain(level_type *top, level_type **levels, int first, int count, const hpgmg_config *cfg, int e_id, int R_id, double a, double b, int dir, int top_e_zero, int below_zero) {
  const int sweeps = hpgmg_smooth_sweeps(), chain = (int)hp_switch(SW_BRICK_CHAIN), max_n = (chain == 1 || (chain == 2 && dir != 1) || (chain == 3 && dir == 1)) ? hpgmg_hip_brick_chain_max_levels() : 1;
  int done = 0;
  while (done < count) {
    hpgmg_hip_brick_level lv[4];
    int n = count - done, j, s;
    if (n > max_n) n = max_n;
    if (brick_op_is_wide(cfg)) {
      /* the levels of one launch are cut into bricks of one size: the level of 4^3 cells (one brick of 4^3) goes on its own */
      const int v = hp_variant();
      if (dir == 1) { const int b0 = hpgmg_hip_brick_wide_supported(&hp_backend_of(levels[first + count - done - 1])->dev, v);
                      int m = 1; while (m < n && hpgmg_hip_brick_wide_supported(&hp_backend_of(levels[first + count - done - 1 - m])->dev, v) == b0) m++; n = m; }
      else          { const int b0 = hpgmg_hip_brick_wide_supported(&hp_backend_of(levels[first + done])->dev, v);
                      int m = 1; while (m < n && hpgmg_hip_brick_wide_supported(&hp_backend_of(levels[first + done + m])->dev, v) == b0) m++; n = m; }
    }
    /* down: the finest levels first; up: the coarsest levels first */
    const int lo = (dir == 1) ? first + count - done - n : first + done;
    for (j = 0; j < n; j++) {
      level_type *L = levels[lo + j];
      lv[j].L = hp_backend_of(L)->dev; lv[j].h2inv = 1.0 / (L->h * L->h);
      for (s = 0; s < 8; s++) lv[j].c1[s] = lv[j].c2[s] = 0.0;
      if (cfg->smoother == HPGMG_SMOOTH_CHEBY) cheby_coefficients(L, sweeps, lv[j].c1, lv[j].c2);
    }
    const int is_first_launch = (done == 0), is_last_launch = (done + n == count);
    TICK(top, smooth, dir == 1 ? "level visits, up (bricks: interpolation + smooth per level, one launch)" :
                      (dir == 0 ? "level visits, down (bricks: smooth + residual + restriction per level, one launch)" : "interpolation_fcycle + level visit, down (bricks, one launch)"));
    if (brick_op_is_wide(cfg))
      HIP_OK(hpgmg_hip_brick_wide_chain(n, lv, &hp_backend_of(levels[lo + n])->dev, sweeps, hp_variant(), cfg->smoother, e_id, R_id, a, b, dir,
                                        dir == 0 ? (is_first_launch ? top_e_zero : 1) : 0, (dir != 1 && is_last_launch) ? below_zero : 0));
    else
    HIP_OK(hpgmg_hip_brick_chain(n, lv, &hp_backend_of(levels[lo + n])->dev, sweeps, hp_variant(), cfg->smoother, e_id, R_id, a, b, dir, (int)hp_switch(SW_BRICK_SIZE),
                                 dir == 0 ? (is_first_launch ? top_e_zero : 1) : 0, (dir != 1 && is_last_launch) ? below_zero : 0));
    TOCK();
    done += n;
  }
}
/* leg 16 + x: would leg x be taken?  (nothing is launched) */
int hp_vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg) {
  hpgmg_config cfg;
  const hpgmg_hip_level *dev[8];
  int l, s, probe = 0;
  double h2inv[8], c1[64], c2[64];
  const int enabled = (int)hp_switch(SW_FUSED_TAIL), bottom_enabled = (int)hp_switch(SW_FUSED_BOTTOM);
  hpgmg_get_config(&cfg);
  if (leg >= HPGMG_LEG_ASK) { probe = 1; leg -= HPGMG_LEG_ASK; }
  if ((leg <= HPGMG_LEG_VCYCLE || leg == HPGMG_LEG_FCYCLE_STEP) && !probe) {
    /* launch-bound levels above the tail: one launch per level visit, then the tail, then one launch per visit on the way up.  leg 6: the step of
     * FMGSolve's climb (mg.c:1289-1293): interpolation_fcycle(levels[0] <- levels[1]) rides in the first launch of the V-cycle that follows it --
     * every brick of that launch reads levels[1]'s correction, so its zero_vector is left to the launch that visits levels[1] (a brick level too). */
    const int fstep = (leg == HPGMG_LEG_FCYCLE_STEP), vleg = fstep ? HPGMG_LEG_VCYCLE : leg;
    if (fstep && (!hp_switch(SW_BRICK_FSTEP) || brick_op_is_wide(&cfg))) return 0;      /* (27-point / fv4: interpolation_fcycle stays a launch of its own) */
    const int k = brick_prefix(levels, n, &cfg);
    const int outer_follows = tail_follows_bricks;      /* (the call for the tail comes through here again, with k == 0) */
    tail_follows_bricks = outer_follows || (k > 0);
    const int tail_ok = (k > (fstep ? 1 : 0)) && hp_vcycle_legs_fused(levels + k, n - k, e_id, R_id, a, b, HPGMG_LEG_ASK + vleg);
    tail_follows_bricks = outer_follows;
    if (tail_ok) {
      /* zero_vector of a brick level below the first: by the launch that visits it (which then does not read the vector either); the tail's first level: here */
      if (vleg != 1) {
        if (fstep) { brick_chain(levels[0], levels, 0, 1, &cfg, e_id, R_id, a, b, 2, 0, 0); brick_chain(levels[0], levels, 1, k - 1, &cfg, e_id, R_id, a, b, 0, 1, 1); }
        else brick_chain(levels[0], levels, 0, k, &cfg, e_id, R_id, a, b, 0, 0, 1);
      }
      tail_books_on = levels[0]; tail_follows_bricks = 1;
      const int taken = hp_vcycle_legs_fused(levels + k, n - k, e_id, R_id, a, b, vleg);
      tail_books_on = NULL; tail_follows_bricks = outer_follows;
      if (!taken) { fprintf(stderr, "hpgmg: the V-cycle tail was refused after being accepted\n"); abort(); }
      if (vleg != 0) brick_chain(levels[0], levels, 0, k, &cfg, e_id, R_id, a, b, 1, 0, 0);
      return 1;
    }
    if (fstep) return 0;
  }
  if (leg == HPGMG_LEG_FCYCLE_STEP) return 0;
  /* A correction or right-hand side that lives among the work vectors of the host-driven Krylov solver (ids >= VECTORS_RESERVED: MGPCG's z,
   * mg.c:1530) ALIASES them on the bottom level -- z is BiCGStab's p there (solvers/bicgstab.c:14-19) -- and the reference's numbers include that.
   * The fused bottom solve keeps the solver's vectors to itself, so the forms that contain it step aside: the legs run without it and the host-driven
   * solver goes through the operators, aliasing included. */
  if (leg >= 2 && (e_id >= hpgmg_vectors_reserved() || R_id >= hpgmg_vectors_reserved())) return 0;
  if (leg >= 2 && hpgmg_get_bottom_solver() != HPGMG_BOTTOM_BICGSTAB) return 0;        /* the device bottom solve is BiCGStab: another host solver runs through the operators */
  const int sweeps = hpgmg_smooth_sweeps();
  const int with_bottom = (leg >= 2);
  if (enabled && cfg.op != HPGMG_OP_7PT && n == 1 && tail_follows_bricks && leg <= 2) {
    /* every level above the bottom one was visited as bricks: what is left of the V-cycle is the bottom solve (leg 2) or nothing (the legs around a host-driven one) */
    if (leg != 2) return 1;
    if (!bottom_enabled || hpgmg_get_bottom_solver() != HPGMG_BOTTOM_BICGSTAB || e_id >= hpgmg_vectors_reserved() || R_id >= hpgmg_vectors_reserved()) return 0;
    return bottom_solve_fused_impl(levels[0], e_id, R_id, a, b, MG_DEFAULT_BOTTOM_NORM, probe);
  }
  if (enabled && cfg.op != HPGMG_OP_7PT) {      /* leg 0 / 1: the way down / up around a bottom solve somebody else runs (the reference's driver, through the queue below) */
    if (leg == 2) return bottom_enabled ? small_vtail_fused(levels, n, e_id, R_id, a, b, 7, probe) : 0;
    if (leg == 0 || leg == 1) return small_vtail_fused(levels, n, e_id, R_id, a, b, leg == 0 ? 1 : 4, probe);
    return 0;
  }
  if (!enabled || !hp_ghost_free_mode() || cfg.op != HPGMG_OP_7PT || n > 8 || n > hpgmg_hip_tail_max_levels() || sweeps > 8) return 0;
  if (with_bottom && !bottom_enabled) return 0;
  if (n < (leg == 3 ? 1 : 2)) return 0;
  if (leg >= 4 && (!hp_switch(SW_FUSED_FTAIL) || levels[0]->dim.i > (int)hp_switch(SW_FTAIL_MAX_DIM))) return 0;
  /* multi-rank jobs: the chain qualifies when this rank owns every box of every level in it (checked below), which is
   * how the coarse levels end up after agglomeration onto rank 0 -- no message and no all-reduce is needed then */
  for (l = 0; l < n; l++) {
    level_type *L = levels[l];
    backend_t *B = hp_backend_of(L);
    const long long cells = (long long)L->dim.i * L->dim.j * L->dim.k;
    if (!L->active || L->num_my_boxes < 1 || !B->all_faces_local) return 0;
    /* the kernel addresses cells by global coordinate: cubic Dirichlet domain, boxes in lexicographic order, halving per level */
    if (L->boundary_condition.type != BC_DIRICHLET || L->dim.i != L->dim.j || L->dim.i != L->dim.k || (sweeps & 1)) return 0;
    if (l > 0 && 2 * L->dim.i != levels[l - 1]->dim.i) return 0;
    {
      const int nb = L->dim.i / L->box_dim;
      int bx;
      if (L->num_my_boxes != nb * nb * nb) return 0;
      for (bx = 0; bx < L->num_my_boxes; bx++) {
        const box_type *X = &L->my_boxes[bx];
        if (X->low.i != (bx % nb) * L->box_dim || X->low.j != ((bx / nb) % nb) * L->box_dim || X->low.k != (bx / (nb * nb)) * L->box_dim) return 0;
      }
    }
    if (l + 1 < n) {
      if (cells > hpgmg_hip_tail_max_cells()) return 0;
      if (L->dominant_eigenvalue_of_DinvA <= 0.0 && cfg.smoother == HPGMG_SMOOTH_CHEBY) return 0;
      cheby_coefficients(L, sweeps, c1 + l * sweeps, c2 + l * sweeps);
    } else {
      for (s = 0; s < sweeps; s++) c1[l * sweeps + s] = c2[l * sweeps + s] = 0.0;
      if (with_bottom) {
        /* solvers.c:27-95: Dirichlet never subtracts the mean; the Krylov vectors must exist */
        if (cells > hpgmg_hip_tail_bottom_max_cells() || L->must_subtract_mean == 1) return 0;
        if (L->numVectors < hpgmg_vectors_reserved() + IterativeSolver_NumVectors()) return 0;
        L->must_subtract_mean = 0;
        if (!B->krylov_pinned) B->krylov_pinned = (int *)hpgmg_hip_host_malloc(64);
        if (!B->krylov_pinned) return 0;
      }
    }
    dev[l] = &B->dev;
    h2inv[l] = 1.0 / (L->h * L->h);
  }
  if (leg == HPGMG_LEG_FCYCLE_TAIL_ASK || probe) return 1;
  TICK(tail_books_on ? tail_books_on : levels[0], smooth, leg == 3 ? "bottom solve (device BiCGStab)" : (leg == 4 ? "fused F-cycle tail" : "fused V-cycle tail"));
  HIP_OK(hpgmg_hip_vcycle_tail(n, dev, h2inv, c1, c2, sweeps, hp_variant(), cfg.smoother, e_id, R_id, a, b, leg,
                               hpgmg_vectors_reserved(), MG_DEFAULT_BOTTOM_NORM, with_bottom ? hp_backend_of(levels[n - 1])->krylov_pinned : NULL));
  TOCK();
  return 1;
}

/* every box of the level is local and local box b sits at lexicographic position b (what the kernels that address
 * cells by global coordinate assume) */
static int boxes_lexicographic(level_type *L) {
  backend_t *B = hp_backend_of(L);
  if (B->lexicographic < 0) {
    int bx, ok = (L->num_my_boxes == L->boxes_in.i * L->boxes_in.j * L->boxes_in.k);
    for (bx = 0; ok && bx < L->num_my_boxes; bx++) {
      const box_type *X = &L->my_boxes[bx];
      if (X->low.i != (bx % L->boxes_in.i) * L->box_dim || X->low.j != ((bx / L->boxes_in.i) % L->boxes_in.j) * L->box_dim ||
          X->low.k != (bx / (L->boxes_in.i * L->boxes_in.j)) * L->box_dim) ok = 0;
    }
    B->lexicographic = ok;
  }
  return B->lexicographic;
}

/* BASELINE config 5: mixed-precision Chebyshev smoother.  32 = the fused sweep pairs read fp32 copies of the five
 * coefficient vectors (the iterate, the right-hand side and all arithmetic stay fp64; residual, restriction,
 * interpolation and every level the pair kernel does not cover are unchanged).  64 (default) = bit-exact fp64. */
void hpgmg_set_smoother_precision(int bits) { hp_switch_set(SW_SMOOTHER_PRECISION, (bits == 32) ? 32 : 64); }
int hpgmg_get_smoother_precision(void) {
  return hp_switch(SW_SMOOTHER_PRECISION) == 32 ? 32 : 64;
}
static const float *const *coef32_of(level_type *L) {
  backend_t *B = hp_backend_of(L);
  if (hpgmg_get_smoother_precision() != 32) return NULL;
  if (!B->coef32) {
    int bx;
    float **base = (float **)calloc((size_t)L->num_my_boxes, sizeof(float *));
    B->coef32 = (float *)hpgmg_hip_malloc(((size_t)L->num_my_boxes * 5 * (size_t)L->box_volume + 4) * sizeof(float));
    B->d_coef32_base = (float **)hpgmg_hip_malloc((size_t)L->num_my_boxes * sizeof(float *));
    if (!B->coef32 || !B->d_coef32_base) { fprintf(stderr, "hpgmg: no memory for the fp32 coefficient copies\n"); abort(); }
    /* pairs (2 floats) must be 8-byte aligned where the fp64 pairs are 16-byte aligned: same parity of the first interior cell */
    const size_t pad = ((uintptr_t)L->my_boxes[0].vectors[0] % 16) / sizeof(double);
    for (bx = 0; bx < L->num_my_boxes; bx++) base[bx] = B->coef32 + pad + (size_t)bx * 5 * (size_t)L->box_volume;
    HIP_OK(hpgmg_hip_memcpy_h2d(B->d_coef32_base, base, (size_t)L->num_my_boxes * sizeof(float *)));
    free(base);
    B->coef32_valid = 0;
  }
  if (!B->coef32_valid) { HIP_OK(hpgmg_hip_coef32_refresh(&B->dev, (float *const *)B->d_coef32_base, L->numVectors)); B->coef32_valid = 1; }
  return (const float *const *)B->d_coef32_base;
}
void hp_coef32_invalidate(level_type *L) { backend_t *B = hp_backend_of(L); B->coef32_valid = 0; if (B->halo) B->halo->coef_valid = 0; hp_images_invalidate_coefficients(B); hpgmg_hip_pair_packed_invalidate(&B->dev); }

static long long pair_remote_smooths = 0;
long long hpgmg_pair_remote_smooths(void) { return pair_remote_smooths; }   /* smooth() calls done as sweep pairs with remote faces (tests) */

/* the two plugin-private vectors per box that hold x1, x2 of the first sweep pair of a smooth() */
void hp_ensure_pair_scratch(level_type *L, backend_t *B) {
  if (!B->pair_scratch) {
    int bx;
    double **base = (double **)calloc((size_t)L->num_my_boxes, sizeof(double *));
    B->pair_scratch = (double *)hpgmg_hip_malloc(((size_t)L->num_my_boxes * 2 * (size_t)L->box_volume + 2) * sizeof(double));
    B->d_pair_base = (double **)hpgmg_hip_malloc((size_t)L->num_my_boxes * sizeof(double *));
    if (!B->pair_scratch || !B->d_pair_base) { fprintf(stderr, "hpgmg: no memory for the sweep-pair scratch vectors\n"); abort(); }
    /* the vector bases share the level's alignment class so the first interior cell is 16-byte aligned here too */
    const size_t pad = ((uintptr_t)L->my_boxes[0].vectors[0] % 16) / sizeof(double);
    for (bx = 0; bx < L->num_my_boxes; bx++) base[bx] = B->pair_scratch + pad + (size_t)bx * 2 * (size_t)L->box_volume;
    HIP_OK(hpgmg_hip_memcpy_h2d(B->d_pair_base, base, (size_t)L->num_my_boxes * sizeof(double *)));
    free(base);
  }
}
void hpgmg_set_fused_sweeps(int on) { hp_switch_set(SW_FUSED_SWEEPS, on ? 1 : 0); }
void hpgmg_set_pair_min_cells(long long cells) { hp_switch_set(SW_PAIR_MIN_CELLS, cells > 0 ? cells : 2000000); }
/* common part: does the level qualify for the sweep-pair kernel, and are its two private vectors there? */
static int pair_kernel_ready(level_type *L, int x_id, int rhs_id, int sweeps) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  backend_t *B = hp_backend_of(L);
  if (!hp_switch(SW_FUSED_SWEEPS) || sweeps != 4 || cfg.op != HPGMG_OP_7PT || !hp_ghost_free_mode() || stencil_get_shape() != STENCIL_SHAPE_STAR) return 0;
  if (L->boundary_condition.type != BC_DIRICHLET || x_id == VECTOR_TEMP || rhs_id == VECTOR_TEMP) return 0;
  if (B->all_faces_local) { if (!hpgmg_hip_smooth_cheby_pair_supported(&B->dev, hp_variant()) || !boxes_lexicographic(L)) return 0; }
  else if (!hp_pair_halo_ready(L, B)) return 0;          /* faces owned by other ranks: two-deep halo, one exchange per pair */
  { /* the pass structure pays down to the 128^3 level of config 2 (2 M cells) since the kernel library picks the launch shape per launch: 36-46 us per pair of
     * sweeps there against 2 x 25 for the tiled single sweeps (with 16 waves per workgroup whatever the level it lost: 80 us) */
    if ((long long)L->dim.i * L->dim.j * L->dim.k < hp_switch(SW_PAIR_MIN_CELLS)) return 0;
  }
  hp_ensure_pair_scratch(L, B);
  hpgmg_hip_set_ghost_free(1);
  return 1;
}


/* Chebyshev smooth() as fused sweep pairs (kernels/cheby_pair.hpp): 4 sweeps = 2 passes of 10 streams instead of
 * 4 x 9.  x1,x2 of the first pair go to two plugin-private vectors, the second pair brings x3 -> VECTOR_TEMP and
 * x4 -> x_id, i.e. exactly the state chebyshev.c:43-99 leaves.  Returns 0 when the level does not qualify. */
/* smooth() called by the cycle driver through hpgmg_smooth_in_cycle(): VECTOR_TEMP (x3 of the four sweeps) is dead after it, so the second
 * pair does not store it */
/* fold_Lc: interpolation_vcycle(L, x_id, 1.0, fold_Lc, x_id) is folded into the first pair (hp_interp_smooth_fused).  One rank: the caller has already told the
 * kernel library (hpgmg_hip_pair_fold_interpolation, consumed by the first pair launch).  Faces on other ranks: every owner adds the parents of the cells it
 * SENDS while packing the pair's halo, and the kernel adds them inside the brick only -- set here, once per part of the two-part launch. */
static long long interp_folded_remote = 0;
long long hpgmg_interp_folded_remote(void) { return interp_folded_remote; }      /* smooth() calls across rank boundaries whose interpolation was folded in (tests) */
/* the first pair of a smooth() with the fold requested first: the request is consumed per launch, and a two-part launch is two launches */
static int first_pair_folded(const hpgmg_hip_level *fold_Lc, backend_t *B, int v, int x_id, int rhs_id, double a, double b, double h2inv, const double *c1, const double *c2) {
  hpgmg_hip_pair_fold_interpolation(fold_Lc, x_id, 1.0);
  return hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, NULL, 0, x_id, 0, VECTOR_TEMP, 1, 0, 1, 1, rhs_id, a, b, h2inv, c1[0], c2[0], c1[1], c2[1]);
}
static int smooth_cheby_pairs_fold(level_type *L, int x_id, int rhs_id, double a, double b, const double *c1, const double *c2, int sweeps, int temp_dead, const hpgmg_hip_level *fold_Lc) {
  if (!pair_kernel_ready(L, x_id, rhs_id, sweeps)) return 0;
  backend_t *B = hp_backend_of(L);
  const double h2inv = 1.0 / (L->h * L->h);
  const int v = hp_variant();
  const int remote = !B->all_faces_local;
  const float *const *c32 = remote ? NULL : coef32_of(L);      /* across ranks the coefficient streams stay fp64 */
  int over = 0;
  if (remote) {
    pair_remote_smooths++;
    if (fold_Lc) { hpgmg_hip_pair_halo_fold_interpolation(fold_Lc, x_id, 1.0); interp_folded_remote++; }
    over = hp_pair_halo_begin(L, B, 1, 0, x_id, 0, VECTOR_TEMP, rhs_id);
  }
  { TICK(L, smooth, "smooth (Chebyshev sweeps 1+2)");
    if (remote && fold_Lc) PAIR_REMOTE_LAUNCH(over, 0, first_pair_folded(fold_Lc, B, v, x_id, rhs_id, a, b, h2inv, c1, c2));
    else if (remote) PAIR_REMOTE_LAUNCH(over, 0, hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, c32, 0, x_id, 0, VECTOR_TEMP, 1, 0, 1, 1, rhs_id, a, b, h2inv, c1[0], c2[0], c1[1], c2[1]));
    else HIP_OK(hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, c32, 0, x_id, 0, VECTOR_TEMP, 1, 0, 1, 1, rhs_id, a, b, h2inv, c1[0], c2[0], c1[1], c2[1]));
    TOCK(); }
  if (remote) over = hp_pair_halo_begin(L, B, 0, 1, 1, 1, 0, rhs_id);
  { TICK(L, smooth, "smooth (Chebyshev sweeps 3+4)");
    if (remote) PAIR_REMOTE_LAUNCH(over, temp_dead, hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, c32, 1, 1, 1, 0, 0, VECTOR_TEMP, 0, x_id, rhs_id, a, b, h2inv, c1[2], c2[2], c1[3], c2[3]));
    else {
      if (temp_dead) hpgmg_hip_pair_discard_x1();
      HIP_OK(hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, c32, 1, 1, 1, 0, 0, VECTOR_TEMP, 0, x_id, rhs_id, a, b, h2inv, c1[2], c2[2], c1[3], c2[3]));
    }
    TOCK(); }
  return 1;
}
static int smooth_cheby_pairs(level_type *L, int x_id, int rhs_id, double a, double b, const double *c1, const double *c2, int sweeps, int temp_dead) {
  return smooth_cheby_pairs_fold(L, x_id, rhs_id, a, b, c1, c2, sweeps, temp_dead, NULL);
}
/* in-place GSRB smooth() (gsrb.c:24-132, 4 coloured half sweeps) as two passes of two half sweeps each:
 * x_id -> private vector -> x_id; VECTOR_TEMP is not touched, as in the reference's in-place form */
static int smooth_gsrb_pairs(level_type *L, int x_id, int rhs_id, double a, double b, int sweeps) {
  if (hpgmg_gsrb_out_of_place() || !pair_kernel_ready(L, x_id, rhs_id, sweeps)) return 0;
  backend_t *B = hp_backend_of(L);
  const double h2inv = 1.0 / (L->h * L->h);
  const int v = hp_variant();
  const int remote = !B->all_faces_local;
  int over = 0;
  if (remote) { pair_remote_smooths++; over = hp_pair_halo_begin(L, B, 1, 0, x_id, 0, x_id, rhs_id); }
  { TICK(L, smooth, "smooth (GSRB half sweeps 1+2)");
    if (remote) PAIR_REMOTE_LAUNCH(over, 0, hpgmg_hip_smooth_gsrb_pair(&B->dev, v, (double *const *)B->d_pair_base, NULL, 0, x_id, 0, 1, 1, rhs_id, a, b, h2inv, 0));
    else HIP_OK(hpgmg_hip_smooth_gsrb_pair(&B->dev, v, (double *const *)B->d_pair_base, NULL, 0, x_id, 0, 1, 1, rhs_id, a, b, h2inv, 0));
    TOCK(); }
  if (remote) over = hp_pair_halo_begin(L, B, 0, 1, 1, 1, 1, rhs_id);
  { TICK(L, smooth, "smooth (GSRB half sweeps 3+4)");
    if (remote) PAIR_REMOTE_LAUNCH(over, 0, hpgmg_hip_smooth_gsrb_pair(&B->dev, v, (double *const *)B->d_pair_base, NULL, 1, 1, 0, 0, x_id, rhs_id, a, b, h2inv, 2));
    else HIP_OK(hpgmg_hip_smooth_gsrb_pair(&B->dev, v, (double *const *)B->d_pair_base, NULL, 1, 1, 0, 0, x_id, rhs_id, a, b, h2inv, 2));
    TOCK(); }
  return 1;
}

/* interpolation_vcycle(Lf, e, 1.0, Lc, e) followed by smooth(Lf, e, R) -- the up-leg of MGVCycle (mg.c:1160-1161) -- with the
 * piecewise-constant interpolation folded into the first sweep pair: the interpolated e is never written or re-read.
 * Same iterate (e = x4) as the two separate operators; VECTOR_TEMP (their x3) is left unspecified -- nothing in a cycle reads it
 * (HPGMG_TEMP_SCRATCH=0 stores it as smooth() does).  0 = not applicable. */
int hpgmg_interp_smooth_fused(level_type *Lf, int e_id, int R_id, level_type *Lc, double a, double b) { hp_lazy_flush(); return hp_interp_smooth_fused(Lf, e_id, R_id, Lc, a, b, 0); }
/* exact_state: VECTOR_TEMP is left as smooth() leaves it (the lazy queue runs behind the reference's own driver, which promises nothing about it) */
/* The same fold on the levels the sweep-pair kernel does not take (the cache-resident 128^3 ... 32^3 levels of config 2: single Chebyshev sweeps): sweep 0
 * reads x_n and sweep 1 reads x_{n-1} as stored + the coarse value above the cell (kernels/stencil_direct.hpp: InterpFold), so interpolation_vcycle is no
 * launch of its own.  The same iterates; the interpolated vector itself never exists, and VECTOR_TEMP ends as smooth() leaves it (x3). */
static long long interp_folded_single = 0;
long long hpgmg_interp_folded_single(void) { return interp_folded_single; }      /* (tests) */
static int interp_smooth_fused_single(level_type *Lf, int e_id, int R_id, level_type *Lc, double a, double b) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  const int sweeps = hpgmg_smooth_sweeps(), v = hp_variant();
  communicator_type *S = &Lc->interpolation, *Rv = &Lf->interpolation;
  if (!hp_switch(SW_FUSED_RESIDUAL) || cfg.op != HPGMG_OP_7PT || cfg.smoother != HPGMG_SMOOTH_CHEBY || !Lf->active || !Lc->active || sweeps < 2) return 0;
  if (Lf->num_my_boxes < 1 || Lc->num_my_boxes < 1 || !hp_ghost_free_mode() || Lf->boundary_condition.type != BC_DIRICHLET) return 0;
  if (S->num_sends || Rv->num_recvs || S->num_blocks[0] || Rv->num_blocks[2]) return 0;   /* all parents local (the coarse level's send side, the fine level's receive side; their other sides belong to other level pairs) */
  if (e_id == VECTOR_TEMP || R_id == VECTOR_TEMP || Lf->dominant_eigenvalue_of_DinvA <= 0.0) return 0;
  backend_t *B = hp_backend_of(Lf), *Bc = hp_backend_of(Lc);
  if (!B->all_faces_local || stencil_get_shape() != STENCIL_SHAPE_STAR) return 0;
  hpgmg_hip_set_ghost_free(1);
  if (!hpgmg_hip_smooth_cheby_fold_supported(&B->dev, v)) return 0;
  const int *map = hp_restrict_map_of(Lf, B);
  if (!map) return 0;
  double c1[16], c2[16];
  const double h2inv = 1.0 / (Lf->h * Lf->h);
  int s;
  cheby_coefficients(Lf, sweeps, c1, c2);
  B->img_active = 0;
  for (s = 0; s < sweeps; s++) {
    const int src = (s & 1) ? VECTOR_TEMP : e_id, dst = (s & 1) ? e_id : VECTOR_TEMP;
    TICK(Lf, smooth, s < 2 ? "smooth (Chebyshev sweep, interpolation folded in)" : "smooth");
    if (s < 2) hpgmg_hip_stencil_fold_interpolation(&Bc->dev, e_id, map, s + 1);
    HIP_OK(hpgmg_hip_smooth_cheby(&B->dev, v, src, dst, R_id, a, b, h2inv, c1[s], c2[s]));
    TOCK();
  }
  interp_folded_single++;
  return 1;
}
int hp_interp_smooth_fused(level_type *Lf, int e_id, int R_id, level_type *Lc, double a, double b, int exact_state) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  const int sweeps = hpgmg_smooth_sweeps();
  communicator_type *S = &Lc->interpolation, *Rv = &Lf->interpolation;
  if (cfg.op != HPGMG_OP_7PT || !Lf->active || !Lc->active) return 0;
  if (cfg.smoother != HPGMG_SMOOTH_CHEBY && !(cfg.smoother == HPGMG_SMOOTH_GSRB && !hpgmg_gsrb_out_of_place())) return 0;
  if (S->num_sends || Rv->num_recvs || S->num_blocks[0] || Rv->num_blocks[2]) return 0;   /* all parents local (the coarse level's send side, the fine level's receive side; their other sides belong to other level pairs) */
  const int remote = !hp_backend_of(Lf)->all_faces_local;      /* across ranks: Chebyshev pairs only (every owner adds the parents to the halo cells it sends) */
  int parents_in_place = (Lc->num_my_boxes == Lf->num_my_boxes), bx;      /* the kernel finds the parent of a cell of fine box b in coarse box b */
  for (bx = 0; parents_in_place && bx < Lf->num_my_boxes; bx++)
    parents_in_place = (2 * Lc->my_boxes[bx].low.i == Lf->my_boxes[bx].low.i && 2 * Lc->my_boxes[bx].low.j == Lf->my_boxes[bx].low.j && 2 * Lc->my_boxes[bx].low.k == Lf->my_boxes[bx].low.k);
  if ((remote && Lf->box_dim % 128 != 0) || Lc->box_dim * 2 != Lf->box_dim || !parents_in_place ||      /* (narrow boxes across ranks: the fold stays with the single sweeps) */ (!remote && !boxes_lexicographic(Lc)) ||
      (remote && (cfg.smoother != HPGMG_SMOOTH_CHEBY || !hp_switch(SW_PAIR_REMOTE))) ||
      !pair_kernel_ready(Lf, e_id, R_id, sweeps))
    return interp_smooth_fused_single(Lf, e_id, R_id, Lc, a, b);      /* (VECTOR_TEMP ends as smooth() leaves it: also for the queue) not a sweep-pair level: the fold of the single sweeps, if it is one of those */
  if (cfg.smoother == HPGMG_SMOOTH_CHEBY && Lf->dominant_eigenvalue_of_DinvA <= 0.0) return 0;
  if (!remote) hpgmg_hip_pair_fold_interpolation(&hp_backend_of(Lc)->dev, e_id, 1.0);
  if (cfg.smoother == HPGMG_SMOOTH_CHEBY) {
    double c1[16], c2[16];
    cheby_coefficients(Lf, sweeps, c1, c2);
    const int done = smooth_cheby_pairs_fold(Lf, e_id, R_id, a, b, c1, c2, sweeps, !exact_state && hp_switch(SW_TEMP_SCRATCH), remote ? &hp_backend_of(Lc)->dev : NULL);      /* the cycle hook: VECTOR_TEMP is dead afterwards */
    if (!done) { fprintf(stderr, "hpgmg: fused interpolation+smooth refused after being accepted\n"); abort(); }
  } else if (!smooth_gsrb_pairs(Lf, e_id, R_id, a, b, sweeps)) { fprintf(stderr, "hpgmg: fused interpolation+smooth refused after being accepted\n"); abort(); }
  return 1;
}

/* Small levels of the 27-point / fv2 / fv4 plugins: smooth(), residual() or apply_op() with their exchange_boundary + apply_BCs steps as
 * ONE single-workgroup launch (kernels/stencil.hip: small_level_kernel).  mode: 0 Chebyshev, 1 GSRB, 2 Jacobi, 3 residual, 4 apply_op.
 * Returns 0 when the level does not qualify (too large, messages needed, 7-point plugin: that one has the LDS-resident tail kernel).
 * OFF by default (HPGMG_SMALL_FUSED=1 enables; bit-identical, covered by the GPU tests): measured on MI355X it is SLOWER than the
 * launches it replaces -- fv4 GSRB `7 8` 17.1 vs 12.7 ms, 27-pt GSRB 9.9 vs 6.2 ms per F-cycle -- because a 16^3 level in 8 boxes has
 * ~160 copy / boundary list entries whose dependent load chains run 16 at a time on one CU, while separate launches spread them over
 * the chip; the launch overhead saved (~5 us each) is smaller than that serialisation. */
void hpgmg_set_small_fused(int mode) { hp_switch_set(SW_SMALL_FUSED, (mode == 1 || mode == 2) ? 2 : 0); }   /* 0 off, 1 every small level, 2 (default) one-box levels in LDS */
static int small_level_try(level_type *L, int mode, int x_id, int rhs_id, int res_id, double a, double b) {
  hpgmg_config cfg;
  const int small_fused = hp_switch(SW_SMALL_FUSED) ? 2 : 0;      /* (mode 1, every small level out of global memory, measured slower in two rounds: removed) */
  /* 0: off.  1 (experiment builds): every qualifying level, out of global memory (slower than the launches it replaces, see above).  2
   * (default): smooth() on levels of ONE box whose vectors fit the LDS -- the kernel then works on an image of the box there (round 3).  With
   * generic (FLAT) accesses to the image a smooth() was one ~60 us launch instead of twelve ~5 us ones: no gain.  With LDS-typed pointers, the
   * boundary descriptors built without scratch memory and the corner / edge extrapolations of apply_BCs_v4 spread over the lanes of a wave it
   * is 27 us (fv4, 8^3): `7 8` fv4 9.45 -> 9.2 ms, fv2 7.05 -> 6.45 ms per F-cycle.  Bit-identical, tested in all three modes. */
  hpgmg_get_config(&cfg);
  /* mode 2 takes what it shortens: a smooth() of many launches (fv4 GSRB: 12, Chebyshev: 8; a residual or apply_op is two launches of ~5 us,
   * the kernel with its copies in and out ~15 us; the 27-point GSRB smoother already runs as two one-workgroup-per-box launches) */
  const int small_27 = (int)hp_switch(SW_SMALL_27PT_GSRB);
  const int worth = (mode <= 2) && (small_27 || !(cfg.op == HPGMG_OP_27PT && cfg.smoother == HPGMG_SMOOTH_GSRB));
  const int enabled = (small_fused == 2 && worth && L->num_my_boxes == 1 && (size_t)9 * (size_t)L->box_volume * sizeof(double) <= (size_t)150 * 1024);
  if (!enabled || cfg.op == HPGMG_OP_7PT || L->num_my_boxes < 1) return 0;
  if ((long long)L->dim.i * L->dim.j * L->dim.k > hpgmg_hip_small_level_max_cells()) return 0;
  if (L->num_my_boxes != L->boxes_in.i * L->boxes_in.j * L->boxes_in.k) return 0;
  const int shape = stencil_get_shape();
  communicator_type *C = &L->exchange_ghosts[shape];
  if (C->num_sends + C->num_recvs > 0 || C->num_blocks[0] || C->num_blocks[2]) return 0;
  int bc_kind = 0, zero_first = 0, n_bc = 0;
  if (L->boundary_condition.type != BC_PERIODIC) {
    n_bc = L->boundary_condition.num_blocks[shape];
    if (cfg.op == HPGMG_OP_27PT) bc_kind = (L->box_dim < 2) ? 1 : 2;                                    /* apply_BCs_p2, boundary_fd.c:93-205 */
    else if (cfg.op == HPGMG_OP_FV2 || L->box_dim < 4) { bc_kind = (L->box_dim < 2) ? 1 : 3; zero_first = (bc_kind == 3 && L->box_ghosts > 1); }   /* apply_BCs_v2 (v4 falls back to it below 4^3) */
    else { bc_kind = 4; zero_first = (L->box_ghosts > 2); }                                            /* apply_BCs_v4 */
  }
  const int sweeps = (mode <= 2) ? hpgmg_smooth_sweeps() : 1;
  double c1[16], c2[16];
  int q;
  for (q = 0; q < 16; q++) c1[q] = c2[q] = 0.0;
  if (mode == 0) cheby_coefficients(L, sweeps, c1, c2);
  if (mode == 2) for (q = 0; q < sweeps; q++) c2[q] = 2.0 / 3.0;
  if (sweeps > 8) return 0;
  backend_t *B = hp_backend_of(L);
  const double t_h2inv = 1.0 / (L->h * L->h);
  hpgmg_tick tk = hpgmg_tick_begin(L, mode <= 2 ? &L->timers.smooth : (mode == 3 ? &L->timers.residual : &L->timers.apply_op), "small level, one launch");
  HIP_OK(hpgmg_hip_small_level_op(&B->dev, hp_variant(), mode, sweeps, x_id, rhs_id, res_id, mode == 1 ? hpgmg_gsrb_out_of_place() : 0, a, b, t_h2inv, c1, c2,
                                  hp_mirror(L, C->blocks[1], C->num_blocks[1]), C->num_blocks[1],
                                  n_bc ? hp_mirror(L, L->boundary_condition.blocks[shape], n_bc) : NULL, n_bc, bc_kind, zero_first));
  hpgmg_tick_end(tk);
  return 1;
}

/* IterativeSolver's BiCGStab on a bottom level of one small box of the 27-point / fv2 / fv4 plugins as ONE launch (kernels/stencil.hip:
 * bottom_bicgstab_kernel; the 7-point plugin's bottom solve lives in its tail kernel).  Driven from the host, an iteration is ~25 launches and
 * ~6 host round trips on a level of 8 cells.  HPGMG_FUSED_BOTTOM=0 keeps the host-driven solver. */
static int bottom_solve_fused_impl(level_type *L, int e_id, int R_id, double a, double b, double want, int ask_only);
int hpgmg_bottom_solve_fused(level_type *L, int e_id, int R_id, double a, double b, double want) { return bottom_solve_fused_impl(L, e_id, R_id, a, b, want, 0); }
static int bottom_solve_fused_impl(level_type *L, int e_id, int R_id, double a, double b, double want, int ask_only) {
  hpgmg_config cfg;
  const int on = (int)hp_switch(SW_FUSED_BOTTOM);
  hpgmg_get_config(&cfg);
  if (!on || cfg.op == HPGMG_OP_7PT || !L->active || L->num_my_boxes != 1 || L->boxes_in.i * L->boxes_in.j * L->boxes_in.k != 1) return 0;
  if (e_id >= hpgmg_vectors_reserved() || R_id >= hpgmg_vectors_reserved()) return 0;       /* they alias the host solver's work vectors (hp_vcycle_legs_fused) */
  if (L->boundary_condition.type == BC_PERIODIC || L->must_subtract_mean == 1) return 0;
  if ((long long)L->dim.i * L->dim.j * L->dim.k > hpgmg_hip_bottom_bicgstab_max_cells()) return 0;
  const int shape = stencil_get_shape();
  communicator_type *C = &L->exchange_ghosts[shape];
  if (C->num_sends + C->num_recvs > 0 || C->num_blocks[0] || C->num_blocks[1] || C->num_blocks[2]) return 0;      /* one box: nothing to exchange */
  int bc_kind, zero_first = 0;
  const int n_bc = L->boundary_condition.num_blocks[shape];
  if (cfg.op == HPGMG_OP_27PT) bc_kind = (L->box_dim < 2) ? 1 : 2;                                    /* as small_level_try / apply_BCs */
  else if (cfg.op == HPGMG_OP_FV2 || L->box_dim < 4) { bc_kind = (L->box_dim < 2) ? 1 : 3; zero_first = (bc_kind == 3 && L->box_ghosts > 1); }
  else { bc_kind = 4; zero_first = (L->box_ghosts > 2); }
  if (L->numVectors < hpgmg_vectors_reserved() + IterativeSolver_NumVectors()) return 0;      /* (the solver's work vectors must exist) */
  if (ask_only) return 1;
  hp_lazy_flush();
  backend_t *B = hp_backend_of(L);
  if (!B->krylov_pinned) { B->krylov_pinned = (int *)hpgmg_hip_host_malloc(64); if (B->krylov_pinned) *B->krylov_pinned = 0; }
  if (!B->krylov_pinned) return 0;
  /* no tick of its own: the caller (MGVCycle -> IterativeSolver) already charges the bottom solve to L->timers.Total */
  HIP_OK(hpgmg_hip_bottom_bicgstab(&B->dev, hp_variant(), e_id, R_id, hpgmg_vectors_reserved(), a, b, 1.0 / (L->h * L->h), want,
                                   n_bc ? hp_mirror(L, L->boundary_condition.blocks[shape], n_bc) : NULL, n_bc, bc_kind, zero_first, B->krylov_pinned));
  return 1;
}

/* smooth() as the cycle driver uses it (mg.c:1148,1161): same iterate, but VECTOR_TEMP is left unspecified -- the next operator of a
 * cycle overwrites or ignores it.  Always returns 1 (the hook exists so that the reference's own driver, which never calls it, keeps
 * the exact state of smooth()). */
int hpgmg_smooth_in_cycle(level_type *L, int x_id, int rhs_id, double a, double b) {
  hp_lazy_flush();
  hp_do_smooth(L, x_id, rhs_id, a, b, (int)hp_switch(SW_TEMP_SCRATCH));
  return 1;
}
/* 4th-order operator, GSRB, inside a cycle (VECTOR_TEMP is scratch afterwards): each red + black pair of half sweeps as ONE pass
 * (kernels/fv4_rb.hpp) instead of gsrb.c:24-132's two.  The passes go x -> TEMP -> private vector 0 -> x (an odd number of passes cannot
 * ping-pong between two vectors); private vector 1 lends its k ghost planes to the intermediate vector's boundary values (the pre-pass).
 * 0 = not applicable, the caller runs the half sweeps one by one. */
static long long fv4_rb_smooths = 0, rb27_smooth_passes = 0;
long long hpgmg_rb27_passes(void) { return rb27_smooth_passes; }      /* red + black passes of the 27-point GSRB smoother so far (tests) */
long long hpgmg_fv4_rb_smooths(void) { return fv4_rb_smooths; }
static void fv4_rb_bcs(level_type *L, backend_t *B, int scratch, int id) {           /* apply_BCs_v4 on the pass's input (neighbouring boxes are read where they live) */
  const int shape = stencil_get_shape();
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (!scratch) { if (!hp_exchange_and_bcs_one_launch(L, id, shape, 4, 0)) apply_BCs(L, id, shape); return; }
  int n = 0;
  const hpgmg_hip_bc_entry *e = hp_bc_entries(L, shape, &n);
  hpgmg_hip_level Ls = B->dev;
  Ls.box_base = (double *const *)B->d_pair_base;
  TICK(L, boundary_conditions, "apply_BCs_v4 (private vector)");
  HIP_OK(hpgmg_hip_exchange_and_bc(&Ls, id, NULL, 0, e, n, 4));
  TOCK();
}
/* The cells whose intermediate value the one-pass kernel must not recompute: a cell next to a tile of ANOTHER box (= on an internal box face) whose
 * stencil reaches outside the domain (= within one cell of a wall in another direction).  The coefficient ghost cells outside the domain are
 * extrapolated with box-relative normals (boundary_fv.c:573-681), so two boxes hold different values for the same place there. */
static const int *fv4_special_cells(level_type *L, backend_t *B, int *n_out) {
  if (B->n_fv4_special < 0) {
    int cap = 1024, n = 0, b, ax, side, u, v;
    int *h = (int *)malloc((size_t)cap * 4 * sizeof(int));
    const int dim = L->box_dim, N[3] = { L->dim.i, L->dim.j, L->dim.k }, nb[3] = { L->boxes_in.i, L->boxes_in.j, L->boxes_in.k };
    if (L->boundary_condition.type != BC_PERIODIC)
    for (b = 0; b < L->num_my_boxes; b++) {
      const int low[3] = { L->my_boxes[b].low.i, L->my_boxes[b].low.j, L->my_boxes[b].low.k };
      for (ax = 0; ax < 3; ax++) for (side = 0; side < 2; side++) {
        const int bpos = low[ax] / dim + (side ? 1 : -1);
        if (bpos < 0 || bpos >= nb[ax]) continue;                          /* a domain wall, not an internal face */
        const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
        for (v = 0; v < dim; v++) for (u = 0; u < dim; u++) {
          const int g1 = low[a1] + u, g2 = low[a2] + v;
          if (!(g1 == 0 || g1 == N[a1] - 1 || g2 == 0 || g2 == N[a2] - 1)) continue;
          int c[3];
          c[ax] = side ? dim - 1 : 0; c[a1] = u; c[a2] = v;
          if (n == cap) { cap *= 2; h = (int *)realloc(h, (size_t)cap * 4 * sizeof(int)); }
          h[4 * n] = b; h[4 * n + 1] = c[0]; h[4 * n + 2] = c[1]; h[4 * n + 3] = c[2]; n++;
        }
      }
    }
    B->d_fv4_special = (int *)hpgmg_hip_malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(int));
    if (!B->d_fv4_special) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
    if (n > 0) HIP_OK(hpgmg_hip_memcpy_h2d(B->d_fv4_special, h, (size_t)n * 4 * sizeof(int)));
    free(h);
    B->n_fv4_special = n;
  }
  *n_out = B->n_fv4_special;
  return B->d_fv4_special;
}
static int smooth_fv4_rb(level_type *L, int x_id, int rhs_id, double a, double b, int sweeps, int temp_dead) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  backend_t *B = hp_backend_of(L);
  const int passes = sweeps / 2, v = hp_variant();
  if (cfg.op != HPGMG_OP_FV4 || !hpgmg_gsrb_out_of_place() || !temp_dead || !hp_ghost_free_mode() || (sweeps & 1) || passes < 1) return 0;
  if (L->num_my_boxes < 1 || x_id == VECTOR_TEMP || rhs_id == VECTOR_TEMP || L->box_dim < 8) return 0;
  /* boxes on other ranks: the same passes on the table with their images (halo_images.c) -- x three cells deep once per PASS, i.e. one
   * exchange per sweep where the reference has two (gsrb.c:30-33), the cells next to the faces recomputed from the owner's inputs */
  const int images = !B->all_faces_local;
  if (images && !hp_images_ready(L, B)) return 0;
  const hpgmg_hip_level *dev = images ? &B->img->dev : &B->dev;
  if (!hpgmg_hip_smooth_gsrb_fv4_rb_supported(dev, v)) return 0;
  int n_k = 0, k_local = 1, n_all = 0, p;
  const hpgmg_hip_bc_entry *e_k = NULL;
  if (images) e_k = hp_images_bc_k(L, B, &n_k);
  else if (L->boundary_condition.type != BC_PERIODIC) {
    e_k = hp_bc_entries_k(L, &n_k, &k_local);
    (void)hp_bc_entries(L, stencil_get_shape(), &n_all);
    if (!k_local || !B->bc_sources_local[stencil_get_shape()]) return 0;
  }
  hp_ensure_pair_scratch(L, B);
  double *const *pair_base = images ? (double *const *)B->img->d_pair_base : (double *const *)B->d_pair_base;
  hpgmg_hip_set_tile_ghost_free(1);
  const double h2inv = 1.0 / (L->h * L->h);
  int n_sp = 0;
  const int *sp_cells = images ? hp_images_fv4_special(L, B, &n_sp) : fv4_special_cells(L, B, &n_sp);
  /* (scratch, id) of the iterate before pass p: x, then TEMP / x alternately; an odd count routes its second pass through private vector 0 */
  int src_s = 0, src_id = x_id;
  for (p = 0; p < passes; p++) {
    int dst_s = 0, dst_id;
    const int left = passes - p;                   /* passes still to do, this one included */
    if (left == 1) dst_id = (passes == 1) ? VECTOR_TEMP : x_id;
    else if (left == 2 && !(src_s == 0 && src_id == x_id)) { dst_s = 1; dst_id = 0; }    /* two to go and not standing on x: step aside so that the last pass can land on x */
    else dst_id = (src_s == 0 && src_id == VECTOR_TEMP) ? x_id : VECTOR_TEMP;
    if (left == 2 && src_s == 0 && src_id == x_id) dst_id = VECTOR_TEMP;
    /* images: the message and the images' boundary conditions go to the exchange stream; under them the launch stream runs the tiles that
     * read neither an image nor anything the pre-pass forms (part 1), then waits, runs the pre-pass and the other tiles (part 2) */
    int overlapped = 0;
    if (images) overlapped = hp_images_refresh_begin(L, B, src_s, src_id, 3, p == 0 ? rhs_id : -1, 4);
    else fv4_rb_bcs(L, B, src_s, src_id);
    TICK(L, smooth, "smooth (fv4 GSRB, red + black half sweeps in one pass)");
    if (overlapped) {
      hpgmg_hip_set_tile_part(1);
      HIP_OK(hpgmg_hip_smooth_gsrb_fv4_rb(dev, v, pair_base, src_s, src_id, dst_s, dst_id, 1, rhs_id, a, b, h2inv, 2 * p));
      hp_images_refresh_end();
      hpgmg_hip_set_tile_part(2);
    }
    /* the pre-pass also works on the images next to the k walls: the main kernel reads the intermediate vector's ghost planes in their columns */
    HIP_OK(hpgmg_hip_fv4_rb_prepass(images ? &B->img->dev_all : dev, v, pair_base, src_s, src_id, 1, rhs_id, a, b, h2inv, 2 * p, e_k, n_k, sp_cells, n_sp));
    HIP_OK(hpgmg_hip_smooth_gsrb_fv4_rb(dev, v, pair_base, src_s, src_id, dst_s, dst_id, 1, rhs_id, a, b, h2inv, 2 * p));
    if (overlapped) hpgmg_hip_set_tile_part(0);
    TOCK();
    src_s = dst_s; src_id = dst_id;
  }
  if (passes == 1) hp_do_scale_vector(L, x_id, 1.0, VECTOR_TEMP);      /* a single pass cannot land on its own input (never the case with the reference's counts) */
  fv4_rb_smooths++;
  return 1;
}
/* temp_dead: the caller declares VECTOR_TEMP scratch after this smooth() (inside a cycle: hpgmg_smooth_in_cycle, or the operator queue saw it
 * overwritten next) -- the in-cycle forms may run: the sweep pair without the x3 store, the red + black passes of the 27-point / fv4 GSRB smoothers */
void hp_do_smooth(level_type *L, int x_id, int rhs_id, double a, double b, int temp_dead) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  const int sweeps = hpgmg_smooth_sweeps(), v = hp_variant();
  const double h2inv = 1.0 / (L->h * L->h);
  backend_t *B = hp_backend_of(L);
  int s;
  if (cfg.op != HPGMG_OP_7PT && small_level_try(L, cfg.smoother == HPGMG_SMOOTH_CHEBY ? 0 : (cfg.smoother == HPGMG_SMOOTH_GSRB ? 1 : 2), x_id, rhs_id, x_id, a, b)) return;
  if (cfg.smoother == HPGMG_SMOOTH_CHEBY) {          /* chebyshev.c:8-100 */
    double c1[16], c2[16];
    if (L->dominant_eigenvalue_of_DinvA <= 0.0 && L->my_rank == 0) fprintf(stderr, "dominant_eigenvalue_of_DinvA <= 0.0 !\n");
    cheby_coefficients(L, sweeps, c1, c2);
    if (smooth_cheby_pairs(L, x_id, rhs_id, a, b, c1, c2, sweeps, temp_dead)) return;
    for (s = 0; s < sweeps; s++) {
      const int src = (s & 1) ? VECTOR_TEMP : x_id, dst = (s & 1) ? x_id : VECTOR_TEMP;
      STENCIL_WITH_GHOSTS(L, src, dst, smooth, hpgmg_hip_smooth_cheby(hp_stencil_dev(B), v, src, dst, rhs_id, a, b, h2inv, c1[s], c2[s]));
    }
  } else if (cfg.smoother == HPGMG_SMOOTH_GSRB) {    /* gsrb.c:24-132 */
    const int oop = hpgmg_gsrb_out_of_place();
    if (smooth_gsrb_pairs(L, x_id, rhs_id, a, b, sweeps)) return;
    /* 27-point, inside a cycle (VECTOR_TEMP is scratch afterwards): each red + black pair of half sweeps as one pass, x -> TEMP -> x.
     * The state the exported smooth() must leave in VECTOR_TEMP (the iterate before the last half sweep) never exists in this form. */
    if (cfg.op == HPGMG_OP_27PT && oop && temp_dead && hp_ghost_free_mode() && sweeps % 4 == 0 && L->num_my_boxes > 0 &&
        (B->all_faces_local || hp_images_ready(L, B)) && x_id != VECTOR_TEMP && rhs_id != VECTOR_TEMP) {
      /* boxes on other ranks: the same pass on the table with their images -- x two cells deep once per pass (one exchange per sweep instead of
       * gsrb.c:30-33's two), the intermediate vector on the cells around a box recomputed from the owner's x, right-hand side and D^{-1} */
      const int images = !B->all_faces_local;
      const hpgmg_hip_level *dev = images ? &B->img->dev : &B->dev;
      const int tiled = hpgmg_hip_smooth_gsrb27_rb_supported(dev);                          /* boxes of side 64 m: marching tiles */
      const int boxed = !tiled && hpgmg_hip_smooth_gsrb27_rb_box_supported(dev);   /* boxes of 2^3 ... 16^3: one workgroup per box */
      if (tiled || boxed) {
        for (s = 0; s < sweeps; s += 2) {
          const int src = (s & 2) ? VECTOR_TEMP : x_id, dst = (s & 2) ? x_id : VECTOR_TEMP;
          int overlapped = 0;
          if (images && tiled) overlapped = hp_images_refresh_begin(L, B, 0, src, 2, s == 0 ? rhs_id : -1, 12);     /* the tiles that read no image run under the exchange */
          else if (images) hp_images_refresh(L, B, 0, src, 2, s == 0 ? rhs_id : -1, 12);
          else if (tiled && !hp_exchange_and_bcs_one_launch(L, src, stencil_get_shape(), 12, 0)) apply_BCs(L, src, stencil_get_shape());
          TICK(L, smooth, "smooth (27-point GSRB, red + black half sweeps in one pass)");
          if (overlapped) {
            hpgmg_hip_set_tile_part(1); HIP_OK(hpgmg_hip_smooth_gsrb27_rb(dev, src, dst, rhs_id, a, b, h2inv, s));
            hp_images_refresh_end();
            hpgmg_hip_set_tile_part(2); HIP_OK(hpgmg_hip_smooth_gsrb27_rb(dev, src, dst, rhs_id, a, b, h2inv, s));
            hpgmg_hip_set_tile_part(0);
          } else
          if (tiled) HIP_OK(hpgmg_hip_smooth_gsrb27_rb(dev, src, dst, rhs_id, a, b, h2inv, s));
          else       HIP_OK(hpgmg_hip_smooth_gsrb27_rb_box(dev, src, dst, rhs_id, a, b, h2inv, s));
          TOCK();
          rb27_smooth_passes++;
        }
        return;
      }
    }
    if (smooth_fv4_rb(L, x_id, rhs_id, a, b, sweeps, temp_dead)) return;
    /* The exported smooth() (VECTOR_TEMP must be left as the separate half sweeps leave it: the iterate before the last one) -- what the
     * reference's own driver calls (Route B): all sweeps but the last as red + black passes x -> TEMP -> x, the last sweep as its two half
     * sweeps x -> TEMP -> x.  The same iterates, the same final x and VECTOR_TEMP; 2 passes + 2 half sweeps instead of 6 half sweeps. */
    int first_half_sweep = 0;
    if (cfg.op == HPGMG_OP_FV4 && oop && !temp_dead && sweeps >= 6 && !(sweeps & 1) && (((sweeps - 2) / 2) & 1) == 0 && !hp_switch(SW_FV4_NO_EXACT_RB)) {
      if (smooth_fv4_rb(L, x_id, rhs_id, a, b, sweeps - 2, 1)) first_half_sweep = sweeps - 2;      /* VECTOR_TEMP is scratch to THESE passes (the half sweeps after them rewrite it); an even number of passes: they end on x */
    }
    for (s = first_half_sweep; s < sweeps; s++) {
      const int src = (oop && (s & 1)) ? VECTOR_TEMP : x_id, dst = oop ? ((s & 1) ? x_id : VECTOR_TEMP) : x_id;
      STENCIL_WITH_GHOSTS(L, src, dst, smooth, hpgmg_hip_smooth_gsrb(hp_stencil_dev(B), v, src, dst, rhs_id, a, b, h2inv, s));
    }
  } else {                                           /* jacobi.c:8-65 */
    for (s = 0; s < sweeps; s++) {
      const int src = (s & 1) ? VECTOR_TEMP : x_id, dst = (s & 1) ? x_id : VECTOR_TEMP;
      STENCIL_WITH_GHOSTS(L, src, dst, smooth, hpgmg_hip_smooth_jacobi(hp_stencil_dev(B), v, src, dst, rhs_id, a, b, h2inv, 2.0 / 3.0));
    }
  }
}

void hp_do_residual(level_type *L, int res_id, int x_id, int rhs_id, double a, double b) {   /* residual.c:9-51 */
  if (small_level_try(L, 3, x_id, rhs_id, res_id, a, b)) return;
  STENCIL_WITH_GHOSTS(L, x_id, res_id, residual, hpgmg_hip_residual(hp_stencil_dev(hp_backend_of(L)), hp_variant(), res_id, x_id, rhs_id, a, b, 1.0 / (L->h * L->h)));
}
void hp_do_apply_op(level_type *L, int Ax_id, int x_id, double a, double b) {               /* apply_op.c:9-48 */
  if (small_level_try(L, 4, x_id, -1, Ax_id, a, b)) return;
  STENCIL_WITH_GHOSTS(L, x_id, Ax_id, apply_op, hpgmg_hip_residual(hp_stencil_dev(hp_backend_of(L)), hp_variant(), Ax_id, x_id, -1, a, b, 1.0 / (L->h * L->h)));
}

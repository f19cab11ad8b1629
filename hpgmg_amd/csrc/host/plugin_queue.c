/*
 * plugin_queue.c -- the lazy operator queue and the exported operators that pass through it.
 * Part of the operator plugin (see operators_hip.c); no arithmetic on vector data happens here.
 */
#include "plugin_internal.h"

/* ---------------------------------------------------------------- lazy void operators
 * The reference's own driver (INTEGRATION.md Route B) knows nothing of the fused hooks above: MGVCycle (mg.c:1145-1164) calls
 *     smooth, residual(TEMP), restriction(from TEMP), zero_vector          on the way down, level after level,
 *     interpolation_vcycle, smooth                                          on the way up,
 * and these calls return nothing.  So the plugin may postpone them: a call that continues one of the two patterns is only recorded; the
 * first call that does not (any other operator, anything that returns a value, a copy to the host -- every device call of this file
 * passes through HIP_OK, which drains the queue first) makes the recorded operators run, as the fused forms where those apply:
 *   - a run of whole down-leg / up-leg units over small levels: the single-launch V-cycle legs (kernels/tail.hip);
 *   - residual + restriction + zero_vector of a large level: one pass (the residual is stored too: exactly the three operators' state);
 *   - interpolation_vcycle + smooth of a large level: the interpolation folded into the first sweep pair;
 *   - residual(res) followed by norm(res): one pass (norm() asks the queue).
 * Every fused form used here leaves exactly the vectors the separate operators leave (VECTOR_TEMP included).  HPGMG_LAZY=0 turns the queue off. */
enum { LZ_SMOOTH = 1, LZ_RESIDUAL, LZ_RESTRICT, LZ_ZERO, LZ_INTERP, LZ_SCALE, LZ_ADD, LZ_MUL, LZ_APPLY };
enum { LZ_NONE = 0, LZ_DOWN, LZ_UP, LZ_RN, LZ_SR, LZ_SMALL };      /* RN: a lone residual() waiting to see whether norm() of its result follows (mg.c:1321-1323);
                                                          * SR: scale_vector(R, 1.0, F) waiting for restriction(coarse R <- R): how FMGSolve starts (mg.c:1266-1277) */
typedef struct { int op; level_type *L, *L2; int i0, i1, i2; double a, b; } lazy_op;
#define LZ_MAX 80
static lazy_op lz[LZ_MAX];
static int lz_n = 0, lz_mode = LZ_NONE, lz_busy = 0;
static long long lazy_fused_legs = 0, lazy_fused_units = 0, lazy_temp_proved_dead = 0;
long long hpgmg_lazy_temp_proved_dead(void) { return lazy_temp_proved_dead; }      /* smooth() calls run in the in-cycle form because the queue saw VECTOR_TEMP overwritten next (tests) */
long long hpgmg_lazy_fused_legs(void) { return lazy_fused_legs; }      /* single-launch legs / fused large-level units issued by the queue so far (tests) */
long long hpgmg_lazy_fused_units(void) { return lazy_fused_units; }
void hpgmg_set_lazy(int on) { hp_lazy_flush(); hp_switch_set(SW_LAZY, on ? 1 : 0); }
void hpgmg_operators_flush(void) { hp_lazy_flush(); }      /* issue every postponed operator hp_now (nothing is ever left behind: any other call does the same) */
/* HPGMG_LAZY_REPORT=1: what the queue did, on stderr when the process ends (tests/test_gpu_route_b.py reads it) */
__attribute__((destructor)) static void lazy_report(void) {
  if (hp_switch(SW_LAZY_REPORT)) fprintf(stderr, "hpgmg lazy queue: %lld single-launch legs, %lld fused large-level units, %lld smooths with VECTOR_TEMP proved dead\n", lazy_fused_legs, lazy_fused_units, lazy_temp_proved_dead);
}
static int lazy_enabled(void) {
  return hp_switch(SW_LAZY) && !lz_busy;
}
/* LZ_SMALL: BLAS-1 calls, apply_op and residual on a level of ONE box of side <= 8 wait for the dot product or norm that follows them -- what a
 * host-driven Krylov solver on the bottom level issues between two scalars it needs (the reference's solvers/bicgstab.c, "Route B"; host/solvers.c
 * with HPGMG_FUSED_BOTTOM=0) -- and go out with it as ONE launch (kernels/stencil.hip: small_ops_kernel): 6 launches per BiCGStab iteration
 * instead of ~18.  HPGMG_SMALL_OPS=0 / hpgmg_set_small_ops(0) turn it off. */
static long long small_ops_groups = 0;
void hpgmg_set_small_ops(int on) { hp_lazy_flush(); hp_switch_set(SW_SMALL_OPS, on ? 1 : 0); }
long long hpgmg_small_ops_groups(void) { return small_ops_groups; }
static int small_ops_kind(int op) { return op == LZ_ADD ? 1 : op == LZ_MUL ? 2 : op == LZ_SCALE ? 3 : op == LZ_APPLY ? 4 : op == LZ_RESIDUAL ? 5 : 0; }
static int small_ops_level_ok(level_type *L) {
  if (!hp_switch(SW_SMALL_OPS) || !L->active || L->num_my_boxes != 1 || L->boxes_in.i * L->boxes_in.j * L->boxes_in.k != 1 || L->box_dim > 8) return 0;
  if (L->boundary_condition.type != BC_DIRICHLET) return 0;
  communicator_type *C = &L->exchange_ghosts[stencil_get_shape()];
  if (C->num_sends + C->num_recvs > 0 || C->num_blocks[0] || C->num_blocks[1] || C->num_blocks[2]) return 0;      /* one box: nothing to exchange */
  { const hpgmg_transport *T = hpgmg_get_transport(); if (T && T->size > 1) { hpgmg_level_ext *X = hpgmg_level_ext_get(L); if (X->num_active_ranks > 1) return 0; } }
  return L->boundary_condition.num_blocks[stencil_get_shape()] <= 64;
}
/* A scalar the host asks for is often followed by another one with no operator in between (BiCGStab: dot(As, As) then dot(As, s); norm(r) then
 * dot(r, r0)).  The queue remembers which request followed which, lets the launch that answers the first form the second as well, and answers
 * the second from that value if it comes -- as long as nothing else was issued or queued in between.  (Forming a reduction nobody asks for
 * changes no vector.) */
typedef struct { level_type *L; int kind, a, b; } so_request;
static so_request so_last, so_pred_key[8], so_pred_val[8], so_cached;
static int so_npred = 0, so_last_fresh = 0, so_cache_valid = 0;
static double so_cache_value = 0.0;
static long long small_ops_answers = 0;
long long hpgmg_small_ops_prefetched(void) { return small_ops_answers; }      /* scalars answered without a launch (tests) */
static int so_same(const so_request *r, level_type *L, int kind, int a, int b) { return r->L == L && r->kind == kind && r->a == a && r->b == b; }
void hp_small_ops_forget(void) { so_npred = 0; so_last_fresh = 0; so_cache_valid = 0; }      /* a level is going away: the remembered requests name it */
static void so_touch(void) { so_last_fresh = 0; so_cache_valid = 0; }          /* something was issued or queued: what is remembered about the last scalar is stale */
/* issue the queue (mode LZ_SMALL, or nothing) on level L as one launch; value_kind 6 / 7: ending in dot(va, vb) / norm(va), whose value is returned;
 * p_kind: a second, predicted request formed by the same launch (its value to *p_out) */
static double small_ops_issue(level_type *L, int value_kind, int va, int vb, int p_kind, int pa, int pb, double *p_out) {
  lz_busy = 1;                                            /* from here on every device call (the first hp_backend_of() of a level uploads its tables) runs at once */
  backend_t *B = hp_backend_of(L);
  hpgmg_config cfg;
  int kinds[16], c[16], a[16], b[16], q, n = 0, bc_kind, zero_first = 0;
  double sa[16], sb[16], op_a = 0.0, op_b = 0.0, v = 0.0;
  hpgmg_get_config(&cfg);
  for (q = 0; q < lz_n; q++, n++) {
    const lazy_op *o = &lz[q];
    kinds[n] = small_ops_kind(o->op); c[n] = o->i0; a[n] = o->i1; b[n] = o->i2; sa[n] = o->a; sb[n] = o->b;
    if (o->op == LZ_APPLY || o->op == LZ_RESIDUAL) { op_a = o->a; op_b = o->b; sa[n] = sb[n] = 0.0; }
  }
  if (value_kind) { kinds[n] = value_kind; c[n] = 0; a[n] = va; b[n] = vb; sa[n] = sb[n] = 0.0; n++; }
  if (value_kind && p_kind) { kinds[n] = p_kind; c[n] = 0; a[n] = pa; b[n] = pb; sa[n] = sb[n] = 0.0; n++; }
  lz_n = 0; lz_mode = LZ_NONE;
  const int shape = stencil_get_shape(), n_bc = L->boundary_condition.num_blocks[shape];
  if (cfg.op == HPGMG_OP_7PT) bc_kind = 1;                                                              /* apply_BCs, as the operators themselves choose */
  else if (cfg.op == HPGMG_OP_27PT) bc_kind = (L->box_dim < 2) ? 1 : 2;
  else if (cfg.op == HPGMG_OP_FV2 || L->box_dim < 4) { bc_kind = (L->box_dim < 2) ? 1 : 3; zero_first = (bc_kind == 3 && L->box_ghosts > 1); }
  else { bc_kind = 4; zero_first = (L->box_ghosts > 2); }
  {
    TICK(L, blas1, "queued small-level operators, one launch");
    HIP_OK(hpgmg_hip_small_ops(&B->dev, hp_variant(), n, kinds, c, a, b, sa, sb, n_bc ? hp_mirror(L, L->boundary_condition.blocks[shape], n_bc) : NULL, n_bc, bc_kind, zero_first,
                               op_a, op_b, 1.0 / (L->h * L->h), value_kind ? &v : NULL, (value_kind && p_kind) ? p_out : NULL));
    TOCK();
  }
  lz_busy = 0;
  small_ops_groups++;
  return v;
}
static void lazy_run_one(const lazy_op *o) {
  switch (o->op) {
    case LZ_SMOOTH:   hp_do_smooth(o->L, o->i0, o->i1, o->a, o->b, 0); break;
    case LZ_RESIDUAL: hp_do_residual(o->L, o->i0, o->i1, o->i2, o->a, o->b); break;
    case LZ_RESTRICT: hp_do_restriction(o->L, o->i0, o->L2, o->i1, o->i2); break;
    case LZ_ZERO:     hp_do_zero_vector(o->L, o->i0); break;
    case LZ_INTERP:   hp_do_interpolation_vcycle(o->L, o->i0, o->a, o->L2, o->i1); break;
    case LZ_SCALE:    hp_do_scale_vector(o->L, o->i0, o->a, o->i1); break;
  }
}
void hp_lazy_flush(void) {
  if (!lz_busy) so_touch();                               /* every device call of the plugin passes here first */
  if (lz_busy || lz_n == 0) return;
  lz_busy = 1;                                            /* the operators below issue device calls themselves */
  const int n = lz_n, mode = lz_mode;
  int q = 0;
  if (mode == LZ_DOWN) {
    const int units = n / 4;
    int u = 0;
    while (u < units) {
      /* levels lz[4u].L, lz[4(u+1)].L, ... and the coarse level of the last whole unit: one launch when they are small enough */
      level_type *chain[LZ_MAX / 4 + 2];
      int m = 0, w;
      for (w = u; w < units; w++) chain[m++] = lz[4 * w].L;
      chain[m++] = lz[4 * (units - 1) + 2].L;
      const lazy_op *s0 = &lz[4 * u];
      if (m >= 2 && hp_vcycle_legs_fused(chain, m, s0->i0, s0->i1, s0->a, s0->b, 0)) { lazy_fused_legs++; u = units; break; }
      /* this unit on its own: smooth, then residual + restriction + zero_vector in one pass where the level allows it.  The unit's next
       * operator is residual(VECTOR_TEMP, ...) (mg.c:1150), which overwrites what smooth() leaves in VECTOR_TEMP before anything can read it:
       * the queue has PROVED the vector dead, so the smoother may run in its in-cycle form (hpgmg_smooth_in_cycle: the sweep pair without the
       * x3 store, the 27-point / fv4 red + black passes) although the reference's driver never says so.  HPGMG_TEMP_SCRATCH=0 keeps the exact form. */
      { const lazy_op *sm = &lz[4 * u];
        if (hp_switch(SW_TEMP_SCRATCH) && sm->i0 != VECTOR_TEMP && sm->i1 != VECTOR_TEMP) { hp_do_smooth(sm->L, sm->i0, sm->i1, sm->a, sm->b, 1); lazy_temp_proved_dead++; }
        else lazy_run_one(sm); }
      const lazy_op *r = &lz[4 * u + 1], *t = &lz[4 * u + 2], *z = &lz[4 * u + 3];
      if (hp_residual_restrict_zero_fused(t->L, t->i0, r->L, r->i0, r->i1, r->i2, r->a, r->b, z->i0)) lazy_fused_units++;
      else { lazy_run_one(r); lazy_run_one(t); lazy_run_one(z); }
      u++;
    }
    q = 4 * units;
  } else if (mode == LZ_UP) {
    const int units = n / 2;
    /* units run from the coarsest pair upwards: the longest prefix that fits the single-launch leg, then unit by unit */
    int done = 0, m;
    for (m = units; m >= 1 && !done; m--) {
      level_type *chain[LZ_MAX / 2 + 2];
      int c = 0, w;
      for (w = m - 1; w >= 0; w--) chain[c++] = lz[2 * w].L;       /* finest first */
      chain[c++] = lz[0].L2;                                        /* the level the first interpolation reads */
      const lazy_op *sm = &lz[1];
      if (hp_vcycle_legs_fused(chain, c, sm->i0, sm->i1, sm->a, sm->b, 1)) { lazy_fused_legs++; done = m; }
    }
    int u;
    for (u = done; u < units; u++) {
      const lazy_op *ip = &lz[2 * u], *sm = &lz[2 * u + 1];
      if (hp_interp_smooth_fused(ip->L, sm->i0, sm->i1, ip->L2, sm->a, sm->b, 1)) lazy_fused_units++;
      else { lazy_run_one(ip); lazy_run_one(sm); }
    }
    q = 2 * units;
  } else if (mode == LZ_SMALL) {                          /* no dot product / norm came: the queue as one launch all the same */
    (void)small_ops_issue(lz[0].L, 0, 0, 0, 0, 0, 0, NULL);   /* (clears the queue and lz_busy) */
    return;
  } else if (mode == LZ_SR && n == 2) {                   /* R = 1.0 * F, then its restriction: one pass over F (the norm the kernel also forms is not asked for) */
    if (hp_norm_scale_restrict_fused(lz[0].L, lz[0].i1, lz[0].i0, lz[1].L, NULL)) { lazy_fused_units++; q = 2; }
  }
  for (; q < n; q++) lazy_run_one(&lz[q]);                /* a unit the caller did not finish */
  lz_n = 0; lz_mode = LZ_NONE;
  lz_busy = 0;
}
/* does this call continue the pattern?  1: recorded, the caller returns; 0: the caller flushes and runs it */
static int lazy_push(int op, level_type *L, level_type *L2, int i0, int i1, int i2, double a, double b) {
  if (!lazy_enabled() || lz_n == LZ_MAX || lz_busy) return 0;      /* busy: the queue is being issued; what its operators call runs at once */
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  int ok = 0;
  if (small_ops_kind(op) && (lz_n == 0 || lz_mode == LZ_SMALL) && small_ops_level_ok(L)) {     /* any plugin */
    if (lz_n == 0) { ok = 1; lz_mode = LZ_SMALL; }
    else if (L == lz[0].L && lz_n < hpgmg_hip_small_ops_max() - 1) {
      ok = 1;
      if (op == LZ_APPLY || op == LZ_RESIDUAL) { int q; for (q = 0; q < lz_n; q++) if ((lz[q].op == LZ_APPLY || lz[q].op == LZ_RESIDUAL) && (lz[q].a != a || lz[q].b != b)) ok = 0; }   /* one (a, b) per launch */
    }
    if (ok) { lazy_op *o = &lz[lz_n++]; o->op = op; o->L = L; o->L2 = L2; o->i0 = i0; o->i1 = i1; o->i2 = i2; o->a = a; o->b = b; so_touch(); return 1; }
    return 0;
  }
  if (lz_mode == LZ_SMALL) return 0;
  if (op == LZ_ADD || op == LZ_MUL || op == LZ_APPLY) return 0;
  /* the legs of MGVCycle are recognised for every plugin (the levels of one box at their end go out as one launch: small_vtail_kernel);
   * the large-level fused forms and the residual + norm / copy + restriction pairs are the 7-point plugin's */
  if (cfg.op != HPGMG_OP_7PT && !(op == LZ_SMOOTH || op == LZ_INTERP || lz_mode == LZ_DOWN || lz_mode == LZ_UP)) return 0;
  if (lz_n == 0) {
    if (op == LZ_SMOOTH) { ok = 1; lz_mode = LZ_DOWN; }
    else if (op == LZ_INTERP && a == 1.0 && i0 == i1) { ok = 1; lz_mode = LZ_UP; }
    else if (op == LZ_RESIDUAL) { ok = 1; lz_mode = LZ_RN; }
    else if (op == LZ_SCALE && a == 1.0 && i0 != i1) { ok = 1; lz_mode = LZ_SR; }
  } else if (lz_mode == LZ_SR) {
    ok = (lz_n == 1 && op == LZ_RESTRICT && L2 == lz[0].L && i1 == lz[0].i0 && i0 == lz[0].i0 && i2 == RESTRICT_CELL && L != lz[0].L);
  } else if (lz_mode == LZ_DOWN) {
    const int pos = lz_n % 4;
    const lazy_op *s0 = &lz[lz_n - pos];                  /* this unit's smooth (pos > 0) */
    if (pos == 0) { const lazy_op *z = &lz[lz_n - 1], *f = &lz[0]; ok = (op == LZ_SMOOTH && L == z->L && i0 == f->i0 && i1 == f->i1 && a == f->a && b == f->b); }
    else if (pos == 1) ok = (op == LZ_RESIDUAL && L == s0->L && i0 == VECTOR_TEMP && i1 == s0->i0 && i2 == s0->i1 && a == s0->a && b == s0->b);
    else if (pos == 2) ok = (op == LZ_RESTRICT && L2 == s0->L && i0 == s0->i1 && i1 == VECTOR_TEMP && i2 == RESTRICT_CELL && L != s0->L);
    else ok = (op == LZ_ZERO && L == lz[lz_n - 1].L && i0 == s0->i0);
  } else if (lz_mode == LZ_UP) {
    const int pos = lz_n % 2;
    if (pos == 0) { const lazy_op *p = &lz[lz_n - 2]; ok = (op == LZ_INTERP && a == 1.0 && i0 == i1 && L2 == p->L && i0 == p->i0); }
    else { const lazy_op *ip = &lz[lz_n - 1]; ok = (op == LZ_SMOOTH && L == ip->L && i0 == ip->i0 && (lz_n == 1 || (i1 == lz[1].i1 && a == lz[1].a && b == lz[1].b))); }
  }
  if (!ok) return 0;
  lazy_op *o = &lz[lz_n++];
  o->op = op; o->L = L; o->L2 = L2; o->i0 = i0; o->i1 = i1; o->i2 = i2; o->a = a; o->b = b;
  return 1;
}
/* the five operators of include/hpgmg_operators.h (= operators.h) that take part */
void smooth(level_type *L, int x_id, int rhs_id, double a, double b) {
  if (lazy_push(LZ_SMOOTH, L, NULL, x_id, rhs_id, 0, a, b)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_SMOOTH, L, NULL, x_id, rhs_id, 0, a, b)) return;      /* it may start the next pattern */
  hp_do_smooth(L, x_id, rhs_id, a, b, 0);
}
void residual(level_type *L, int res_id, int x_id, int rhs_id, double a, double b) {
  if (lazy_push(LZ_RESIDUAL, L, NULL, res_id, x_id, rhs_id, a, b)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_RESIDUAL, L, NULL, res_id, x_id, rhs_id, a, b)) return;      /* it may start the residual + norm pattern (the convergence check after the last V-cycle) */
  hp_do_residual(L, res_id, x_id, rhs_id, a, b);
}
void restriction(level_type *Lc, int id_c, level_type *Lf, int id_f, int type) {
  if (lazy_push(LZ_RESTRICT, Lc, Lf, id_c, id_f, type, 0.0, 0.0)) return;
  hp_lazy_flush();
  hp_do_restriction(Lc, id_c, Lf, id_f, type);
}
void scale_vector(level_type *L, int c, double s, int a) {
  if (lazy_push(LZ_SCALE, L, NULL, c, a, 0, s, 0.0)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_SCALE, L, NULL, c, a, 0, s, 0.0)) return;
  hp_do_scale_vector(L, c, s, a);
}
void zero_vector(level_type *L, int id) {
  if (lazy_push(LZ_ZERO, L, NULL, id, 0, 0, 0.0, 0.0)) return;
  hp_lazy_flush();
  hp_do_zero_vector(L, id);
}
void add_vectors(level_type *L, int c, double sa, int a, double sb, int b) {
  if (lazy_push(LZ_ADD, L, NULL, c, a, b, sa, sb)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_ADD, L, NULL, c, a, b, sa, sb)) return;
  hp_do_add_vectors(L, c, sa, a, sb, b);
}
void mul_vectors(level_type *L, int c, double s, int a, int b) {
  if (lazy_push(LZ_MUL, L, NULL, c, a, b, s, 0.0)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_MUL, L, NULL, c, a, b, s, 0.0)) return;
  hp_do_mul_vectors(L, c, s, a, b);
}
void apply_op(level_type *L, int Ax_id, int x_id, double a, double b) {
  if (lazy_push(LZ_APPLY, L, NULL, Ax_id, x_id, 0, a, b)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_APPLY, L, NULL, Ax_id, x_id, 0, a, b)) return;
  hp_do_apply_op(L, Ax_id, x_id, a, b);
}
static int small_value_request(level_type *L, int kind, int a, int b, double *out);
double dot(level_type *L, int a, int b) {
  { double v; if (small_value_request(L, 6, a, b, &v)) return hp_allreduce_scalar(L, v, HPGMG_REDUCE_SUM); }
  return hp_do_dot(L, a, b);
}
void interpolation_vcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  if (lazy_push(LZ_INTERP, Lf, Lc, id_f, id_c, 0, prescale, 0.0)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_INTERP, Lf, Lc, id_f, id_c, 0, prescale, 0.0)) return;
  hp_do_interpolation_vcycle(Lf, id_f, prescale, Lc, id_c);
}
/* dot() / norm() on a level of one small box: with what is queued for that level, with the request that usually follows, or from the value a
 * previous launch formed in advance.  0: not such a level (the caller takes the ordinary path). */
static int small_value_request(level_type *L, int kind, int a, int b, double *out) {
  int q;
  if (lz_busy || !lazy_enabled() || !small_ops_level_ok(L)) return 0;
  if (lz_n > 0 && !(lz_mode == LZ_SMALL && lz[0].L == L)) return 0;                       /* something else is queued: the ordinary path flushes it */
  const int was_fresh = so_last_fresh && so_last.L == L && lz_n == 0;
  if (was_fresh) {                                         /* learn: this request follows the last one with nothing in between */
    for (q = 0; q < so_npred; q++) if (so_same(&so_pred_key[q], so_last.L, so_last.kind, so_last.a, so_last.b)) break;
    if (q == so_npred && so_npred < 8) so_npred++;
    if (q < 8) { so_pred_key[q] = so_last; so_pred_val[q].L = L; so_pred_val[q].kind = kind; so_pred_val[q].a = a; so_pred_val[q].b = b; }
  }
  if (was_fresh && so_cache_valid && so_same(&so_cached, L, kind, a, b)) {
    *out = so_cache_value; small_ops_answers++;
    so_cache_valid = 0; so_last.L = L; so_last.kind = kind; so_last.a = a; so_last.b = b; so_last_fresh = 1;
    return 1;
  }
  int p_kind = 0, pa = 0, pb = 0;
  for (q = 0; q < so_npred; q++) if (so_same(&so_pred_key[q], L, kind, a, b)) { p_kind = so_pred_val[q].kind; pa = so_pred_val[q].a; pb = so_pred_val[q].b; }
  if (p_kind && lz_n >= hpgmg_hip_small_ops_max() - 2) p_kind = 0;
  if (p_kind && (pa < 0 || pa >= L->numVectors || pb < 0 || pb >= L->numVectors)) p_kind = 0;      /* a remembered request must name vectors this level has */
  double pv = 0.0;
  *out = small_ops_issue(L, kind, a, b, p_kind, pa, pb, &pv);
  so_last.L = L; so_last.kind = kind; so_last.a = a; so_last.b = b; so_last_fresh = 1;
  so_cache_valid = p_kind != 0; so_cached.L = L; so_cached.kind = p_kind; so_cached.a = pa; so_cached.b = pb; so_cache_value = pv;
  return 1;
}
double norm(level_type *L, int a) {
  { double v; if (small_value_request(L, 7, a, 0, &v)) return hp_allreduce_scalar(L, v, HPGMG_REDUCE_MAX); }
  if (lz_mode == LZ_RN && lz_n == 1 && !lz_busy && lz[0].L == L && lz[0].i0 == a) {       /* residual(a, ...) then norm(a): one pass, the residual stored as usual */
    const lazy_op o = lz[0];
    double v = 0.0;
    lz_n = 0; lz_mode = LZ_NONE;
    if (hpgmg_residual_norm_fused(L, o.i0, o.i1, o.i2, o.a, o.b, &v)) { lazy_fused_units++; return v; }
    hp_do_residual(L, o.i0, o.i1, o.i2, o.a, o.b);
  }
  return hp_do_norm(L, a);
}

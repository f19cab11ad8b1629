/*
 * plugin_ghosts.c -- ghost zones: exchange_boundary (operators/exchange_boundary.c:12-117), the overlapped exchange, apply_BCs_* (operators/boundary_fd.c, boundary_fv.c) and their host-computed entries, the black-box operator rebuild.
 * Part of the operator plugin (see operators_hip.c); no arithmetic on vector data happens here.
 */
#include "plugin_internal.h"

void hp_transport_phase(const communicator_type *recv_side, const communicator_type *send_side, int tag) {
  const hpgmg_transport *T = hpgmg_get_transport();
  int nr = recv_side ? recv_side->num_recvs : 0, ns = send_side ? send_side->num_sends : 0;
  if (nr + ns == 0) return;
  if (!T) { fprintf(stderr, "hpgmg: level needs %d messages but no transport is set\n", nr + ns); abort(); }
  T->sendrecv(T->ctx, nr, nr ? recv_side->recv_buffers : NULL, nr ? recv_side->recv_sizes : NULL, nr ? recv_side->recv_ranks : NULL,
              ns, ns ? send_side->send_buffers : NULL, ns ? send_side->send_sizes : NULL, ns ? send_side->send_ranks : NULL, tag);
}

/* ---------------------------------------------------------------- exchange_boundary.c:12-117 */
void exchange_boundary(level_type *L, int id, int shape) {
  TICK(L, ghostZone_total, "exchange_boundary");
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  communicator_type *C = &L->exchange_ghosts[shape];
  backend_t *B = hp_backend_of(L);
  HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, hp_mirror(L, C->blocks[0], C->num_blocks[0]), C->num_blocks[0]));   /* pack */
  hp_transport_phase(C, C, (L->tag << 4) | shape);
  HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, hp_mirror(L, C->blocks[1], C->num_blocks[1]), C->num_blocks[1]));   /* box -> box */
  HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, hp_mirror(L, C->blocks[2], C->num_blocks[2]), C->num_blocks[2]));   /* unpack */
  TOCK();
}

/* What the stencil routines call instead of exchange_boundary()+apply_BCs() (chebyshev.c:45-46,
 * gsrb.c:29-34, residual.c:11-12, apply_op.c:11-12).  In ghost-free mode (default for the 7-pt STAR
 * stencil; HPGMG_GHOST_FREE=0 restores the reference's three-step form) the kernel reads local
 * neighbours and the Dirichlet condition itself, so only messages from other ranks still go
 * through the ghost zone: pack -> send/recv -> unpack, no local copies, no BC launch. */
int hp_ghost_free_mode(void) { return (int)hp_switch(SW_GHOST_FREE); }
void hpgmg_set_ghost_free(int on) { hp_switch_set(SW_GHOST_FREE, on ? 1 : 0); hpgmg_hip_set_ghost_free(on ? 1 : 0); }
/* exchange_boundary(L, id, shape) + apply_BCs_p1 / p2 / v2 / v4 (order 1 / 12 / 2 / 4) as ONE launch, when the level has no messages and every
 * boundary-condition block can read its sources from the box that owns them (then the box-to-box copies and the conditions are
 * independent of each other).  with_copies = 0: only the conditions (the caller's kernel reads neighbouring boxes itself).  Returns 0 when the
 * caller must issue the two operators. */
int hp_exchange_and_bcs_one_launch(level_type *L, int id, int shape, int order, int with_copies) {
  if (!hp_switch(SW_ONE_LAUNCH_GHOSTS) || !hp_ghost_free_mode() || L->num_my_boxes < 1 || L->boundary_condition.type == BC_PERIODIC) return 0;
  if (order == 1 && L->box_ghosts != 1) return 0;
  if (order == 12 && !(L->box_dim >= 2 && L->box_ghosts == 1)) return 0;      /* the fall-backs of apply_BCs_p2 / v2 / v4 for tiny boxes stay separate launches */
  if (order == 2 && !(L->box_dim >= 2)) return 0;
  if (order == 4 && !(L->box_dim >= 4)) return 0;
  communicator_type *C = &L->exchange_ghosts[shape];
  if (C->num_sends + C->num_recvs > 0 || C->num_blocks[0] || C->num_blocks[2]) return 0;
  backend_t *B = hp_backend_of(L);
  int n = 0;
  const hpgmg_hip_bc_entry *e = hp_bc_entries(L, shape, &n);
  if (!B->bc_sources_local[shape]) return 0;
  TICK(L, ghostZone_total, "exchange_boundary + apply_BCs (one launch)");
  HIP_OK(hpgmg_hip_exchange_and_bc(&B->dev, id, with_copies ? hp_mirror(L, C->blocks[1], C->num_blocks[1]) : NULL, with_copies ? C->num_blocks[1] : 0, e, n, order));
  TOCK();
  return 1;
}
/* Returns 1 when the operand's images are being refreshed on the exchange stream: the caller then issues the stencil launch in its two parts
 * (hpgmg_hip_set_tile_part 1, hp_images_refresh_end(), 2) -- STENCIL_WITH_GHOSTS does; 0: the ghost data is in place (in stream order), one whole launch. */
int hp_ghosts_for_stencil(level_type *L, int id, int out_id) {
  const int shape = stencil_get_shape();
  hpgmg_config c;
  hpgmg_get_config(&c);
  const int fuse = hp_ghost_free_mode() && c.op == HPGMG_OP_7PT && shape == STENCIL_SHAPE_STAR;
  hpgmg_hip_set_ghost_free(fuse);   /* the in-kernel -x(centre) rule IS apply_BCs_p1; other plugins (fv2: v2 BCs) need real ghosts */
  hpgmg_hip_set_tile_ghost_free(0);
  if (fuse) {
    communicator_type *C = &L->exchange_ghosts[shape];
    if (C->num_sends + C->num_recvs > 0) {
      TICK(L, ghostZone_total, "exchange_boundary (remote faces)");
      backend_t *B = hp_backend_of(L);
      HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, hp_mirror(L, C->blocks[0], C->num_blocks[0]), C->num_blocks[0]));
      hp_transport_phase(C, C, (L->tag << 4) | shape);
      HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, hp_mirror(L, C->blocks[2], C->num_blocks[2]), C->num_blocks[2]));
      TOCK();
    }
    return 0;
  }
  /* 27-point and fv4 on a level whose boxes are all local, about to run the LDS-tiled kernel: it reads a neighbouring box's cells
   * where they live, so only the domain-boundary ghost cells are needed (each box's own, from its own interior) */
  if (hp_ghost_free_mode() && (c.op == HPGMG_OP_27PT || c.op == HPGMG_OP_FV4) && L->num_my_boxes > 0) {
    backend_t *B = hp_backend_of(L);
    if (B->all_faces_local && hpgmg_hip_tile_kernel_applies(&B->dev, hp_variant(), id != out_id)) {
      hpgmg_hip_set_tile_ghost_free(1);
      if (!hp_exchange_and_bcs_one_launch(L, id, shape, c.op == HPGMG_OP_27PT ? 12 : 4, 0)) apply_BCs(L, id, shape);
      return 0;
    }
    /* boxes on other ranks: the same kernel on the table with their images -- one message per neighbouring rank carries the cells it reads there */
    if (!B->all_faces_local && hp_images_ready(L, B) && hpgmg_hip_tile_kernel_applies(&B->img->dev, hp_variant(), id != out_id)) {
      hpgmg_hip_set_tile_ghost_free(1);
      return hp_images_refresh_begin(L, B, 0, id, stencil_get_radius(), -1, c.op == HPGMG_OP_27PT ? 12 : 4);
    }
  }
  {
    int order = 0;
    if (c.op == HPGMG_OP_27PT) order = 12; else if (c.op == HPGMG_OP_FV2) order = 2; else if (c.op == HPGMG_OP_FV4) order = 4;
    if (order && hp_exchange_and_bcs_one_launch(L, id, shape, order, 1)) return 0;
  }
  exchange_boundary(L, id, shape);
  apply_BCs(L, id, shape);
  return 0;
}

/* Halo exchange overlapped with the stencil launch that consumes it (north_star: "ghost-zone exchange on RCCL over
 * xGMI overlapped with interior smoothing"; the reference only overlaps local copies with MPI latency,
 * exchange_boundary.c:81-90).  Ghost-free 7-point path with faces owned by other ranks:
 *     launch stream:  pack | stencil on every cell whose neighbours are local or Dirichlet  | wait | shell cells
 *     comm stream:         | wait pack, grouped ncclSend/ncclRecv, unpack into ghost zones |
 * hp_overlap_begin() returns 0 when the level does not qualify (then the caller uses hp_ghosts_for_stencil()). */
void *hp_comm_stream = NULL, *hp_ev_packed = NULL, *hp_ev_landed = NULL;      /* the exchange stream of the overlapped halo exchanges and its two events (also plugin_pair_halo.c) */
long long hp_overlap_count = 0;
long long hpgmg_overlap_count(void) { return hp_overlap_count; }   /* overlapped exchanges so far (tests) */
void hpgmg_set_overlap(int on) { hp_switch_set(SW_OVERLAP, on ? 1 : 0); }
int hp_overlap_enabled(void) { return (int)hp_switch(SW_OVERLAP); }
void hp_overlap_counted(void) { hp_overlap_count++; }
int hp_overlap_begin(level_type *L, int id) {
  const int shape = stencil_get_shape();
  const hpgmg_transport *T = hpgmg_get_transport();
  hpgmg_config c;
  if (!hp_overlap_enabled() || !T || T->size < 2) return 0;
  hpgmg_get_config(&c);
  if (!(hp_ghost_free_mode() && c.op == HPGMG_OP_7PT && shape == STENCIL_SHAPE_STAR) || L->box_dim < 8) return 0;
  communicator_type *C = &L->exchange_ghosts[shape];
  if (C->num_sends + C->num_recvs == 0 || L->num_my_boxes < 1) return 0;
  if (!hp_comm_stream) {
    hp_comm_stream = hpgmg_hip_stream_create(); hp_ev_packed = hpgmg_hip_event_create(); hp_ev_landed = hpgmg_hip_event_create();
    if (!hp_comm_stream || !hp_ev_packed || !hp_ev_landed) { fprintf(stderr, "hpgmg: cannot create the exchange stream\n"); abort(); }
  }
  const double t0 = (hpgmg_get_timer_mode() == TIMERS_DEVICE) ? 0.0 : hp_now();   /* two streams: the exchange is hidden behind the stencil launch by design, only the host modes time it */
  backend_t *B = hp_backend_of(L);
  void *launch_stream = hpgmg_hip_get_stream();
  hpgmg_hip_set_ghost_free(1);
  HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, hp_mirror(L, C->blocks[0], C->num_blocks[0]), C->num_blocks[0]));          /* pack */
  const blockCopy_type *unpack = hp_mirror(L, C->blocks[2], C->num_blocks[2]);
  HIP_OK(hpgmg_hip_event_record(hp_ev_packed));
  hpgmg_hip_set_stream(hp_comm_stream);
  HIP_OK(hpgmg_hip_stream_wait_event(hp_ev_packed));
  hp_transport_phase(C, C, (L->tag << 4) | shape);
  HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, unpack, C->num_blocks[2]));                                              /* unpack */
  HIP_OK(hpgmg_hip_event_record(hp_ev_landed));
  hpgmg_hip_set_stream(launch_stream);
  if (hpgmg_get_timer_mode() != TIMERS_DEVICE) L->timers.ghostZone_total += hp_now() - t0;
  hp_overlap_count++;
  return 1;
}
void hp_overlap_end(void) { HIP_OK(hpgmg_hip_stream_wait_event(hp_ev_landed)); }
/* run a stencil launch with its operand's ghost zones: overlapped (two launches: all but the shell, then the shell) or plain */

/* ---------------------------------------------------------------- boundary_fd.c / boundary_fv.c */
void apply_BCs_p1(level_type *L, int x_id, int shape) {
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  TICK(L, boundary_conditions, "apply_BCs_p1");
  backend_t *B = hp_backend_of(L);
  const int n = L->boundary_condition.num_blocks[shape];
  HIP_OK(hpgmg_hip_apply_bc_p1(&B->dev, x_id, hp_mirror(L, L->boundary_condition.blocks[shape], n), n));
  TOCK();
}
void hp_no_kernel(const char *what) { fprintf(stderr, "hpgmg: %s has no HIP kernel yet\n", what); abort(); }
void apply_BCs_p2(level_type *L, int x_id, int shape) {                                /* boundary_fd.c:93-205 */
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (L->box_dim < 2) { apply_BCs_p1(L, x_id, shape); return; }
  TICK(L, boundary_conditions, "apply_BCs_p2");
  backend_t *B = hp_backend_of(L);
  int n = L->boundary_condition.num_blocks[shape];
  if (L->box_ghosts == 1) { const hpgmg_hip_bc_entry *e = hp_bc_entries(L, shape, &n); HIP_OK(hpgmg_hip_apply_bc_fv(&B->dev, x_id, e, n, 12)); }
  else HIP_OK(hpgmg_hip_apply_bc_p2(&B->dev, x_id, hp_mirror(L, L->boundary_condition.blocks[shape], n), n));
  TOCK();
}
void apply_BCs_v1(level_type *L, int x_id, int shape) { apply_BCs_p1(L, x_id, shape); }   /* boundary_fv.c:6-90: same one-point formula */
/* The finite-volume conditions work on a block's DOMAIN normal (its subtype): the axes leaving the domain sit at ghost index -1 / dim and
 * step inward, the others run over the block's extent.  That geometry is fixed per block, so it is worked out here once; the kernel
 * then only loads it (faces first: they are the long entries). */
int hp_bc_entry_from_block(const level_type *L, int box, const int bpos[3], const int lo[3], const int len[3], int subtype,
                           int (*find)(void *, int), void *ctx, hpgmg_hip_bc_entry *o) {
  const int strides[3] = {1, L->box_jStride, L->box_kStride};
  const int d[3] = {subtype % 3 - 1, (subtype % 9) / 3 - 1, subtype / 9 - 1};
  int ax, nf = 0, local = 1;
  o->box = box; o->nn = 0; o->base = 0; o->len0 = o->len1 = 1; o->fs0 = o->fs1 = 0;
  o->zbase = lo[0] * strides[0] + lo[1] * strides[1] + lo[2] * strides[2]; o->zi = len[0]; o->zj = len[1]; o->zk = len[2];
  int nbr[3] = {0, 0, 0};                  /* in-face axes whose range lies in the ghost zone: the block runs along that neighbour's face */
  for (ax = 0; ax < 3; ax++) {
    if (d[ax]) { o->base += (d[ax] < 0 ? -1 : L->box_dim) * strides[ax]; o->step[o->nn++] = -d[ax] * strides[ax]; }
    else {
      if (nf == 0) { o->base += lo[ax] * strides[ax]; o->len0 = len[ax]; o->fs0 = strides[ax]; nf++; }
      else if (nf == 1) { o->base += lo[ax] * strides[ax]; o->len1 = len[ax]; o->fs1 = strides[ax]; nf++; }
      if (lo[ax] < 0) nbr[ax] = -1; else if (lo[ax] >= L->box_dim) nbr[ax] = 1;
    }
  }
  /* read the cells the condition is formed from where they live: the box that owns them (same offsets, shifted by a box length) */
  o->src_box = o->box; o->src_base = o->base;
  if (nbr[0] || nbr[1] || nbr[2]) {
    const int ni = bpos[0] + nbr[0], nj = bpos[1] + nbr[1], nk = bpos[2] + nbr[2];
    int src = -1;
    if (ni >= 0 && nj >= 0 && nk >= 0 && ni < L->boxes_in.i && nj < L->boxes_in.j && nk < L->boxes_in.k) src = find(ctx, ni + L->boxes_in.i * (nj + L->boxes_in.j * nk));
    if (src >= 0) { o->src_box = src; for (ax = 0; ax < 3; ax++) o->src_base -= nbr[ax] * L->box_dim * strides[ax]; }
    else local = 0;                        /* not in the table (another rank's): that block keeps reading the ghost zone an exchange has filled */
  }
  return local;
}
static int find_own_box(void *ctx, int gid) {
  const level_type *L = (const level_type *)ctx;
  int q;
  if (L->rank_of_box[gid] != L->my_rank) return -1;
  for (q = 0; q < L->num_my_boxes; q++) if (L->my_boxes[q].global_box_id == gid) return q;
  return -1;
}
/* host list of the entries of boundary_condition.blocks[shape] (k_only: only those whose domain normal has a k component) */
static hpgmg_hip_bc_entry *bc_entries_host(level_type *L, int shape, int k_only, int *n_out, int *all_local_out) {
  const int n = L->boundary_condition.num_blocks[shape];
  const blockCopy_type *blocks = L->boundary_condition.blocks[shape];
  hpgmg_hip_bc_entry *h = (hpgmg_hip_bc_entry *)calloc((size_t)(n > 0 ? n : 1), sizeof *h);
  int kind, q, m = 0, all_local = 1, skipped = 0;
  for (kind = 1; kind <= 3; kind++) for (q = 0; q < n; q++) {
    const blockCopy_type *e = &blocks[q];
    const int d[3] = {e->subtype % 3 - 1, (e->subtype % 9) / 3 - 1, e->subtype / 9 - 1};
    const int lo[3] = {e->read.i, e->read.j, e->read.k}, len[3] = {e->dim.i, e->dim.j, e->dim.k};
    if ((d[0] != 0) + (d[1] != 0) + (d[2] != 0) != kind) continue;
    if (k_only && !d[2]) { skipped++; continue; }
    const box_type *bx = &L->my_boxes[e->read.box];
    const int bpos[3] = { bx->low.i / L->box_dim, bx->low.j / L->box_dim, bx->low.k / L->box_dim };
    if (!hp_bc_entry_from_block(L, e->read.box, bpos, lo, len, e->subtype, find_own_box, L, &h[m++])) all_local = 0;
  }
  if (m + skipped != n) { fprintf(stderr, "hpgmg: boundary-condition block without a domain normal\n"); abort(); }
  *n_out = m; *all_local_out = all_local;
  return h;
}
static hpgmg_hip_bc_entry *bc_entries_upload(hpgmg_hip_bc_entry *h, int n) {
  hpgmg_hip_bc_entry *d = (hpgmg_hip_bc_entry *)hpgmg_hip_malloc((size_t)(n > 0 ? n : 1) * sizeof *h);
  if (!d) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
  if (n > 0) HIP_OK(hpgmg_hip_memcpy_h2d(d, h, (size_t)n * sizeof *h));
  free(h);
  return d;
}
const hpgmg_hip_bc_entry *hp_bc_entries(level_type *L, int shape, int *n_out) {
  backend_t *B = hp_backend_of(L);
  const int n = L->boundary_condition.num_blocks[shape];
  *n_out = n;
  if (n <= 0) return NULL;
  if (B->d_bc[shape] && B->n_bc[shape] == n) return B->d_bc[shape];
  if (B->d_bc[shape]) hpgmg_hip_free(B->d_bc[shape]);
  int m = 0, all_local = 1;
  hpgmg_hip_bc_entry *h = bc_entries_host(L, shape, 0, &m, &all_local);
  B->bc_sources_local[shape] = all_local;
  B->d_bc[shape] = bc_entries_upload(h, n);
  B->n_bc[shape] = n;
  return B->d_bc[shape];
}
/* the blocks of the stencil's shape whose domain normal has a k component (faces below / above the domain, i-k and j-k edges) */
const hpgmg_hip_bc_entry *hp_bc_entries_k(level_type *L, int *n_out, int *all_local_out) {
  backend_t *B = hp_backend_of(L);
  if (B->n_bc_k < 0) {
    int m = 0;
    hpgmg_hip_bc_entry *h = bc_entries_host(L, stencil_get_shape(), 1, &m, &B->bc_k_local);
    B->d_bc_k = bc_entries_upload(h, m);
    B->n_bc_k = m;
  }
  *n_out = B->n_bc_k; *all_local_out = B->bc_k_local;
  return B->d_bc_k;
}
void apply_BCs_v2(level_type *L, int x_id, int shape) {                                   /* boundary_fv.c:101-250 */
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (L->box_dim < 2) { apply_BCs_v1(L, x_id, shape); return; }
  TICK(L, boundary_conditions, "apply_BCs_v2");
  int n = L->boundary_condition.num_blocks[shape];
  { const hpgmg_hip_bc_entry *e = hp_bc_entries(L, shape, &n); HIP_OK(hpgmg_hip_apply_bc_fv(&hp_backend_of(L)->dev, x_id, e, n, 2)); }   /* clears the deeper layers first when there are any */
  TOCK();
}
void apply_BCs_v4(level_type *L, int x_id, int shape) {                                   /* boundary_fv.c:262-569 */
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (L->box_ghosts < 2) { fprintf(stderr, "called quartic BC's with only 1 ghost zone!!!\n"); abort(); }
  if (L->box_dim < 4) { apply_BCs_v2(L, x_id, shape); return; }
  TICK(L, boundary_conditions, "apply_BCs_v4");
  int n = L->boundary_condition.num_blocks[shape];
  { const hpgmg_hip_bc_entry *e = hp_bc_entries(L, shape, &n); HIP_OK(hpgmg_hip_apply_bc_fv(&hp_backend_of(L)->dev, x_id, e, n, 4)); }   /* clears the deeper layers first when there are any */
  TOCK();
}
void extrapolate_betas(level_type *L) {                                                    /* boundary_fv.c:573-681 */
  if (L->boundary_condition.type == BC_PERIODIC) return;
  TICK(L, boundary_conditions, "extrapolate_betas");
  const int n = L->boundary_condition.num_blocks[STENCIL_SHAPE_BOX];
  HIP_OK(hpgmg_hip_extrapolate_betas(&hp_backend_of(L)->dev, hp_mirror(L, L->boundary_condition.blocks[STENCIL_SHAPE_BOX], n), n));
  TOCK();
}
/* operators/rebuild.c:47-208: probe with colors^3 0/1 colourings (exchange + BCs each time), accumulate on the device */
void rebuild_operator_blackbox(level_type *L, double a, double b, int colors) {
  hp_coef32_invalidate(L);
  if (L->dim.i < colors) colors = L->dim.i;
  if (L->dim.j < colors) colors = L->dim.j;
  if (L->dim.k < colors) colors = L->dim.k;
  if (L->my_rank == 0 && hpgmg_verbose) { fprintf(stdout, "  calculating D^{-1} exactly for level h=%e using %3d colors...  ", L->h, colors * colors * colors); fflush(stdout); }
  const int x_id = VECTOR_TEMP, Aii_id = VECTOR_DINV, sum_id = (hpgmg_vectors_reserved() > VECTOR_L1INV) ? VECTOR_L1INV : VECTOR_E;
  const double h2inv = 1.0 / (L->h * L->h);
  int ic, jc, kc;
  hp_do_zero_vector(L, Aii_id);
  hp_do_zero_vector(L, sum_id);
  for (kc = 0; kc < colors; kc++) for (jc = 0; jc < colors; jc++) for (ic = 0; ic < colors; ic++) {
    color_vector(L, x_id, colors, ic, jc, kc);
    exchange_boundary(L, x_id, stencil_get_shape());
    apply_BCs(L, x_id, stencil_get_shape());
    HIP_OK(hpgmg_hip_blackbox_accumulate(&hp_backend_of(L)->dev, hp_variant(), x_id, Aii_id, sum_id, a, b, h2inv));
  }
  double lambda = -1e9;
  HIP_OK(hpgmg_hip_blackbox_finalize(&hp_backend_of(L)->dev, Aii_id, sum_id, a, b, h2inv, &lambda));
  if (L->my_rank == 0 && hpgmg_verbose) fprintf(stdout, "done\n");
  { const hpgmg_transport *T = hpgmg_get_transport();
    if (T && T->size > 1) { int r, *all = (int *)malloc((size_t)T->size * sizeof(int)); for (r = 0; r < T->size; r++) all[r] = r;
      T->allreduce(T->ctx, &lambda, 1, HPGMG_REDUCE_MAX, all, T->size); free(all); } }
  { hpgmg_config cfg; hpgmg_get_config(&cfg);
    if (cfg.smoother == HPGMG_SMOOTH_CHEBY && L->my_rank == 0 && hpgmg_verbose) { fprintf(stdout, "  estimating  lambda_max... <%1.15e\n", lambda); fflush(stdout); } }
  L->dominant_eigenvalue_of_DinvA = lambda;
}
void apply_BCs(level_type *L, int x_id, int shape) {
  hpgmg_config c;
  hpgmg_get_config(&c);
  switch (c.op) {
    case HPGMG_OP_7PT:  apply_BCs_p1(L, x_id, shape); break;
    case HPGMG_OP_27PT: apply_BCs_p2(L, x_id, shape); break;
    case HPGMG_OP_FV2:  apply_BCs_v2(L, x_id, shape); break;
    default:            apply_BCs_v4(L, x_id, shape); break;
  }
}

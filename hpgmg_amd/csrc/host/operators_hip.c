/*
 * operators_hip.c -- the product's operator plugin: implements every symbol of
 * include/hpgmg_operators.h (= reference finite-volume/source/operators.h:14-50)
 * by forwarding to the gfx950 kernels behind include/hpgmg_hip.h.
 *
 * It plays the role operators.7pt.c plays in the reference (it is the ONE
 * operators.X.c compiled into the binary), but contains no arithmetic on vector
 * data: vectors live in device memory, this file only sequences launches and
 * keeps device mirrors of the immutable block lists.  There is no CPU fallback:
 * if a kernel launch fails the process aborts with the HIP error.
 *
 * Host-side structure of each routine follows the reference routine named in
 * its comment (exchange -> BC -> kernel, pack -> send/recv -> local -> unpack).
 */
#include "plugin_internal.h"

static int timer_mode = -1;
void hpgmg_set_timer_mode(int mode) { timer_mode = (mode >= 0 && mode <= 2) ? mode : 0; }
void hpgmg_set_sync_timers(int on) { timer_mode = on ? TIMERS_SYNC : TIMERS_HOST; }
int hpgmg_get_timer_mode(void) {
  if (timer_mode < 0) {
    const char *e = getenv("HPGMG_TIMERS"), *s = getenv("HPGMG_SYNC_TIMERS");
    timer_mode = TIMERS_HOST;
    if (e && !strcmp(e, "device")) timer_mode = TIMERS_DEVICE;
    if ((e && !strcmp(e, "sync")) || (s && s[0] == '1')) timer_mode = TIMERS_SYNC;
  }
  return timer_mode;
}
static double now(void) {
  struct timespec ts;
  if (hpgmg_get_timer_mode() == TIMERS_SYNC) hpgmg_hip_sync();
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
hpgmg_tick hpgmg_tick_begin(level_type *L, double *acc, const char *what) {
  hpgmg_tick t;
  t.acc = acc; t.slot = -1; t.t0 = 0.0; t.range = 0;
  if (hpgmg_hip_range_enabled()) {
    char label[96];
    snprintf(label, sizeof label, "%d^3 %s", L ? L->dim.i : 0, what);
    hpgmg_hip_range_push(label);
    t.range = 1;
  }
  if (hpgmg_get_timer_mode() == TIMERS_DEVICE) { t.slot = hpgmg_hip_timer_begin(acc); if (t.slot < 0) t.acc = NULL; }
  else t.t0 = now();
  return t;
}
void hpgmg_tick_end(hpgmg_tick t) {
  if (hpgmg_get_timer_mode() == TIMERS_DEVICE) hpgmg_hip_timer_end(t.slot);
  else if (t.acc) *t.acc += now() - t.t0;
  if (t.range) hpgmg_hip_range_pop();
}
void hpgmg_timers_settle(void) { hpgmg_hip_timer_flush(); }
static void do_smooth(level_type *L, int x_id, int rhs_id, double a, double b, int temp_dead);
static void do_residual(level_type *L, int res_id, int x_id, int rhs_id, double a, double b);
static void do_restriction(level_type *Lc, int id_c, level_type *Lf, int id_f, int type);
static void do_interpolation_vcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c);
static void do_zero_vector(level_type *L, int id);

/* ---------------------------------------------------------------- storage hooks */
const char *hpgmg_backend_name(void) { return "hip"; }
double *hpgmg_vector_alloc(size_t n) {
  double *p = (double *)hpgmg_hip_malloc(n * sizeof(double));
  if (!p) { fprintf(stderr, "hpgmg: device allocation of %zu doubles failed: %s\n", n, hpgmg_hip_last_error()); abort(); }
  return p;
}
void hpgmg_vector_free(double *p) { hp_lazy_flush(); hpgmg_hip_free(p); }      /* a postponed operator may still hold this storage */
void hpgmg_vector_copy(double *d, const double *s, size_t n) { HIP_OK(hpgmg_hip_memcpy_d2d(d, s, n * sizeof(double))); }
void hpgmg_vector_upload(double *d, const double *s, size_t n) { HIP_OK(hpgmg_hip_memcpy_h2d(d, s, n * sizeof(double))); }
void hpgmg_vector_download(double *d, const double *s, size_t n) { HIP_OK(hpgmg_hip_memcpy_d2h(d, s, n * sizeof(double))); }

/* ---------------------------------------------------------------- transport selection
 * One process per GPU: rank 0 obtains an id with hpgmg_hip_rccl_unique_id(), the launcher
 * (bench.py via torch.distributed, or any bootstrap) hands it to every rank, and each rank
 * calls this once before creating levels.  Replaces MPI_Init/MPI_Comm_rank of hpgmg-fv.c:129-136. */
int hpgmg_transport_init_rccl(const char *id128, int rank, int size) {
  hpgmg_transport t;
  int e = hpgmg_hip_rccl_init(id128, rank, size);
  if (e) return e;
  t.rank = rank; t.size = size; t.ctx = NULL;
  t.sendrecv = hpgmg_hip_rccl_sendrecv;
  t.allreduce = hpgmg_hip_rccl_allreduce;
  hpgmg_set_transport(&t);
  return 0;
}
void hpgmg_transport_finalize_rccl(void) { hpgmg_set_transport(NULL); hpgmg_hip_rccl_finalize(); }
/* The node-local alternative: direct peer copies between the ranks' device buffers (kernels/comm_ipc.hip). */
int hpgmg_transport_init_ipc(const char *name, int rank, int size) {
  hpgmg_transport t;
  int e = hpgmg_hip_ipc_init(name, rank, size);
  if (e) return e;
  t.rank = rank; t.size = size; t.ctx = NULL;
  t.sendrecv = hpgmg_hip_ipc_sendrecv;
  t.allreduce = hpgmg_hip_ipc_allreduce;
  hpgmg_set_transport(&t);
  return 0;
}
void hpgmg_transport_finalize_ipc(void) { hpgmg_set_transport(NULL); hpgmg_hip_ipc_finalize(); }

/* ---------------------------------------------------------------- hipGraph segments (see hpgmg_operators.h) */
/* hipGraph capture/replay of the launch-bound segments is available but OFF by default: with the launch stream
 * kept full by asynchronous eager launches the GPU is already 99 % busy, and on ROCm 7 replaying the segments as graphs
 * measured 2-5 % slower (3.88 vs 3.81 ms per 256^3 F-cycle, 0.574 vs 0.548 ms at 64^3).  HPGMG_GRAPH=1 or
 * hpgmg_set_graphs(1) turns it on (useful when the host thread is the bottleneck). */
void hpgmg_set_graphs(int on) { hp_switch_set(SW_GRAPH, on ? 1 : 0); }
void hpgmg_segment_begin(long long key) {
  hp_lazy_flush();                                                     /* nothing postponed may slip into (or past) the captured stretch */
  if (!hp_switch(SW_GRAPH) || hpgmg_get_timer_mode() == TIMERS_SYNC) return;       /* per-operator synchronisation: stay eager */
  /* multi-rank: segments cover levels of <= 64^3 cells; they are message-free (capturable) only when the rank map
   * gathers those levels on rank 0 (mg.c: hpgmg_gather_dim, the default) */
  { extern int hpgmg_gather_dim; const hpgmg_transport *T = hpgmg_get_transport(); if (T && T->size > 1 && hpgmg_gather_dim < 64) return; }
  if (hpgmg_hip_graph_begin(key) < 0) { fprintf(stderr, "hpgmg: graph segment failed: %s\n", hpgmg_hip_last_error()); abort(); }
}
void hpgmg_segment_end(void) { HIP_OK(hpgmg_hip_graph_end()); }


static void coef32_invalidate(level_type *L);
backend_t *hp_backend_of(level_type *L) {
  hpgmg_level_ext *X = hpgmg_level_ext_get(L);
  backend_t *B = (backend_t *)X->backend;
  if (!B) { B = (backend_t *)calloc(1, sizeof(*B)); X->backend = B; B->lexicographic = -1; B->n_bc_k = -1; B->n_fv4_special = -1; }
  double *v0 = L->num_my_boxes ? L->my_boxes[0].vectors[0] : NULL;
  if (B->seen_v0 != v0 || B->seen_nv != L->numVectors || B->seen_boxes != L->num_my_boxes || !B->d_box_low) {
    int b, n = L->num_my_boxes > 0 ? L->num_my_boxes : 1;
    double **base = (double **)calloc((size_t)n, sizeof(double *));
    int *low = (int *)calloc((size_t)n * 3, sizeof(int));
    for (b = 0; b < L->num_my_boxes; b++) {
      base[b] = L->my_boxes[b].vectors[0];
      low[3 * b] = L->my_boxes[b].low.i; low[3 * b + 1] = L->my_boxes[b].low.j; low[3 * b + 2] = L->my_boxes[b].low.k;
    }
    /* face-neighbour table for the ghost-free stencil: local box index, -1 Dirichlet face, -2 remote box */
    int *nbr = (int *)calloc((size_t)n * 6, sizeof(int));
    B->all_faces_local = 1;
    for (b = 0; b < L->num_my_boxes; b++) {
      static const int step[6][3] = { {-1,0,0}, {1,0,0}, {0,-1,0}, {0,1,0}, {0,0,-1}, {0,0,1} };
      const int bi = L->my_boxes[b].low.i / L->box_dim, bj = L->my_boxes[b].low.j / L->box_dim, bk = L->my_boxes[b].low.k / L->box_dim;
      int d;
      for (d = 0; d < 6; d++) {
        int ni = bi + step[d][0], nj = bj + step[d][1], nk = bk + step[d][2], code;
        if (L->boundary_condition.type == BC_PERIODIC) {
          ni = (ni + L->boxes_in.i) % L->boxes_in.i; nj = (nj + L->boxes_in.j) % L->boxes_in.j; nk = (nk + L->boxes_in.k) % L->boxes_in.k;
        }
        if (ni < 0 || nj < 0 || nk < 0 || ni >= L->boxes_in.i || nj >= L->boxes_in.j || nk >= L->boxes_in.k) code = -1;
        else {
          const int id = ni + L->boxes_in.i * (nj + L->boxes_in.j * nk);
          code = -2;
          if (L->rank_of_box[id] == L->my_rank) { int q; for (q = 0; q < L->num_my_boxes; q++) if (L->my_boxes[q].global_box_id == id) code = q; }
          if (code == -2) B->all_faces_local = 0;
        }
        nbr[6 * b + d] = code;
      }
    }
    if (B->d_box_base) hpgmg_hip_free(B->d_box_base);
    if (B->d_box_low) hpgmg_hip_free(B->d_box_low);
    if (B->d_box_nbr) hpgmg_hip_free(B->d_box_nbr);
    B->d_box_base = (double **)hpgmg_hip_malloc((size_t)n * sizeof(double *));
    B->d_box_low = (int *)hpgmg_hip_malloc((size_t)n * 3 * sizeof(int));
    B->d_box_nbr = (int *)hpgmg_hip_malloc((size_t)n * 6 * sizeof(int));
    if (!B->d_box_base || !B->d_box_low || !B->d_box_nbr) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
    HIP_OK(hpgmg_hip_memcpy_h2d(B->d_box_base, base, (size_t)n * sizeof(double *)));
    HIP_OK(hpgmg_hip_memcpy_h2d(B->d_box_low, low, (size_t)n * 3 * sizeof(int)));
    HIP_OK(hpgmg_hip_memcpy_h2d(B->d_box_nbr, nbr, (size_t)n * 6 * sizeof(int)));
    free(base); free(low); free(nbr);
    B->seen_v0 = v0; B->seen_nv = L->numVectors; B->seen_boxes = L->num_my_boxes;
  }
  B->dev.box_base = (double *const *)B->d_box_base;
  B->dev.box_low = B->d_box_low;
  B->dev.num_boxes = L->num_my_boxes;
  B->dev.dim = L->box_dim;       B->dev.ghosts = L->box_ghosts;
  B->dev.jStride = L->box_jStride; B->dev.kStride = L->box_kStride; B->dev.volume = L->box_volume;
  B->dev.dim_i = L->dim.i; B->dev.dim_j = L->dim.j; B->dev.dim_k = L->dim.k;
  B->dev.periodic = (L->boundary_condition.type == BC_PERIODIC);
  B->dev.box_nbr = B->d_box_nbr;
  { /* 16-byte alignment of every (box, vector) interior origin: base aligned and all strides even */
    int b, ok = (L->box_jStride % 2 == 0) && (L->box_kStride % 2 == 0) && (L->box_volume % 2 == 0);
    const size_t first = (size_t)L->box_ghosts * (size_t)(1 + L->box_jStride + L->box_kStride);
    for (b = 0; ok && b < L->num_my_boxes; b++) if (((uintptr_t)(L->my_boxes[b].vectors[0] + first)) % 16) ok = 0;
    B->dev.flags = ok ? 1 : 0;
    B->dev.box_stride = 0;
    if (L->num_my_boxes > 1) {
      const long long d = (long long)(L->my_boxes[1].vectors[0] - L->my_boxes[0].vectors[0]);
      for (b = 1; b < L->num_my_boxes && (long long)(L->my_boxes[b].vectors[0] - L->my_boxes[b - 1].vectors[0]) == d; b++) ;
      if (b == L->num_my_boxes && d > 0) B->dev.box_stride = d;
    } }
  return B;
}

/* device mirror of one immutable host list (uploaded on first use) */
const blockCopy_type *hp_mirror(level_type *owner, const blockCopy_type *host, int n) {
  int s;
  if (n <= 0 || !host) return NULL;
  backend_t *B = hp_backend_of(owner);
  for (s = 0; s < B->num_lists; s++) if (B->lists[s].host == host && B->lists[s].n == n) return B->lists[s].dev;
  if (B->num_lists == MAX_LISTS) { fprintf(stderr, "hpgmg: too many block lists on one level\n"); abort(); }
  blockCopy_type *d = (blockCopy_type *)hpgmg_hip_malloc((size_t)n * sizeof(blockCopy_type));
  if (!d) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
  HIP_OK(hpgmg_hip_memcpy_h2d(d, host, (size_t)n * sizeof(blockCopy_type)));
  B->lists[B->num_lists].host = host; B->lists[B->num_lists].n = n; B->lists[B->num_lists].dev = d;
  B->num_lists++;
  return d;
}

static void small_ops_forget(void);
void hpgmg_level_release(level_type *L) {
  hp_lazy_flush();                                  /* postponed operators hold a pointer to their level */
  small_ops_forget();                            /* ... and so do the remembered scalar requests */
  hpgmg_hip_graph_reset();                       /* cached graphs hold pointers into this level */
  hpgmg_hip_timer_forget(&L->timers, &L->timers + 1);   /* pending device timers point into this level */
  hpgmg_level_ext *X = hpgmg_level_ext_get(L);
  backend_t *B = (backend_t *)X->backend;
  int s;
  if (!B) return;
  hpgmg_hip_pair_packed_forget(&B->dev);
  hp_images_release(B);
  for (s = 0; s < B->num_lists; s++) hpgmg_hip_free(B->lists[s].dev);
  for (s = 0; s < STENCIL_MAX_SHAPES; s++) if (B->d_bc[s]) hpgmg_hip_free(B->d_bc[s]);
  if (B->d_bc_k) hpgmg_hip_free(B->d_bc_k);
  if (B->d_fv4_special) hpgmg_hip_free(B->d_fv4_special);
  if (B->d_box_base) hpgmg_hip_free(B->d_box_base);
  if (B->d_box_low) hpgmg_hip_free(B->d_box_low);
  if (B->d_box_nbr) hpgmg_hip_free(B->d_box_nbr);
  if (B->krylov_pinned) hpgmg_hip_host_free(B->krylov_pinned);
  if (B->pair_scratch) hpgmg_hip_free(B->pair_scratch);
  if (B->coef32) hpgmg_hip_free(B->coef32);
  if (B->d_coef32_base) hpgmg_hip_free(B->d_coef32_base);
  if (B->d_pair_base) hpgmg_hip_free(B->d_pair_base);
  if (B->d_restrict_map) hpgmg_hip_free(B->d_restrict_map);
  if (B->halo) {
    int q;
    for (q = 0; q < HALO_PLANS; q++) {
      halo_plan *P = &B->halo->plan[q];
      if (P->d_send) hpgmg_hip_free(P->d_send);
      if (P->d_recv) hpgmg_hip_free(P->d_recv);
      free(P->sp_rank); free(P->rp_rank); free(P->sp_size); free(P->rp_size); free(P->sp_ptr); free(P->rp_ptr);
    }
    if (B->halo->sendbuf) hpgmg_hip_free(B->halo->sendbuf);
    if (B->halo->recvbuf) hpgmg_hip_free(B->halo->recvbuf);
    if (B->halo->deep) hpgmg_hip_free(B->halo->deep);
    if (B->halo->deep_beta) hpgmg_hip_free(B->halo->deep_beta);
    free(B->halo);
  }
  free(B);
  X->backend = NULL;
}

int hp_variant(void) {
  hpgmg_config c;
  hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_7PT || c.op == HPGMG_OP_FV2)   /* operators.fv2.c: the 7-pt stencil with finite-volume BCs/interpolation */
    return !c.variable_coeff ? HPGMG_HIP_7PT_CC : (c.helmholtz ? HPGMG_HIP_7PT_VC_HELMHOLTZ : HPGMG_HIP_7PT_VC_POISSON);
  if (c.op == HPGMG_OP_27PT) return HPGMG_HIP_27PT_CC;
  if (c.op == HPGMG_OP_FV4 && c.variable_coeff) return c.helmholtz ? HPGMG_HIP_FV4_VC_HELMHOLTZ : HPGMG_HIP_FV4_VC_POISSON;
  fprintf(stderr, "hpgmg: operator %d has no HIP kernels yet\n", c.op);
  abort();
}

static void transport_phase(const communicator_type *recv_side, const communicator_type *send_side, int tag) {
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
  transport_phase(C, C, (L->tag << 4) | shape);
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
/* exchange_boundary(L, id, shape) + apply_BCs_p2 / v2 / v4 (order 12 / 2 / 4) as ONE launch, when the level has no messages and every
 * boundary-condition block can read its sources from the box that owns them (then the box-to-box copies and the conditions are
 * independent of each other).  with_copies = 0: only the conditions (the caller's kernel reads neighbouring boxes itself).  Returns 0 when the
 * caller must issue the two operators. */
static const hpgmg_hip_bc_entry *bc_entries(level_type *L, int shape, int *n_out);
static int exchange_and_bcs_one_launch(level_type *L, int id, int shape, int order, int with_copies) {
  if (!hp_switch(SW_ONE_LAUNCH_GHOSTS) || !hp_ghost_free_mode() || L->num_my_boxes < 1 || L->boundary_condition.type == BC_PERIODIC) return 0;
  if (order == 12 && !(L->box_dim >= 2 && L->box_ghosts == 1)) return 0;      /* the fall-backs of apply_BCs_p2 / v2 / v4 for tiny boxes stay separate launches */
  if (order == 2 && !(L->box_dim >= 2)) return 0;
  if (order == 4 && !(L->box_dim >= 4)) return 0;
  communicator_type *C = &L->exchange_ghosts[shape];
  if (C->num_sends + C->num_recvs > 0 || C->num_blocks[0] || C->num_blocks[2]) return 0;
  backend_t *B = hp_backend_of(L);
  int n = 0;
  const hpgmg_hip_bc_entry *e = bc_entries(L, shape, &n);
  if (!B->bc_sources_local[shape]) return 0;
  TICK(L, ghostZone_total, "exchange_boundary + apply_BCs (one launch)");
  HIP_OK(hpgmg_hip_exchange_and_bc(&B->dev, id, with_copies ? hp_mirror(L, C->blocks[1], C->num_blocks[1]) : NULL, with_copies ? C->num_blocks[1] : 0, e, n, order));
  TOCK();
  return 1;
}
static void ghosts_for_stencil(level_type *L, int id, int out_id) {
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
      transport_phase(C, C, (L->tag << 4) | shape);
      HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, hp_mirror(L, C->blocks[2], C->num_blocks[2]), C->num_blocks[2]));
      TOCK();
    }
    return;
  }
  /* 27-point and fv4 on a level whose boxes are all local, about to run the LDS-tiled kernel: it reads a neighbouring box's cells
   * where they live, so only the domain-boundary ghost cells are needed (each box's own, from its own interior) */
  if (hp_ghost_free_mode() && (c.op == HPGMG_OP_27PT || c.op == HPGMG_OP_FV4) && L->num_my_boxes > 0) {
    backend_t *B = hp_backend_of(L);
    if (B->all_faces_local && hpgmg_hip_tile_kernel_applies(&B->dev, hp_variant(), id != out_id)) {
      hpgmg_hip_set_tile_ghost_free(1);
      if (!exchange_and_bcs_one_launch(L, id, shape, c.op == HPGMG_OP_27PT ? 12 : 4, 0)) apply_BCs(L, id, shape);
      return;
    }
    /* boxes on other ranks: the same kernel on the table with their images -- one message per neighbouring rank carries the cells it reads there */
    if (!B->all_faces_local && hp_images_ready(L, B) && hpgmg_hip_tile_kernel_applies(&B->img->dev, hp_variant(), id != out_id)) {
      hpgmg_hip_set_tile_ghost_free(1);
      hp_images_refresh(L, B, 0, id, stencil_get_radius(), -1, c.op == HPGMG_OP_27PT ? 12 : 4);
      return;
    }
  }
  {
    int order = 0;
    if (c.op == HPGMG_OP_27PT) order = 12; else if (c.op == HPGMG_OP_FV2) order = 2; else if (c.op == HPGMG_OP_FV4) order = 4;
    if (order && exchange_and_bcs_one_launch(L, id, shape, order, 1)) return;
  }
  exchange_boundary(L, id, shape);
  apply_BCs(L, id, shape);
}

/* Halo exchange overlapped with the stencil launch that consumes it (north_star: "ghost-zone exchange on RCCL over
 * xGMI overlapped with interior smoothing"; the reference only overlaps local copies with MPI latency,
 * exchange_boundary.c:81-90).  Ghost-free 7-point path with faces owned by other ranks:
 *     launch stream:  pack | stencil on every cell whose neighbours are local or Dirichlet  | wait | shell cells
 *     comm stream:         | wait pack, grouped ncclSend/ncclRecv, unpack into ghost zones |
 * overlap_begin() returns 0 when the level does not qualify (then the caller uses ghosts_for_stencil()). */
static void *comm_stream = NULL, *ev_packed = NULL, *ev_landed = NULL;
static long long overlap_count = 0;
long long hpgmg_overlap_count(void) { return overlap_count; }   /* overlapped exchanges so far (tests) */
void hpgmg_set_overlap(int on) { hp_switch_set(SW_OVERLAP, on ? 1 : 0); }
int hp_overlap_enabled(void) { return (int)hp_switch(SW_OVERLAP); }
void hp_overlap_counted(void) { overlap_count++; }
static int overlap_begin(level_type *L, int id) {
  const int shape = stencil_get_shape();
  const hpgmg_transport *T = hpgmg_get_transport();
  hpgmg_config c;
  if (!hp_overlap_enabled() || !T || T->size < 2) return 0;
  hpgmg_get_config(&c);
  if (!(hp_ghost_free_mode() && c.op == HPGMG_OP_7PT && shape == STENCIL_SHAPE_STAR) || L->box_dim < 8) return 0;
  communicator_type *C = &L->exchange_ghosts[shape];
  if (C->num_sends + C->num_recvs == 0 || L->num_my_boxes < 1) return 0;
  if (!comm_stream) {
    comm_stream = hpgmg_hip_stream_create(); ev_packed = hpgmg_hip_event_create(); ev_landed = hpgmg_hip_event_create();
    if (!comm_stream || !ev_packed || !ev_landed) { fprintf(stderr, "hpgmg: cannot create the exchange stream\n"); abort(); }
  }
  const double t0 = (hpgmg_get_timer_mode() == TIMERS_DEVICE) ? 0.0 : now();   /* two streams: the exchange is hidden behind the stencil launch by design, only the host modes time it */
  backend_t *B = hp_backend_of(L);
  void *launch_stream = hpgmg_hip_get_stream();
  hpgmg_hip_set_ghost_free(1);
  HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, hp_mirror(L, C->blocks[0], C->num_blocks[0]), C->num_blocks[0]));          /* pack */
  const blockCopy_type *unpack = hp_mirror(L, C->blocks[2], C->num_blocks[2]);
  HIP_OK(hpgmg_hip_event_record(ev_packed));
  hpgmg_hip_set_stream(comm_stream);
  HIP_OK(hpgmg_hip_stream_wait_event(ev_packed));
  transport_phase(C, C, (L->tag << 4) | shape);
  HIP_OK(hpgmg_hip_copy_blocks(&B->dev, id, unpack, C->num_blocks[2]));                                              /* unpack */
  HIP_OK(hpgmg_hip_event_record(ev_landed));
  hpgmg_hip_set_stream(launch_stream);
  if (hpgmg_get_timer_mode() != TIMERS_DEVICE) L->timers.ghostZone_total += now() - t0;
  overlap_count++;
  return 1;
}
static void overlap_end(void) { HIP_OK(hpgmg_hip_stream_wait_event(ev_landed)); }
/* run a stencil launch with its operand's ghost zones: overlapped (two launches: all but the shell, then the shell) or plain */
#define STENCIL_WITH_GHOSTS(L, id, out_id, TIMER, CALL) do {                                                     \
    hp_backend_of(L)->img_active = 0;                                                                    \
    if (overlap_begin(L, id)) {                                                                          \
      TICK(L, TIMER, #TIMER " (overlapped with the halo exchange)");                                     \
      hpgmg_hip_set_defer_mode(1); HIP_OK(CALL);                                                         \
      overlap_end();                                                                                     \
      hpgmg_hip_set_defer_mode(2); HIP_OK(CALL); hpgmg_hip_set_defer_mode(0);                            \
      TOCK();                                                                                            \
    } else {                                                                                             \
      ghosts_for_stencil(L, id, out_id);                                                                 \
      TICK(L, TIMER, #TIMER);                                                                            \
      HIP_OK(CALL);                                                                                      \
      TOCK();                                                                                            \
    } } while (0)

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
static void no_kernel(const char *what) { fprintf(stderr, "hpgmg: %s has no HIP kernel yet\n", what); abort(); }
static const hpgmg_hip_bc_entry *bc_entries(level_type *L, int shape, int *n_out);
void apply_BCs_p2(level_type *L, int x_id, int shape) {                                /* boundary_fd.c:93-205 */
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (L->box_dim < 2) { apply_BCs_p1(L, x_id, shape); return; }
  TICK(L, boundary_conditions, "apply_BCs_p2");
  backend_t *B = hp_backend_of(L);
  int n = L->boundary_condition.num_blocks[shape];
  if (L->box_ghosts == 1) { const hpgmg_hip_bc_entry *e = bc_entries(L, shape, &n); HIP_OK(hpgmg_hip_apply_bc_fv(&B->dev, x_id, e, n, 12)); }
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
static const hpgmg_hip_bc_entry *bc_entries(level_type *L, int shape, int *n_out) {
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
static const hpgmg_hip_bc_entry *bc_entries_k(level_type *L, int *n_out, int *all_local_out) {
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
  { const hpgmg_hip_bc_entry *e = bc_entries(L, shape, &n); HIP_OK(hpgmg_hip_apply_bc_fv(&hp_backend_of(L)->dev, x_id, e, n, 2)); }   /* clears the deeper layers first when there are any */
  TOCK();
}
void apply_BCs_v4(level_type *L, int x_id, int shape) {                                   /* boundary_fv.c:262-569 */
  if (shape >= STENCIL_MAX_SHAPES) shape = STENCIL_SHAPE_BOX;
  if (L->boundary_condition.type == BC_PERIODIC) return;
  if (L->box_ghosts < 2) { fprintf(stderr, "called quartic BC's with only 1 ghost zone!!!\n"); abort(); }
  if (L->box_dim < 4) { apply_BCs_v2(L, x_id, shape); return; }
  TICK(L, boundary_conditions, "apply_BCs_v4");
  int n = L->boundary_condition.num_blocks[shape];
  { const hpgmg_hip_bc_entry *e = bc_entries(L, shape, &n); HIP_OK(hpgmg_hip_apply_bc_fv(&hp_backend_of(L)->dev, x_id, e, n, 4)); }   /* clears the deeper layers first when there are any */
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
  coef32_invalidate(L);
  if (L->dim.i < colors) colors = L->dim.i;
  if (L->dim.j < colors) colors = L->dim.j;
  if (L->dim.k < colors) colors = L->dim.k;
  if (L->my_rank == 0 && hpgmg_verbose) { fprintf(stdout, "  calculating D^{-1} exactly for level h=%e using %3d colors...  ", L->h, colors * colors * colors); fflush(stdout); }
  const int x_id = VECTOR_TEMP, Aii_id = VECTOR_DINV, sum_id = (hpgmg_vectors_reserved() > VECTOR_L1INV) ? VECTOR_L1INV : VECTOR_E;
  const double h2inv = 1.0 / (L->h * L->h);
  int ic, jc, kc;
  do_zero_vector(L, Aii_id);
  do_zero_vector(L, sum_id);
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

/* leg 0/1: the legs around a host-driven bottom solve; leg 2: legs + bottom solve; leg 3: bottom solve only (n == 1);
 * leg 4: the whole F-cycle below levels[0] (right-hand side restricted down the chain, bottom solve, interpolation_fcycle + V-cycle per
 * level upwards); leg 5: only answer whether leg 4 would be accepted */
static int vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg);
int hpgmg_vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg) { hp_lazy_flush(); return vcycle_legs_fused(levels, n, e_id, R_id, a, b, leg); }
/* The same for the 27-point / fv2 / fv4 plugins, leg 2 only (smooth ... bottom solve ... smooth as one launch): every level of the chain is ONE
 * box whose vectors fit the LDS (kernels/stencil.hip: small_vtail_kernel).  `7 8`: the levels of 8^3, 4^3, 2^3 (and 1^3) cells. */
/* On by default except for the 27-point plugin with GSRB (HPGMG_SMALL_VTAIL=0 / 1, hpgmg_set_small_vtail(); bit-identical, tested both ways).
 * Measured on MI355X, `7 8` F-cycles with it on / off: fv4 GSRB 7.72 / 7.77 ms, fv4 Chebyshev 8.13 / 8.23, fv2 GSRB 6.35 / 6.56, fv2 Chebyshev
 * 6.65 / 6.91, 27-point Chebyshev 4.72 / 4.84 -- and 27-point GSRB 4.04 / 3.94: that plugin's one-launch red + black box kernel beats two half
 * sweeps of the generic form.  The first version (1024 lanes) was slower everywhere: the bottom solve's 240 registers per lane spilled into
 * scratch memory under the 128-register cap; with 512 lanes the launch of 8^3 + 4^3 + 2^3 levels takes 169 instead of 191 us (fv4 GSRB;
 * tools/exp_vtail_timeline.py): 4 x 24 us of smoothing, 28 us of bottom solve, the rest image traffic and interpolation. */
static long long small_vtails = 0;
long long hpgmg_small_vtails(void) { return small_vtails; }
void hpgmg_set_small_vtail(int on) { hp_switch_set(SW_SMALL_VTAIL, (on == 2) ? 2 : (on ? 1 : 0)); }      /* 0 off, 1 on for every plugin, 2 the default (not for 27-point GSRB) */
static int small_vtail_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int legs) {
  hpgmg_config cfg;
  hpgmg_hip_small_tail_args T;
  int l;
  const int small_vtail_on = (int)hp_switch(SW_SMALL_VTAIL);      /* 2: the default */
  if (!small_vtail_on) return 0;
  hpgmg_get_config(&cfg);
  if (small_vtail_on == 2 && cfg.op == HPGMG_OP_27PT && cfg.smoother == HPGMG_SMOOTH_GSRB) return 0;
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
  TICK(levels[0], smooth, "fused V-cycle tail (levels of one box)");
  HIP_OK(hpgmg_hip_small_vtail(&T, hp_variant()));
  TOCK();
  small_vtails++;
  return 1;
}
void hpgmg_set_fused_tail(int on) { hp_switch_set(SW_FUSED_TAIL, on ? 1 : 0); }      /* tests: 0 = every operator of the small levels as its own launch(es) */
void hpgmg_set_fused_bottom(int on) { hp_switch_set(SW_FUSED_BOTTOM, on ? 1 : 0); }  /* tests: 0 = the bottom solve driven from the host (host/solvers.c BiCGStab through the operators) */
static int vcycle_legs_fused(level_type **levels, int n, int e_id, int R_id, double a, double b, int leg) {
  hpgmg_config cfg;
  const hpgmg_hip_level *dev[8];
  int l, s;
  double h2inv[8], c1[64], c2[64];
  const int enabled = (int)hp_switch(SW_FUSED_TAIL), bottom_enabled = (int)hp_switch(SW_FUSED_BOTTOM);
  hpgmg_get_config(&cfg);
  const int sweeps = hpgmg_smooth_sweeps();
  const int with_bottom = (leg >= 2);
  if (enabled && cfg.op != HPGMG_OP_7PT) {      /* leg 0 / 1: the way down / up around a bottom solve somebody else runs (the reference's driver, through the queue below) */
    if (leg == 2) return bottom_enabled ? small_vtail_fused(levels, n, e_id, R_id, a, b, 7) : 0;
    if (leg == 0 || leg == 1) return small_vtail_fused(levels, n, e_id, R_id, a, b, leg == 0 ? 1 : 4);
    return 0;
  }
  if (!enabled || !hp_ghost_free_mode() || cfg.op != HPGMG_OP_7PT || n > 8 || n > hpgmg_hip_tail_max_levels() || sweeps > 8) return 0;
  if (with_bottom && !bottom_enabled) return 0;
  if (n < (leg == 3 ? 1 : 2)) return 0;
  if (leg >= 4 && !hp_switch(SW_FUSED_FTAIL)) return 0;
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
  if (leg == 5) return 1;
  TICK(levels[0], smooth, leg == 3 ? "bottom solve (device BiCGStab)" : (leg == 4 ? "fused F-cycle tail" : "fused V-cycle tail"));
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
static void coef32_invalidate(level_type *L) { backend_t *B = hp_backend_of(L); B->coef32_valid = 0; if (B->halo) B->halo->coef_valid = 0; hp_images_invalidate_coefficients(B); hpgmg_hip_pair_packed_invalidate(&B->dev); }

/* ---------------------------------------------------------------- sweep pairs across rank boundaries: halo plans */
static int pair_remote_enabled(void) { return (int)hp_switch(SW_PAIR_REMOTE); }
int hp_box_rank_at(const level_type *L, int bi, int bj, int bk) {           /* -1 outside the (non-periodic) domain */
  if (bi < 0 || bj < 0 || bk < 0 || bi >= L->boxes_in.i || bj >= L->boxes_in.j || bk >= L->boxes_in.k) return -1;
  return L->rank_of_box[bi + L->boxes_in.i * (bj + L->boxes_in.j * bk)];
}
/* Every rank's boxes form a brick (a box-aligned sub-block of the domain)?  Decided from the global box -> rank table, so all
 * ranks reach the same answer (they must: the message pattern of a smooth() depends on it).  Returns my brick in lo/n. */
static int every_rank_owns_a_brick(const level_type *L, int lo[3], int n[3]) {
  const hpgmg_transport *T = hpgmg_get_transport();
  const int nr = T ? T->size : 1;
  int *mn = (int *)malloc((size_t)nr * 3 * sizeof(int)), *mx = (int *)malloc((size_t)nr * 3 * sizeof(int)), *cnt = (int *)calloc((size_t)nr, sizeof(int));
  int r, bi, bj, bk, ok = 1;
  for (r = 0; r < 3 * nr; r++) { mn[r] = 1 << 30; mx[r] = -1; }
  for (bk = 0; bk < L->boxes_in.k; bk++) for (bj = 0; bj < L->boxes_in.j; bj++) for (bi = 0; bi < L->boxes_in.i; bi++) {
    const int c[3] = { bi, bj, bk };
    int a;
    r = hp_box_rank_at(L, bi, bj, bk);
    if (r < 0 || r >= nr) { ok = 0; continue; }
    cnt[r]++;
    for (a = 0; a < 3; a++) { if (c[a] < mn[3 * r + a]) mn[3 * r + a] = c[a]; if (c[a] > mx[3 * r + a]) mx[3 * r + a] = c[a]; }
  }
  for (r = 0; r < nr && ok; r++)
    if (cnt[r] > 0 && cnt[r] != (mx[3 * r] - mn[3 * r] + 1) * (mx[3 * r + 1] - mn[3 * r + 1] + 1) * (mx[3 * r + 2] - mn[3 * r + 2] + 1)) ok = 0;
  if (ok && L->my_rank < nr && cnt[L->my_rank] > 0) { int a; for (a = 0; a < 3; a++) { lo[a] = mn[3 * L->my_rank + a]; n[a] = mx[3 * L->my_rank + a] - mn[3 * L->my_rank + a] + 1; } }
  else ok = 0;
  free(mn); free(mx); free(cnt);
  return ok;
}

typedef struct { int send_id, sdir, item; hpgmg_hip_halo_entry e; int peer; } halo_rec;
static int halo_rec_cmp(const void *pa, const void *pb) {
  const halo_rec *a = (const halo_rec *)pa, *b = (const halo_rec *)pb;
  if (a->peer != b->peer) return a->peer < b->peer ? -1 : 1;
  if (a->send_id != b->send_id) return a->send_id < b->send_id ? -1 : 1;
  if (a->sdir != b->sdir) return a->sdir < b->sdir ? -1 : 1;
  return (a->item > b->item) - (a->item < b->item);
}
static int local_box_of(const level_type *L, int gid) { int b; for (b = 0; b < L->num_my_boxes; b++) if (L->my_boxes[b].global_box_id == gid) return b; return -1; }

/* turn sorted records into a plan: per-peer message sizes / offsets and the device region lists */
static size_t halo_finish_side(halo_rec *rec, int n, hpgmg_hip_halo_entry **d_list, int *n_msg, int **ranks, int **sizes, long long **offs) {
  int q, m = 0;
  size_t total = 0;
  qsort(rec, (size_t)n, sizeof(halo_rec), halo_rec_cmp);
  *ranks = (int *)malloc((size_t)(n + 1) * sizeof(int)); *sizes = (int *)calloc((size_t)(n + 1), sizeof(int)); *offs = (long long *)calloc((size_t)(n + 1), sizeof(long long));
  hpgmg_hip_halo_entry *host = (hpgmg_hip_halo_entry *)malloc((size_t)(n + 1) * sizeof(*host));
  for (q = 0; q < n; q++) {
    if (m == 0 || (*ranks)[m - 1] != rec[q].peer) { (*ranks)[m] = rec[q].peer; (*offs)[m] = (long long)total; m++; }
    rec[q].e.off = (long long)total;
    const int len = rec[q].e.ni * rec[q].e.nj * rec[q].e.nk;
    (*sizes)[m - 1] += len; total += (size_t)len;
    host[q] = rec[q].e;
  }
  *n_msg = m;
  *d_list = NULL;
  if (n > 0) {
    *d_list = (hpgmg_hip_halo_entry *)hpgmg_hip_malloc((size_t)n * sizeof(*host));
    if (!*d_list) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
    HIP_OK(hpgmg_hip_memcpy_h2d(*d_list, host, (size_t)n * sizeof(*host)));
  }
  free(host);
  return total;
}

static pair_halo *pair_halo_build(level_type *L, const int lo[3], const int n[3]) {
  const int me = L->my_rank, dim = L->box_dim;
  pair_halo *H = (pair_halo *)calloc(1, sizeof(*H));
  int a, which, bi, bj, bk, dir;
  for (a = 0; a < 3; a++) H->brick[a] = n[a];
  { /* a brick face is the domain boundary or belongs to another rank (never to me: the brick is my whole share) */
    const int bl[3] = { L->boxes_in.i, L->boxes_in.j, L->boxes_in.k };
    for (a = 0; a < 3; a++) { H->rem[2 * a] = (lo[a] > 0); H->rem[2 * a + 1] = (lo[a] + n[a] < bl[a]); }
  }
  const int max_rec = L->boxes_in.i * L->boxes_in.j * L->boxes_in.k * 18 * 4 + 4;
  size_t need_send = 0, need_recv = 0;
  long long **soffs = NULL; (void)soffs;
  for (which = 0; which < HALO_PLANS; which++) {
    halo_plan *P = &H->plan[which];
    halo_rec *snd = (halo_rec *)malloc((size_t)max_rec * sizeof(halo_rec)), *rcv = (halo_rec *)malloc((size_t)max_rec * sizeof(halo_rec));
    int ns = 0, nrv = 0;
    /* every (receiving box, direction) of the level; both sides evaluate the same rule from the global box -> rank table */
    for (bk = 0; bk < L->boxes_in.k; bk++) for (bj = 0; bj < L->boxes_in.j; bj++) for (bi = 0; bi < L->boxes_in.i; bi++) for (dir = 0; dir < 27; dir++) {
      const int d[3] = { dir % 3 - 1, (dir / 3) % 3 - 1, dir / 9 - 1 };
      const int order = (d[0] != 0) + (d[1] != 0) + (d[2] != 0);
      if (order != 1 && order != 2) continue;
      if (which == HALO_COEF && order != 1) continue;
      const int M = hp_box_rank_at(L, bi, bj, bk), S = hp_box_rank_at(L, bi + d[0], bj + d[1], bk + d[2]);
      if (M < 0 || S < 0 || M == S || (M != me && S != me)) continue;
      if (order == 2) {   /* an edge value is read only where BOTH faces it touches belong to other ranks (a Dirichlet face overrides it) */
        int need = 1;
        for (a = 0; a < 3; a++) if (d[a]) { const int r = hp_box_rank_at(L, bi + (a == 0 ? d[0] : 0), bj + (a == 1 ? d[1] : 0), bk + (a == 2 ? d[2] : 0)); if (r < 0 || r == M) need = 0; }
        if (!need) continue;
      }
      const int recv_id = bi + L->boxes_in.i * (bj + L->boxes_in.j * bk);
      const int send_id = (bi + d[0]) + L->boxes_in.i * ((bj + d[1]) + L->boxes_in.j * (bk + d[2]));
      const int sdir = 26 - dir;
      int face = -1;
      if (order == 1) face = d[0] ? (d[0] < 0 ? 0 : 1) : (d[1] ? (d[1] < 0 ? 2 : 3) : (d[2] < 0 ? 4 : 5));
      /* items: (vector, depth).  depth 1 / 2: the region one / two cells beyond the face; depth 10 + t (coefficients only): the line of
       * HIGH-face values of beta_t (index dim along the tangential axis t) on the ghost layer -- the ghost cells' own upper faces, which
       * the BOX exchange of rebuild_operator only delivers where a diagonal neighbour box exists, i.e. not along the domain boundary */
      int items[4][2], nitems = 0;
      if (which == HALO_COEF) {
        const int beta_of[3] = { VECTOR_BETA_I, VECTOR_BETA_J, VECTOR_BETA_K };
        if (d[0] + d[1] + d[2] > 0) { items[nitems][0] = 16 + (d[0] ? VECTOR_BETA_I : (d[1] ? VECTOR_BETA_J : VECTOR_BETA_K)); items[nitems++][1] = 2; }
        for (a = 0; a < 3; a++) if (!d[a]) { items[nitems][0] = 16 + beta_of[a]; items[nitems++][1] = 10 + a; }
      } else {
        items[nitems][0] = 0; items[nitems++][1] = 1;
        if (order == 1) {
          items[nitems][0] = 0; items[nitems++][1] = 2;
          items[nitems][0] = 1; items[nitems++][1] = 1;
          if (which == HALO_FIRST) { items[nitems][0] = 2; items[nitems++][1] = 1; }
        }
      }
      int it;
      for (it = 0; it < nitems; it++) {
        const int depth = items[it][1];
        halo_rec R;
        memset(&R, 0, sizeof(R));
        R.send_id = send_id; R.sdir = sdir; R.item = it;
        R.e.vec = items[it][0]; R.e.deep = -1;
        int lo3[3], len3[3];
        if (M == me) {                                  /* what I receive: the ghost region (depth 1) or a deep plane (depth 2) */
          for (a = 0; a < 3; a++) { lo3[a] = d[a] < 0 ? -1 : (d[a] > 0 ? dim : 0); len3[a] = d[a] ? 1 : dim; }
          if (depth >= 10) { lo3[depth - 10] = dim; len3[depth - 10] = 1; }
          R.e.box = local_box_of(L, recv_id);
          if (depth == 2) R.e.deep = (which == HALO_COEF) ? 8 + face / 2 : face;
          R.e.i = lo3[0]; R.e.j = lo3[1]; R.e.k = lo3[2]; R.e.ni = len3[0]; R.e.nj = len3[1]; R.e.nk = len3[2];
          R.peer = S;
          rcv[nrv++] = R;
        }
        if (S == me) {                                  /* what I send: my cells next to (depth 1) / one further from (depth 2) that face */
          for (a = 0; a < 3; a++) {
            /* seen from the sender the receiver lies in direction -d: d > 0 means the sender is on the receiver's high side and sends its LOW cells */
            if (depth == 2 && which == HALO_COEF) lo3[a] = d[a] ? 1 : 0;        /* beta face index 1 of the sender = index dim + 1 of the receiver */
            else { const int dd = depth >= 10 ? 1 : depth; lo3[a] = d[a] > 0 ? (dd - 1) : (d[a] < 0 ? dim - dd : 0); }
            len3[a] = d[a] ? 1 : dim;
          }
          if (depth >= 10) { lo3[depth - 10] = dim; len3[depth - 10] = 1; }
          R.e.box = local_box_of(L, send_id);
          R.e.deep = -1;
          R.e.i = lo3[0]; R.e.j = lo3[1]; R.e.k = lo3[2]; R.e.ni = len3[0]; R.e.nj = len3[1]; R.e.nk = len3[2];
          R.peer = M;
          snd[ns++] = R;
        }
      }
    }
    long long *so = NULL, *ro = NULL;
    const size_t ts = halo_finish_side(snd, ns, &P->d_send, &P->n_sp, &P->sp_rank, &P->sp_size, &so);
    const size_t tr = halo_finish_side(rcv, nrv, &P->d_recv, &P->n_rp, &P->rp_rank, &P->rp_size, &ro);
    P->n_send = ns; P->n_recv = nrv;
    P->sp_ptr = (double **)calloc((size_t)(P->n_sp + 1), sizeof(double *)); P->rp_ptr = (double **)calloc((size_t)(P->n_rp + 1), sizeof(double *));
    { int q; for (q = 0; q < P->n_sp; q++) P->sp_ptr[q] = (double *)(uintptr_t)so[q]; for (q = 0; q < P->n_rp; q++) P->rp_ptr[q] = (double *)(uintptr_t)ro[q]; }   /* offsets for now */
    free(so); free(ro); free(snd); free(rcv);
    if (ts > need_send) need_send = ts;
    if (tr > need_recv) need_recv = tr;
  }
  H->sendbuf = hpgmg_vector_alloc(need_send + 2);
  H->recvbuf = hpgmg_vector_alloc(need_recv + 2);
  H->deep = hpgmg_vector_alloc((size_t)L->num_my_boxes * 6 * (size_t)dim * dim + 2);
  H->deep_beta = hpgmg_vector_alloc((size_t)L->num_my_boxes * 3 * (size_t)dim * dim + 2);
  for (which = 0; which < HALO_PLANS; which++) {
    halo_plan *P = &H->plan[which];
    int q;
    for (q = 0; q < P->n_sp; q++) P->sp_ptr[q] = H->sendbuf + (size_t)(uintptr_t)P->sp_ptr[q];
    for (q = 0; q < P->n_rp; q++) P->rp_ptr[q] = H->recvbuf + (size_t)(uintptr_t)P->rp_ptr[q];
  }
  return H;
}

/* may smooth() on this level run as sweep pairs although some faces belong to other ranks?  (same answer on every rank) */
static int pair_halo_ready(level_type *L, backend_t *B) {
  if (B->halo_state == 0) {
    const hpgmg_transport *T = hpgmg_get_transport();
    int lo[3], n[3], b, ok;
    B->halo_state = -1;
    ok = pair_remote_enabled() && T && T->size > 1 && L->boundary_condition.type == BC_DIRICHLET && L->box_dim % 128 == 0 && L->num_my_boxes > 0;
    if (ok) ok = every_rank_owns_a_brick(L, lo, n);
    for (b = 0; ok && b < L->num_my_boxes; b++) {     /* local numbering = lexicographic inside the brick */
      const box_type *X = &L->my_boxes[b];
      const int ci = X->low.i / L->box_dim - lo[0], cj = X->low.j / L->box_dim - lo[1], ck = X->low.k / L->box_dim - lo[2];
      if (ci + n[0] * (cj + n[1] * ck) != b) ok = 0;
    }
    if (ok && !hpgmg_hip_smooth_cheby_pair_supported_brick(&B->dev, hp_variant(), n[0], n[1], n[2])) ok = 0;
    if (ok) { B->halo = pair_halo_build(L, lo, n); B->halo_state = 1; }
  }
  return B->halo_state > 0;
}
static void pair_halo_exchange(level_type *L, backend_t *B, int which, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id) {
  const hpgmg_transport *T = hpgmg_get_transport();
  pair_halo *H = B->halo;
  halo_plan *P = &H->plan[which];
  if (P->n_send + P->n_recv == 0) return;
  TICK(L, ghostZone_total, which == HALO_COEF ? "coefficient halo (sweep pairs)" : "sweep-pair halo exchange");
  HIP_OK(hpgmg_hip_pair_halo_pack(&B->dev, (double *const *)B->d_pair_base, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id, P->d_send, P->n_send, H->sendbuf));
  T->sendrecv(T->ctx, P->n_rp, P->rp_ptr, P->rp_size, P->rp_rank, P->n_sp, P->sp_ptr, P->sp_size, P->sp_rank, (L->tag << 4) | 0x8 | which);
  HIP_OK(hpgmg_hip_pair_halo_unpack(&B->dev, (double *const *)B->d_pair_base, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id, P->d_recv, P->n_recv, H->recvbuf, H->deep, H->deep_beta));
  TOCK();
}
/* The same with the message hidden behind computation: the exchange goes to the exchange stream, the launch stream runs the workgroups of the pair
 * launch that touch no face of another rank (part 1: hpgmg_hip_set_tile_part), waits, and runs the others (part 2).  Returns 1 when set up that way
 * -- the caller issues part 1, overlap_end(), part 2, each after hpgmg_hip_pair_set_halo() (consumed per launch) -- and 0 when the exchange was done
 * in line (HPGMG_OVERLAP=0): one whole launch. */
static int pair_halo_begin(level_type *L, backend_t *B, int first, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id) {
  pair_halo *H = B->halo;
  if (!H->coef_valid) { pair_halo_exchange(L, B, HALO_COEF, 0, 0, 0, 0, 0); H->coef_valid = 1; }
  if (!hp_overlap_enabled() || hpgmg_get_timer_mode() == TIMERS_SYNC) {
    pair_halo_exchange(L, B, first ? HALO_FIRST : HALO_NEXT, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id);
    return 0;
  }
  if (!comm_stream) {
    comm_stream = hpgmg_hip_stream_create(); ev_packed = hpgmg_hip_event_create(); ev_landed = hpgmg_hip_event_create();
    if (!comm_stream || !ev_packed || !ev_landed) { fprintf(stderr, "hpgmg: cannot create the exchange stream\n"); abort(); }
  }
  void *launch_stream = hpgmg_hip_get_stream();
  HIP_OK(hpgmg_hip_event_record(ev_packed));                   /* the vectors to be sent are complete once everything issued so far has run */
  hpgmg_hip_set_stream(comm_stream);
  HIP_OK(hpgmg_hip_stream_wait_event(ev_packed));
  pair_halo_exchange(L, B, first ? HALO_FIRST : HALO_NEXT, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id);
  HIP_OK(hpgmg_hip_event_record(ev_landed));
  hpgmg_hip_set_stream(launch_stream);
  overlap_count++;
  return 1;
}
/* one sweep-pair launch of a level with faces on other ranks: whole, or as its two parts around the arrival of the halo */
#define PAIR_REMOTE_LAUNCH(OVERLAPPED, DISCARD_X1, CALL) do {                                                   \
    pair_halo *H_ = B->halo;                                                                                   \
    if (OVERLAPPED) {                                                                                          \
      hpgmg_hip_pair_set_halo(H_->brick, H_->rem, H_->deep, H_->deep_beta); hpgmg_hip_set_tile_part(1);        \
      if (DISCARD_X1) hpgmg_hip_pair_discard_x1();                                                             \
      HIP_OK(CALL);                                                                                            \
      overlap_end();                                                                                           \
      hpgmg_hip_pair_set_halo(H_->brick, H_->rem, H_->deep, H_->deep_beta); hpgmg_hip_set_tile_part(2);        \
      if (DISCARD_X1) hpgmg_hip_pair_discard_x1();                                                             \
      HIP_OK(CALL);                                                                                            \
      hpgmg_hip_set_tile_part(0);                                                                              \
    } else {                                                                                                   \
      hpgmg_hip_pair_set_halo(H_->brick, H_->rem, H_->deep, H_->deep_beta);                                    \
      if (DISCARD_X1) hpgmg_hip_pair_discard_x1();                                                             \
      HIP_OK(CALL);                                                                                            \
    } } while (0)
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
/* common part: does the level qualify for the sweep-pair kernel, and are its two private vectors there? */
static int pair_kernel_ready(level_type *L, int x_id, int rhs_id, int sweeps) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  backend_t *B = hp_backend_of(L);
  if (!hp_switch(SW_FUSED_SWEEPS) || sweeps != 4 || cfg.op != HPGMG_OP_7PT || !hp_ghost_free_mode() || stencil_get_shape() != STENCIL_SHAPE_STAR) return 0;
  if (L->boundary_condition.type != BC_DIRICHLET || x_id == VECTOR_TEMP || rhs_id == VECTOR_TEMP) return 0;
  if (B->all_faces_local) { if (!hpgmg_hip_smooth_cheby_pair_supported(&B->dev, hp_variant()) || !boxes_lexicographic(L)) return 0; }
  else if (!pair_halo_ready(L, B)) return 0;          /* faces owned by other ranks: two-deep halo, one exchange per pair */
  { /* the pass structure pays when the level is bandwidth bound; a cache-resident level (128^3 and smaller) is latency
     * bound and faster with many small single-sweep workgroups (measured: 128^3 pair 80 us vs 2 x 27 us) */
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
static int smooth_cheby_pairs(level_type *L, int x_id, int rhs_id, double a, double b, const double *c1, const double *c2, int sweeps, int temp_dead) {
  if (!pair_kernel_ready(L, x_id, rhs_id, sweeps)) return 0;
  backend_t *B = hp_backend_of(L);
  const double h2inv = 1.0 / (L->h * L->h);
  const int v = hp_variant();
  const int remote = !B->all_faces_local;
  const float *const *c32 = remote ? NULL : coef32_of(L);      /* across ranks the coefficient streams stay fp64 */
  int over = 0;
  if (remote) { pair_remote_smooths++; over = pair_halo_begin(L, B, 1, 0, x_id, 0, VECTOR_TEMP, rhs_id); }
  { TICK(L, smooth, "smooth (Chebyshev sweeps 1+2)");
    if (remote) PAIR_REMOTE_LAUNCH(over, 0, hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, c32, 0, x_id, 0, VECTOR_TEMP, 1, 0, 1, 1, rhs_id, a, b, h2inv, c1[0], c2[0], c1[1], c2[1]));
    else HIP_OK(hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, c32, 0, x_id, 0, VECTOR_TEMP, 1, 0, 1, 1, rhs_id, a, b, h2inv, c1[0], c2[0], c1[1], c2[1]));
    TOCK(); }
  if (remote) over = pair_halo_begin(L, B, 0, 1, 1, 1, 0, rhs_id);
  { TICK(L, smooth, "smooth (Chebyshev sweeps 3+4)");
    if (remote) PAIR_REMOTE_LAUNCH(over, temp_dead, hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, c32, 1, 1, 1, 0, 0, VECTOR_TEMP, 0, x_id, rhs_id, a, b, h2inv, c1[2], c2[2], c1[3], c2[3]));
    else {
      if (temp_dead) hpgmg_hip_pair_discard_x1();
      HIP_OK(hpgmg_hip_smooth_cheby_pair(&B->dev, v, (double *const *)B->d_pair_base, c32, 1, 1, 1, 0, 0, VECTOR_TEMP, 0, x_id, rhs_id, a, b, h2inv, c1[2], c2[2], c1[3], c2[3]));
    }
    TOCK(); }
  return 1;
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
  if (remote) { pair_remote_smooths++; over = pair_halo_begin(L, B, 1, 0, x_id, 0, x_id, rhs_id); }
  { TICK(L, smooth, "smooth (GSRB half sweeps 1+2)");
    if (remote) PAIR_REMOTE_LAUNCH(over, 0, hpgmg_hip_smooth_gsrb_pair(&B->dev, v, (double *const *)B->d_pair_base, NULL, 0, x_id, 0, 1, 1, rhs_id, a, b, h2inv, 0));
    else HIP_OK(hpgmg_hip_smooth_gsrb_pair(&B->dev, v, (double *const *)B->d_pair_base, NULL, 0, x_id, 0, 1, 1, rhs_id, a, b, h2inv, 0));
    TOCK(); }
  if (remote) over = pair_halo_begin(L, B, 0, 1, 1, 1, 1, rhs_id);
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
static int interp_smooth_fused(level_type *Lf, int e_id, int R_id, level_type *Lc, double a, double b, int exact_state);
int hpgmg_interp_smooth_fused(level_type *Lf, int e_id, int R_id, level_type *Lc, double a, double b) { hp_lazy_flush(); return interp_smooth_fused(Lf, e_id, R_id, Lc, a, b, 0); }
/* exact_state: VECTOR_TEMP is left as smooth() leaves it (the lazy queue runs behind the reference's own driver, which promises nothing about it) */
static int interp_smooth_fused(level_type *Lf, int e_id, int R_id, level_type *Lc, double a, double b, int exact_state) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  const int sweeps = hpgmg_smooth_sweeps();
  communicator_type *S = &Lc->interpolation, *Rv = &Lf->interpolation;
  if (cfg.op != HPGMG_OP_7PT || !Lf->active || !Lc->active) return 0;
  if (cfg.smoother != HPGMG_SMOOTH_CHEBY && !(cfg.smoother == HPGMG_SMOOTH_GSRB && !hpgmg_gsrb_out_of_place())) return 0;
  if (S->num_sends || S->num_recvs || Rv->num_sends || Rv->num_recvs || S->num_blocks[0] || Rv->num_blocks[2]) return 0;   /* all parents local */
  if (Lf->box_dim % 128 != 0 || Lc->box_dim * 2 != Lf->box_dim || Lc->num_my_boxes != Lf->num_my_boxes || !boxes_lexicographic(Lc)) return 0;
  if (!hp_backend_of(Lf)->all_faces_local) return 0;       /* across ranks the pair kernel takes x0 as stored (interpolation stays its own launch) */
  if (!pair_kernel_ready(Lf, e_id, R_id, sweeps)) return 0;
  if (cfg.smoother == HPGMG_SMOOTH_CHEBY && Lf->dominant_eigenvalue_of_DinvA <= 0.0) return 0;
  hpgmg_hip_pair_fold_interpolation(&hp_backend_of(Lc)->dev, e_id, 1.0);
  if (cfg.smoother == HPGMG_SMOOTH_CHEBY) {
    double c1[16], c2[16];
    cheby_coefficients(Lf, sweeps, c1, c2);
    const int done = smooth_cheby_pairs(Lf, e_id, R_id, a, b, c1, c2, sweeps, !exact_state && hp_switch(SW_TEMP_SCRATCH));      /* the cycle hook: VECTOR_TEMP is dead afterwards */
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
int hpgmg_bottom_solve_fused(level_type *L, int e_id, int R_id, double a, double b, double want) {
  hpgmg_config cfg;
  const int on = (int)hp_switch(SW_FUSED_BOTTOM);
  hpgmg_get_config(&cfg);
  if (!on || cfg.op == HPGMG_OP_7PT || !L->active || L->num_my_boxes != 1 || L->boxes_in.i * L->boxes_in.j * L->boxes_in.k != 1) return 0;
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
  do_smooth(L, x_id, rhs_id, a, b, (int)hp_switch(SW_TEMP_SCRATCH));
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
  if (!scratch) { if (!exchange_and_bcs_one_launch(L, id, shape, 4, 0)) apply_BCs(L, id, shape); return; }
  int n = 0;
  const hpgmg_hip_bc_entry *e = bc_entries(L, shape, &n);
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
static void do_scale_vector(level_type *L, int c, double s, int a);
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
    e_k = bc_entries_k(L, &n_k, &k_local);
    (void)bc_entries(L, stencil_get_shape(), &n_all);
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
  if (passes == 1) do_scale_vector(L, x_id, 1.0, VECTOR_TEMP);      /* a single pass cannot land on its own input (never the case with the reference's counts) */
  fv4_rb_smooths++;
  return 1;
}
/* temp_dead: the caller declares VECTOR_TEMP scratch after this smooth() (inside a cycle: hpgmg_smooth_in_cycle, or the operator queue saw it
 * overwritten next) -- the in-cycle forms may run: the sweep pair without the x3 store, the red + black passes of the 27-point / fv4 GSRB smoothers */
static void do_smooth(level_type *L, int x_id, int rhs_id, double a, double b, int temp_dead) {
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
          else if (tiled && !exchange_and_bcs_one_launch(L, src, stencil_get_shape(), 12, 0)) apply_BCs(L, src, stencil_get_shape());
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

static void do_residual(level_type *L, int res_id, int x_id, int rhs_id, double a, double b) {   /* residual.c:9-51 */
  if (small_level_try(L, 3, x_id, rhs_id, res_id, a, b)) return;
  STENCIL_WITH_GHOSTS(L, x_id, res_id, residual, hpgmg_hip_residual(hp_stencil_dev(hp_backend_of(L)), hp_variant(), res_id, x_id, rhs_id, a, b, 1.0 / (L->h * L->h)));
}
static void do_apply_op(level_type *L, int Ax_id, int x_id, double a, double b) {               /* apply_op.c:9-48 */
  if (small_level_try(L, 4, x_id, -1, Ax_id, a, b)) return;
  STENCIL_WITH_GHOSTS(L, x_id, Ax_id, apply_op, hpgmg_hip_residual(hp_stencil_dev(hp_backend_of(L)), hp_variant(), Ax_id, x_id, -1, a, b, 1.0 / (L->h * L->h)));
}

/* ---------------------------------------------------------------- restriction.c:104-212 */
static void do_restriction(level_type *Lc, int id_c, level_type *Lf, int id_f, int type) {
  TICK(Lf, restriction_total, "restriction");
  communicator_type *S = &Lf->restriction[type], *R = &Lc->restriction[type];
  backend_t *Bc = hp_backend_of(Lc), *Bf = hp_backend_of(Lf);
  HIP_OK(hpgmg_hip_restrict_blocks(&Bc->dev, id_c, &Bf->dev, id_f, hp_mirror(Lf, S->blocks[0], S->num_blocks[0]), S->num_blocks[0], type));
  transport_phase(R, S, (Lf->tag << 4) | 0x5);
  HIP_OK(hpgmg_hip_restrict_blocks(&Bc->dev, id_c, &Bf->dev, id_f, hp_mirror(Lf, S->blocks[1], S->num_blocks[1]), S->num_blocks[1], type));
  HIP_OK(hpgmg_hip_copy_blocks(&Bc->dev, id_c, hp_mirror(Lc, R->blocks[2], R->num_blocks[2]), R->num_blocks[2]));
  TOCK();
}

/* restriction(Lc, id_c, Lf, id_f, RESTRICT_CELL) followed by zero_vector(Lc, zero_id) -- the end of MGVCycle's down-leg
 * (mg.c:1152-1153) -- as one launch when every contribution is local.  0 = the driver calls the two operators. */
int hpgmg_restrict_zero_fused(level_type *Lc, int id_c, level_type *Lf, int id_f, int zero_id) {
  communicator_type *S = &Lf->restriction[RESTRICT_CELL], *R = &Lc->restriction[RESTRICT_CELL];
  if (!Lf->active || !Lc->active || Lc->num_my_boxes < 1 || zero_id == id_c) return 0;
  if (S->num_sends || R->num_recvs || S->num_blocks[0] || R->num_blocks[2] || S->num_blocks[1] < 1) return 0;
  TICK(Lf, restriction_total, "restriction + zero_vector");
  backend_t *Bc = hp_backend_of(Lc), *Bf = hp_backend_of(Lf);
  HIP_OK(hpgmg_hip_restrict_cell_and_zero(&Bc->dev, id_c, &Bf->dev, id_f, hp_mirror(Lf, S->blocks[1], S->num_blocks[1]), S->num_blocks[1], zero_id));
  TOCK();
  return 1;
}

/* residual(Lf, TEMP, x, rhs) ; restriction(Lc, id_c, Lf, TEMP, RESTRICT_CELL) ; zero_vector(Lc, zero_id) -- the end of MGVCycle's down
 * leg (mg.c:1150-1153) -- as ONE pass over the fine level: the residual is restricted on the fly and never stored (VECTOR_TEMP of the
 * fine level keeps its previous content; nothing reads it before the up leg's smooth() overwrites it).  0 = not applicable. */
static double allreduce_scalar(level_type *L, double v, int op);
/* per fine box: the coarse box it restricts into and the coarse cell under its first cell -- read off the local restriction list */
static const int *restrict_map_of(level_type *Lf, backend_t *Bf) {
  communicator_type *S = &Lf->restriction[RESTRICT_CELL];
  if (!Bf->d_restrict_map) {
    int *map = (int *)malloc((size_t)Lf->num_my_boxes * 4 * sizeof(int)), n, b;
    for (b = 0; b < 4 * Lf->num_my_boxes; b++) map[b] = -1;
    for (n = 0; n < S->num_blocks[1]; n++) {
      const blockCopy_type *e = &S->blocks[1][n];
      if (e->read.box < 0 || e->write.box < 0 || map[4 * e->read.box] >= 0) continue;
      map[4 * e->read.box] = e->write.box;
      map[4 * e->read.box + 1] = e->write.i - e->read.i / 2; map[4 * e->read.box + 2] = e->write.j - e->read.j / 2; map[4 * e->read.box + 3] = e->write.k - e->read.k / 2;
    }
    for (b = 0; b < Lf->num_my_boxes; b++) if (map[4 * b] < 0) { free(map); return NULL; }
    Bf->d_restrict_map = (int *)hpgmg_hip_malloc((size_t)Lf->num_my_boxes * 4 * sizeof(int));
    if (!Bf->d_restrict_map) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
    HIP_OK(hpgmg_hip_memcpy_h2d(Bf->d_restrict_map, map, (size_t)Lf->num_my_boxes * 4 * sizeof(int)));
    free(map);
  }
  return Bf->d_restrict_map;
}
static int fused_residual_on(void) {
  return (int)hp_switch(SW_FUSED_RESIDUAL);
}
/* the operand of a fused residual form is made ready: 7-point -- nothing (the kernel applies the Dirichlet rule and reads neighbouring boxes);
 * 27-point / fv4 -- the domain-boundary ghost cells (the tiled kernel reads neighbouring boxes itself).  0 = the level does not qualify. */
static int fused_residual_operand(level_type *L, backend_t *B, int x_id) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  const int shape = stencil_get_shape();
  if (!fused_residual_on() || !hp_ghost_free_mode() || !L->active || L->num_my_boxes < 1 || L->boundary_condition.type == BC_PERIODIC) return 0;
  B->img_active = 0;
  if (!B->all_faces_local) {           /* 27-point / fv4 with boxes on other ranks: the same pass on the table with their images */
    if ((cfg.op != HPGMG_OP_27PT && cfg.op != HPGMG_OP_FV4) || !hp_images_ready(L, B) || !hpgmg_hip_residual_fused_supported(&B->img->dev, hp_variant())) return 0;
    hpgmg_hip_set_ghost_free(0);
    hpgmg_hip_set_tile_ghost_free(1);
    hp_images_refresh(L, B, 0, x_id, stencil_get_radius(), -1, cfg.op == HPGMG_OP_27PT ? 12 : 4);
    return 1;
  }
  if (cfg.op == HPGMG_OP_7PT) {
    if (shape != STENCIL_SHAPE_STAR) return 0;
    hpgmg_hip_set_ghost_free(1);
    return hpgmg_hip_residual_fused_supported(&B->dev, hp_variant());
  }
  if (cfg.op != HPGMG_OP_27PT && cfg.op != HPGMG_OP_FV4) return 0;
  if (!hpgmg_hip_residual_fused_supported(&B->dev, hp_variant())) return 0;
  hpgmg_hip_set_ghost_free(0);
  hpgmg_hip_set_tile_ghost_free(1);
  if (!exchange_and_bcs_one_launch(L, x_id, shape, cfg.op == HPGMG_OP_27PT ? 12 : 4, 0)) apply_BCs(L, x_id, shape);
  return 1;
}
static int residual_restrict_zero_fused(level_type *Lc, int id_c, level_type *Lf, int res_id, int x_id, int rhs_id, double a, double b, int zero_id);
int hpgmg_residual_restrict_zero_fused(level_type *Lc, int id_c, level_type *Lf, int x_id, int rhs_id, double a, double b, int zero_id) {
  hp_lazy_flush();
  return residual_restrict_zero_fused(Lc, id_c, Lf, -1, x_id, rhs_id, a, b, zero_id);
}
/* res_id >= 0: the residual is stored as well (7-point): the exact state of the three operators */
static int residual_restrict_zero_fused(level_type *Lc, int id_c, level_type *Lf, int res_id, int x_id, int rhs_id, double a, double b, int zero_id) {
  communicator_type *S = &Lf->restriction[RESTRICT_CELL], *R = &Lc->restriction[RESTRICT_CELL];
  if (!Lf->active || !Lc->active || Lc->num_my_boxes < 1 || Lf->num_my_boxes < 1 || zero_id == id_c) return 0;
  if (S->num_sends || R->num_recvs || S->num_blocks[0] || R->num_blocks[2] || S->num_blocks[1] < 1) return 0;
  backend_t *Bc = hp_backend_of(Lc), *Bf = hp_backend_of(Lf);
  if (!restrict_map_of(Lf, Bf)) return 0;
  { hpgmg_config cfg; hpgmg_get_config(&cfg); if (res_id >= 0 && (cfg.op != HPGMG_OP_7PT || res_id == x_id || res_id == rhs_id)) return 0; }
  if (!fused_residual_operand(Lf, Bf, x_id)) return 0;
  TICK(Lf, residual, "residual + restriction + zero_vector (fused)");
  HIP_OK(hpgmg_hip_residual_restrict_store(hp_stencil_dev(Bf), hp_variant(), res_id, x_id, rhs_id, a, b, 1.0 / (Lf->h * Lf->h), &Bc->dev, id_c, Bf->d_restrict_map, zero_id));
  TOCK();
  return 1;
}
/* norm(L, F) ; scale_vector(L, R, 1.0, F) ; restriction(Lc, R, L, R, RESTRICT_CELL) -- how FMGSolve starts (mg.c:1262-1270) -- in one pass over F.
 * Every rank's share of the restriction must be local.  0 = not applicable. */
static int norm_scale_restrict_fused(level_type *L, int F_id, int R_id, level_type *Lc, double *norm_out) {
  communicator_type *S = &L->restriction[RESTRICT_CELL], *R = &Lc->restriction[RESTRICT_CELL];
  if (!fused_residual_on() || !L->active || !Lc->active || L->num_my_boxes < 1 || Lc->num_my_boxes < 1 || F_id == R_id) return 0;
  if (S->num_sends || R->num_recvs || S->num_blocks[0] || R->num_blocks[2] || S->num_blocks[1] < 1) return 0;
  backend_t *Bc = hp_backend_of(Lc), *B = hp_backend_of(L);
  if ((L->box_dim & 1) || !(B->dev.flags & 1) || (L->box_jStride & 1) || (L->box_kStride & 1) || (L->box_volume & 1) || L->box_dim < 16) return 0;
  if (!restrict_map_of(L, B)) return 0;
  double v = 0.0;
  { TICK(L, blas1, norm_out ? "norm(F) + R = F + restriction (fused)" : "R = F + restriction (fused)");
    HIP_OK(hpgmg_hip_norm_copy_restrict(&B->dev, F_id, R_id, &Bc->dev, R_id, B->d_restrict_map, norm_out ? &v : NULL));
    TOCK(); }
  if (norm_out) *norm_out = allreduce_scalar(L, v, HPGMG_REDUCE_MAX);
  return 1;
}
int hpgmg_norm_scale_restrict_fused(level_type *L, int F_id, int R_id, level_type *Lc, double *norm_out) { return norm_scale_restrict_fused(L, F_id, R_id, Lc, norm_out); }
/* residual(L, res, x, rhs) ; norm(L, res) -- the convergence check of MGSolve / FMGSolve (mg.c:1321-1323) -- in one pass: the residual is
 * stored as usual (res_id < 0: not stored -- the cycle driver's check, after which VECTOR_TEMP is dead) and its max-abs comes out of the
 * same kernel.  0 = not applicable. */
int hpgmg_residual_norm_fused(level_type *L, int res_id, int x_id, int rhs_id, double a, double b, double *norm_out) {
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (cfg.op != HPGMG_OP_7PT && res_id >= 0) return 0;          /* the tiled kernels of the other plugins only carry the norm-only form */
  if (!L->active || L->num_my_boxes < 1) return 0;
  backend_t *B = hp_backend_of(L);
  if (!fused_residual_operand(L, B, x_id)) return 0;
  double v = 0.0;
  { TICK(L, residual, "residual + norm (fused)");
    HIP_OK(hpgmg_hip_residual_norm(hp_stencil_dev(B), hp_variant(), res_id, x_id, rhs_id, a, b, 1.0 / (L->h * L->h), &v));
    TOCK(); }
  *norm_out = allreduce_scalar(L, v, HPGMG_REDUCE_MAX);
  return 1;
}

/* ---------------------------------------------------------------- interpolation_p0.c:52-159, interpolation_p1.c:70-180 */
static void interpolation_lists(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c, int order, int tagbits) {
  TICK(Lf, interpolation_total, "interpolation");
  communicator_type *S = &Lc->interpolation, *R = &Lf->interpolation;
  backend_t *Bc = hp_backend_of(Lc), *Bf = hp_backend_of(Lf);
  HIP_OK(hpgmg_hip_interpolate_blocks(&Bf->dev, id_f, 0.0, &Bc->dev, id_c, hp_mirror(Lc, S->blocks[0], S->num_blocks[0]), S->num_blocks[0], order));
  transport_phase(R, S, (Lf->tag << 4) | tagbits);
  HIP_OK(hpgmg_hip_interpolate_blocks(&Bf->dev, id_f, prescale, &Bc->dev, id_c, hp_mirror(Lc, S->blocks[1], S->num_blocks[1]), S->num_blocks[1], order));
  HIP_OK(hpgmg_hip_increment_blocks(&Bf->dev, id_f, prescale, hp_mirror(Lf, R->blocks[2], R->num_blocks[2]), R->num_blocks[2]));
  TOCK();
}
static void do_interpolation_vcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  hpgmg_config c; hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_27PT) {                                  /* interpolation_p2.c:228-230 */
    if (!exchange_and_bcs_one_launch(Lc, id_c, STENCIL_SHAPE_BOX, 12, 1)) { exchange_boundary(Lc, id_c, STENCIL_SHAPE_BOX); apply_BCs_p2(Lc, id_c, STENCIL_SHAPE_BOX); }
    interpolation_lists(Lf, id_f, prescale, Lc, id_c, 2, 0x7);
    return;
  }
  if (c.op == HPGMG_OP_FV2 || c.op == HPGMG_OP_FV4) {           /* interpolation_v2.c:210-212 (V-cycle of fv2 and fv4) */
    if (!exchange_and_bcs_one_launch(Lc, id_c, STENCIL_SHAPE_BOX, 2, 1)) { exchange_boundary(Lc, id_c, STENCIL_SHAPE_BOX); apply_BCs_v2(Lc, id_c, STENCIL_SHAPE_BOX); }
    interpolation_lists(Lf, id_f, prescale, Lc, id_c, 3, 0x7);
    return;
  }
  if (c.op != HPGMG_OP_7PT) no_kernel("interpolation_vcycle for this operator");
  interpolation_lists(Lf, id_f, prescale, Lc, id_c, 0, 0x6);
}
void interpolation_fcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  hpgmg_config c; hpgmg_get_config(&c);
  if (c.op == HPGMG_OP_27PT || c.op == HPGMG_OP_FV2) { do_interpolation_vcycle(Lf, id_f, prescale, Lc, id_c); return; } /* operators.27pt.c:150-151, .fv2.c:151-152 */
  if (c.op == HPGMG_OP_FV4) {                                   /* interpolation_v4.c:276-278 */
    if (!exchange_and_bcs_one_launch(Lc, id_c, STENCIL_SHAPE_BOX, 4, 1)) { exchange_boundary(Lc, id_c, STENCIL_SHAPE_BOX); apply_BCs_v4(Lc, id_c, STENCIL_SHAPE_BOX); }
    interpolation_lists(Lf, id_f, prescale, Lc, id_c, 4, 0x7);
    return;
  }
  if (c.op != HPGMG_OP_7PT) no_kernel("interpolation_fcycle for this operator");
  exchange_boundary(Lc, id_c, STENCIL_SHAPE_BOX);
  apply_BCs_p1(Lc, id_c, STENCIL_SHAPE_BOX);
  interpolation_lists(Lf, id_f, prescale, Lc, id_c, 1, 0x7);
}

/* ---------------------------------------------------------------- misc.c */
#define BLAS1(call) do { TICK(L, blas1, "BLAS1"); HIP_OK(call); TOCK(); } while (0)
static void do_zero_vector(level_type *L, int id) { BLAS1(hpgmg_hip_fill(&hp_backend_of(L)->dev, id, 0.0)); }
void init_vector(level_type *L, int id, double s) { BLAS1(hpgmg_hip_fill(&hp_backend_of(L)->dev, id, s)); }
static void do_add_vectors(level_type *L, int c, double sa, int a, double sb, int b) { BLAS1(hpgmg_hip_axpby(&hp_backend_of(L)->dev, c, sa, a, sb, b)); }
static void do_mul_vectors(level_type *L, int c, double s, int a, int b) { BLAS1(hpgmg_hip_mul(&hp_backend_of(L)->dev, c, s, a, b)); }
void invert_vector(level_type *L, int c, double s, int a) { BLAS1(hpgmg_hip_invert(&hp_backend_of(L)->dev, c, s, a)); }
static void do_scale_vector(level_type *L, int c, double s, int a) { BLAS1(hpgmg_hip_scale(&hp_backend_of(L)->dev, c, s, a)); }
void shift_vector(level_type *L, int c, int a, double shift) { BLAS1(hpgmg_hip_shift(&hp_backend_of(L)->dev, c, a, shift)); }
void color_vector(level_type *L, int id, int colors, int ic, int jc, int kc) { BLAS1(hpgmg_hip_color(&hp_backend_of(L)->dev, id, colors, ic, jc, kc)); }
void random_vector(level_type *L, int id) { BLAS1(hpgmg_hip_random(&hp_backend_of(L)->dev, id)); }

static double allreduce_scalar(level_type *L, double v, int op) {
  const hpgmg_transport *T = hpgmg_get_transport();
  if (T && T->size > 1) {
    hpgmg_level_ext *X = hpgmg_level_ext_get(L);
    if (X->num_active_ranks > 1) { const double t0 = now(); T->allreduce(T->ctx, &v, 1, op, X->active_ranks, X->num_active_ranks); L->timers.collectives += now() - t0; }   /* host-synchronous by nature: host clock in every mode */
  }
  return v;
}
static double do_dot(level_type *L, int a, int b) { double v; BLAS1(hpgmg_hip_dot(&hp_backend_of(L)->dev, a, b, &v)); return allreduce_scalar(L, v, HPGMG_REDUCE_SUM); }
static double do_norm(level_type *L, int a) {
  double v; BLAS1(hpgmg_hip_norm_max(&hp_backend_of(L)->dev, a, &v)); return allreduce_scalar(L, v, HPGMG_REDUCE_MAX); }
double mean(level_type *L, int a) {
  double v; BLAS1(hpgmg_hip_sum(&hp_backend_of(L)->dev, a, &v));
  v = allreduce_scalar(L, v, HPGMG_REDUCE_SUM);
  return v / (double)((double)L->dim.i * (double)L->dim.j * (double)L->dim.k);
}
double error(level_type *L, int a, int b) { add_vectors(L, VECTOR_TEMP, 1.0, a, -1.0, b); return norm(L, VECTOR_TEMP); }

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
  coef32_invalidate(L);
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
  coef32_invalidate(L);
  hpgmg_config cfg;
  hpgmg_get_config(&cfg);
  if (cfg.op != HPGMG_OP_7PT) {                                 /* operators.27pt.c:96-121, .fv2.c:98-124, .fv4.c:145-172 */
    if (from) {
      if (cfg.helmholtz) do_restriction(L, VECTOR_ALPHA, from, VECTOR_ALPHA, RESTRICT_CELL);
      do_restriction(L, VECTOR_BETA_I, from, VECTOR_BETA_I, RESTRICT_FACE_I);
      do_restriction(L, VECTOR_BETA_J, from, VECTOR_BETA_J, RESTRICT_FACE_J);
      do_restriction(L, VECTOR_BETA_K, from, VECTOR_BETA_K, RESTRICT_FACE_K);
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
  if (cfg.op != HPGMG_OP_7PT) no_kernel("rebuild_operator for this operator");
  if (L->my_rank == 0 && hpgmg_verbose) { fprintf(stdout, "  rebuilding operator for level...  h=%e  ", L->h); fflush(stdout); }
  if (from) {
    if (cfg.helmholtz) do_restriction(L, VECTOR_ALPHA, from, VECTOR_ALPHA, RESTRICT_CELL);
    do_restriction(L, VECTOR_BETA_I, from, VECTOR_BETA_I, RESTRICT_FACE_I);
    do_restriction(L, VECTOR_BETA_J, from, VECTOR_BETA_J, RESTRICT_FACE_J);
    do_restriction(L, VECTOR_BETA_K, from, VECTOR_BETA_K, RESTRICT_FACE_K);
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
void hpgmg_operators_flush(void) { hp_lazy_flush(); }      /* issue every postponed operator now (nothing is ever left behind: any other call does the same) */
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
static void small_ops_forget(void) { so_npred = 0; so_last_fresh = 0; so_cache_valid = 0; }      /* a level is going away: the remembered requests name it */
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
    case LZ_SMOOTH:   do_smooth(o->L, o->i0, o->i1, o->a, o->b, 0); break;
    case LZ_RESIDUAL: do_residual(o->L, o->i0, o->i1, o->i2, o->a, o->b); break;
    case LZ_RESTRICT: do_restriction(o->L, o->i0, o->L2, o->i1, o->i2); break;
    case LZ_ZERO:     do_zero_vector(o->L, o->i0); break;
    case LZ_INTERP:   do_interpolation_vcycle(o->L, o->i0, o->a, o->L2, o->i1); break;
    case LZ_SCALE:    do_scale_vector(o->L, o->i0, o->a, o->i1); break;
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
      if (m >= 2 && vcycle_legs_fused(chain, m, s0->i0, s0->i1, s0->a, s0->b, 0)) { lazy_fused_legs++; u = units; break; }
      /* this unit on its own: smooth, then residual + restriction + zero_vector in one pass where the level allows it.  The unit's next
       * operator is residual(VECTOR_TEMP, ...) (mg.c:1150), which overwrites what smooth() leaves in VECTOR_TEMP before anything can read it:
       * the queue has PROVED the vector dead, so the smoother may run in its in-cycle form (hpgmg_smooth_in_cycle: the sweep pair without the
       * x3 store, the 27-point / fv4 red + black passes) although the reference's driver never says so.  HPGMG_TEMP_SCRATCH=0 keeps the exact form. */
      { const lazy_op *sm = &lz[4 * u];
        if (hp_switch(SW_TEMP_SCRATCH) && sm->i0 != VECTOR_TEMP && sm->i1 != VECTOR_TEMP) { do_smooth(sm->L, sm->i0, sm->i1, sm->a, sm->b, 1); lazy_temp_proved_dead++; }
        else lazy_run_one(sm); }
      const lazy_op *r = &lz[4 * u + 1], *t = &lz[4 * u + 2], *z = &lz[4 * u + 3];
      if (residual_restrict_zero_fused(t->L, t->i0, r->L, r->i0, r->i1, r->i2, r->a, r->b, z->i0)) lazy_fused_units++;
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
      if (vcycle_legs_fused(chain, c, sm->i0, sm->i1, sm->a, sm->b, 1)) { lazy_fused_legs++; done = m; }
    }
    int u;
    for (u = done; u < units; u++) {
      const lazy_op *ip = &lz[2 * u], *sm = &lz[2 * u + 1];
      if (interp_smooth_fused(ip->L, sm->i0, sm->i1, ip->L2, sm->a, sm->b, 1)) lazy_fused_units++;
      else { lazy_run_one(ip); lazy_run_one(sm); }
    }
    q = 2 * units;
  } else if (mode == LZ_SMALL) {                          /* no dot product / norm came: the queue as one launch all the same */
    (void)small_ops_issue(lz[0].L, 0, 0, 0, 0, 0, 0, NULL);   /* (clears the queue and lz_busy) */
    return;
  } else if (mode == LZ_SR && n == 2) {                   /* R = 1.0 * F, then its restriction: one pass over F (the norm the kernel also forms is not asked for) */
    if (norm_scale_restrict_fused(lz[0].L, lz[0].i1, lz[0].i0, lz[1].L, NULL)) { lazy_fused_units++; q = 2; }
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
  do_smooth(L, x_id, rhs_id, a, b, 0);
}
void residual(level_type *L, int res_id, int x_id, int rhs_id, double a, double b) {
  if (lazy_push(LZ_RESIDUAL, L, NULL, res_id, x_id, rhs_id, a, b)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_RESIDUAL, L, NULL, res_id, x_id, rhs_id, a, b)) return;      /* it may start the residual + norm pattern (the convergence check after the last V-cycle) */
  do_residual(L, res_id, x_id, rhs_id, a, b);
}
void restriction(level_type *Lc, int id_c, level_type *Lf, int id_f, int type) {
  if (lazy_push(LZ_RESTRICT, Lc, Lf, id_c, id_f, type, 0.0, 0.0)) return;
  hp_lazy_flush();
  do_restriction(Lc, id_c, Lf, id_f, type);
}
void scale_vector(level_type *L, int c, double s, int a) {
  if (lazy_push(LZ_SCALE, L, NULL, c, a, 0, s, 0.0)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_SCALE, L, NULL, c, a, 0, s, 0.0)) return;
  do_scale_vector(L, c, s, a);
}
void zero_vector(level_type *L, int id) {
  if (lazy_push(LZ_ZERO, L, NULL, id, 0, 0, 0.0, 0.0)) return;
  hp_lazy_flush();
  do_zero_vector(L, id);
}
void add_vectors(level_type *L, int c, double sa, int a, double sb, int b) {
  if (lazy_push(LZ_ADD, L, NULL, c, a, b, sa, sb)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_ADD, L, NULL, c, a, b, sa, sb)) return;
  do_add_vectors(L, c, sa, a, sb, b);
}
void mul_vectors(level_type *L, int c, double s, int a, int b) {
  if (lazy_push(LZ_MUL, L, NULL, c, a, b, s, 0.0)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_MUL, L, NULL, c, a, b, s, 0.0)) return;
  do_mul_vectors(L, c, s, a, b);
}
void apply_op(level_type *L, int Ax_id, int x_id, double a, double b) {
  if (lazy_push(LZ_APPLY, L, NULL, Ax_id, x_id, 0, a, b)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_APPLY, L, NULL, Ax_id, x_id, 0, a, b)) return;
  do_apply_op(L, Ax_id, x_id, a, b);
}
static int small_value_request(level_type *L, int kind, int a, int b, double *out);
double dot(level_type *L, int a, int b) {
  { double v; if (small_value_request(L, 6, a, b, &v)) return allreduce_scalar(L, v, HPGMG_REDUCE_SUM); }
  return do_dot(L, a, b);
}
void interpolation_vcycle(level_type *Lf, int id_f, double prescale, level_type *Lc, int id_c) {
  if (lazy_push(LZ_INTERP, Lf, Lc, id_f, id_c, 0, prescale, 0.0)) return;
  hp_lazy_flush();
  if (lazy_push(LZ_INTERP, Lf, Lc, id_f, id_c, 0, prescale, 0.0)) return;
  do_interpolation_vcycle(Lf, id_f, prescale, Lc, id_c);
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
  { double v; if (small_value_request(L, 7, a, 0, &v)) return allreduce_scalar(L, v, HPGMG_REDUCE_MAX); }
  if (lz_mode == LZ_RN && lz_n == 1 && !lz_busy && lz[0].L == L && lz[0].i0 == a) {       /* residual(a, ...) then norm(a): one pass, the residual stored as usual */
    const lazy_op o = lz[0];
    double v = 0.0;
    lz_n = 0; lz_mode = LZ_NONE;
    if (hpgmg_residual_norm_fused(L, o.i0, o.i1, o.i2, o.a, o.b, &v)) { lazy_fused_units++; return v; }
    do_residual(L, o.i0, o.i1, o.i2, o.a, o.b);
  }
  return do_norm(L, a);
}

/*
 * plugin_runtime.c -- timers of the timing table, storage hooks, transport selection, hipGraph segments, the per-level device record and its mirrors.
 * Part of the operator plugin (see operators_hip.c); no arithmetic on vector data happens here.
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
double hp_now(void) {
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
  else t.t0 = hp_now();
  return t;
}
void hpgmg_tick_end(hpgmg_tick t) {
  if (hpgmg_get_timer_mode() == TIMERS_DEVICE) hpgmg_hip_timer_end(t.slot);
  else if (t.acc) *t.acc += hp_now() - t.t0;
  if (t.range) hpgmg_hip_range_pop();
}
void hpgmg_timers_settle(void) { hpgmg_hip_timer_flush(); }

/* ---------------------------------------------------------------- storage hooks */
const char *hpgmg_backend_name(void) { return "hip"; }
double *hpgmg_vector_alloc(size_t n) {
  double *p = (double *)hpgmg_hip_malloc(n * sizeof(double));
  if (!p) { fprintf(stderr, "hpgmg: device allocation of %zu doubles failed: %s\n", n, hpgmg_hip_last_error()); abort(); }
  return p;
}
/* a postponed operator may still hold this storage, and so may a captured launch graph (create_vectors() growing a level after solves have run:
 * FMGSolve -> MGPCG -> FMGSolve would otherwise replay graphs over the freed slab) */
void hpgmg_vector_free(double *p) { hp_lazy_flush(); hpgmg_hip_graph_reset(); hpgmg_hip_free(p); }
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
  t.prepare_subset = hpgmg_hip_rccl_prepare_subset;
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
  t.prepare_subset = NULL;      /* peer copies: a subset reduction is one round of 8-byte copies among its members already */
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

void hpgmg_level_release(level_type *L) {
  hp_lazy_flush();                                  /* postponed operators hold a pointer to their level */
  hp_small_ops_forget();                            /* ... and so do the remembered scalar requests */
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

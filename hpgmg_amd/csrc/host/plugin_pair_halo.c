/*
 * plugin_pair_halo.c -- the two-deep halo of a sweep pair across rank boundaries: message plans, the exchange, its overlapped form.
 * Part of the operator plugin (see operators_hip.c); no arithmetic on vector data happens here.
 */
#include "plugin_internal.h"

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
    { int q; for (q = 0; q < P->n_sp; q++) P->sp_ptr[q] = (double *)(uintptr_t)so[q]; for (q = 0; q < P->n_rp; q++) P->rp_ptr[q] = (double *)(uintptr_t)ro[q]; }   /* offsets for hp_now */
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
int hp_pair_halo_ready(level_type *L, backend_t *B) {
  if (B->halo_state == 0) {
    const hpgmg_transport *T = hpgmg_get_transport();
    int lo[3], n[3], b, ok;
    B->halo_state = -1;
    ok = pair_remote_enabled() && T && T->size > 1 && L->boundary_condition.type == BC_DIRICHLET && (L->box_dim % 128 == 0 || (128 % L->box_dim == 0 && L->box_dim >= 16)) && L->num_my_boxes > 0;      /* (boxes narrower than a 128-cell row: several per row) */
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
 * -- the caller issues part 1, hp_overlap_end(), part 2, each after hpgmg_hip_pair_set_halo() (consumed per launch) -- and 0 when the exchange was done
 * in line (HPGMG_OVERLAP=0): one whole launch. */
int hp_pair_halo_begin(level_type *L, backend_t *B, int first, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id) {
  pair_halo *H = B->halo;
  if (!H->coef_valid) { pair_halo_exchange(L, B, HALO_COEF, 0, 0, 0, 0, 0); H->coef_valid = 1; }
  if (!hp_overlap_enabled() || hpgmg_get_timer_mode() == TIMERS_SYNC) {
    pair_halo_exchange(L, B, first ? HALO_FIRST : HALO_NEXT, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id);
    return 0;
  }
  if (!hp_comm_stream) {
    hp_comm_stream = hpgmg_hip_stream_create(); hp_ev_packed = hpgmg_hip_event_create(); hp_ev_landed = hpgmg_hip_event_create();
    if (!hp_comm_stream || !hp_ev_packed || !hp_ev_landed) { fprintf(stderr, "hpgmg: cannot create the exchange stream\n"); abort(); }
  }
  void *launch_stream = hpgmg_hip_get_stream();
  HIP_OK(hpgmg_hip_event_record(hp_ev_packed));                   /* the vectors to be sent are complete once everything issued so far has run */
  hpgmg_hip_set_stream(hp_comm_stream);
  HIP_OK(hpgmg_hip_stream_wait_event(hp_ev_packed));
  pair_halo_exchange(L, B, first ? HALO_FIRST : HALO_NEXT, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id);
  HIP_OK(hpgmg_hip_event_record(hp_ev_landed));
  hpgmg_hip_set_stream(launch_stream);
  hp_overlap_count++;
  return 1;
}
/* one sweep-pair launch of a level with faces on other ranks: whole, or as its two parts around the arrival of the halo */

/*
 * halo_images.c -- images of the neighbouring ranks' boxes (plugin_internal.h explains the idea): what lets the one-pass red + black
 * kernels, the LDS-tiled kernels and the fused residual passes of the 27-point and fv4 operators keep their single-GPU structure when
 * boxes live on several ranks.  Replaces, for those launches, the pack / MPI / unpack sequence of operators/exchange_boundary.c:12-117
 * (one message per neighbouring rank, as there) and moves the boundary conditions of operators/boundary_fv.c / boundary_fd.c onto the
 * images as well, so that a cell outside the domain is formed from the same interior cells with the same formula wherever it is read.
 *
 * No arithmetic on vector data happens here: this file builds tables and message plans and sequences launches.
 */
#include "plugin_internal.h"

long long hp_images_exchanges = 0;
void hpgmg_set_images(int on) { hp_switch_set(SW_IMAGES, on ? 1 : 0); }             /* tests: 0 = the exchange_boundary path on every level */
long long hpgmg_image_exchanges(void) { return hp_images_exchanges; }

static int table_find(void *ctx, int gid) {
  const halo_images *I = (const halo_images *)ctx;
  int q;
  for (q = 0; q < I->n_all; q++) if (I->gid[q] == gid) return q;
  return -1;
}
static void box_position(const level_type *L, int gid, int p[3]) {
  p[0] = gid % L->boxes_in.i; p[1] = (gid / L->boxes_in.i) % L->boxes_in.j; p[2] = gid / (L->boxes_in.i * L->boxes_in.j);
}

/* Every rank's boxes form a brick?  From the global box -> rank table, so all ranks reach the same answer. */
static int all_bricks(const level_type *L, int nr, int *blo, int *bn) {
  int *mx = (int *)malloc((size_t)nr * 3 * sizeof(int)), *cnt = (int *)calloc((size_t)nr, sizeof(int));
  int r, bi, bj, bk, a, ok = 1;
  for (r = 0; r < 3 * nr; r++) { blo[r] = 1 << 30; mx[r] = -1; }
  for (bk = 0; bk < L->boxes_in.k; bk++) for (bj = 0; bj < L->boxes_in.j; bj++) for (bi = 0; bi < L->boxes_in.i; bi++) {
    const int c[3] = { bi, bj, bk };
    r = hp_box_rank_at(L, bi, bj, bk);
    if (r < 0 || r >= nr) { ok = 0; continue; }
    cnt[r]++;
    for (a = 0; a < 3; a++) { if (c[a] < blo[3 * r + a]) blo[3 * r + a] = c[a]; if (c[a] > mx[3 * r + a]) mx[3 * r + a] = c[a]; }
  }
  for (r = 0; r < nr; r++) {
    if (cnt[r] == 0) { for (a = 0; a < 3; a++) { blo[3 * r + a] = 0; bn[3 * r + a] = 0; } continue; }
    for (a = 0; a < 3; a++) bn[3 * r + a] = mx[3 * r + a] - blo[3 * r + a] + 1;
    if (cnt[r] != bn[3 * r] * bn[3 * r + 1] * bn[3 * r + 2]) ok = 0;
  }
  free(mx); free(cnt);
  return ok;
}

/* the cells of the box at position p (in boxes; its own coordinates, ghost zone of G cells included) within `depth` cells of a brick */
static int clip_region(const level_type *L, const int p[3], const int *blo, const int *bn, int depth, int G, int lo[3], int len[3]) {
  const int dim = L->box_dim;
  int a;
  for (a = 0; a < 3; a++) {
    if (bn[a] <= 0) return 0;
    int l = blo[a] * dim - depth - p[a] * dim, h = (blo[a] + bn[a]) * dim + depth - p[a] * dim;
    if (l < -G) l = -G;
    if (h > dim + G) h = dim + G;
    if (h <= l) return 0;
    lo[a] = l; len[a] = h - l;
  }
  return 1;
}

typedef struct { int peer, gid, item; hpgmg_hip_halo_entry e; } plan_rec;
static int plan_rec_cmp(const void *pa, const void *pb) {
  const plan_rec *a = (const plan_rec *)pa, *b = (const plan_rec *)pb;
  if (a->peer != b->peer) return a->peer < b->peer ? -1 : 1;
  if (a->gid != b->gid) return a->gid < b->gid ? -1 : 1;
  return (a->item > b->item) - (a->item < b->item);
}
static size_t plan_finish(plan_rec *rec, int n, hpgmg_hip_halo_entry **d_list, int *n_msg, int **ranks, int **sizes, long long **offs) {
  int q, m = 0;
  size_t total = 0;
  qsort(rec, (size_t)n, sizeof(plan_rec), plan_rec_cmp);
  *ranks = (int *)malloc((size_t)(n + 1) * sizeof(int)); *sizes = (int *)calloc((size_t)(n + 1), sizeof(int)); *offs = (long long *)calloc((size_t)(n + 1), sizeof(long long));
  hpgmg_hip_halo_entry *host = (hpgmg_hip_halo_entry *)malloc((size_t)(n + 1) * sizeof(*host));
  for (q = 0; q < n; q++) {
    if (m == 0 || (*ranks)[m - 1] != rec[q].peer) { (*ranks)[m] = rec[q].peer; (*offs)[m] = (long long)total; m++; }
    rec[q].e.off = (long long)total;
    const int len = rec[q].e.ni * rec[q].e.nj * rec[q].e.nk;
    (*sizes)[m - 1] += len; total += (size_t)len;
    host[q] = rec[q].e;
  }
  *n_msg = m; *d_list = NULL;
  if (n > 0) {
    *d_list = (hpgmg_hip_halo_entry *)hpgmg_hip_malloc((size_t)n * sizeof(*host));
    if (!*d_list) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
    HIP_OK(hpgmg_hip_memcpy_h2d(*d_list, host, (size_t)n * sizeof(*host)));
  }
  free(host);
  return total;
}

/* Plan `which`: what this rank sends (cells of its boxes that other ranks hold images of) and receives (into its images).  Both sides
 * evaluate clip_region() for the same (box, receiving rank) pairs and order a message by (global box id, item), like level.c derives
 * its message layout on both sides independently. */
static void plan_build(level_type *L, halo_images *I, int which) {
  const hpgmg_transport *T = hpgmg_get_transport();
  const int me = L->my_rank, nr = T->size;
  image_plan *P = &I->plan[which];
  int items[8][3], nitems = 0, it, b, q, r;      /* (vec, depth, ghost rim) */
  if (which == IMG_PLAN_COEF) {
    const int G = L->box_ghosts;
    items[nitems][0] = 16 + VECTOR_DINV;   items[nitems][1] = I->depth_max; items[nitems++][2] = G;
    items[nitems][0] = 16 + VECTOR_BETA_I; items[nitems][1] = I->depth_max; items[nitems++][2] = G;
    items[nitems][0] = 16 + VECTOR_BETA_J; items[nitems][1] = I->depth_max; items[nitems++][2] = G;
    items[nitems][0] = 16 + VECTOR_BETA_K; items[nitems][1] = I->depth_max; items[nitems++][2] = G;
    if (hpgmg_vectors_reserved() > VECTOR_ALPHA) { items[nitems][0] = 16 + VECTOR_ALPHA; items[nitems][1] = I->depth_max; items[nitems++][2] = G; }
  } else {
    items[nitems][0] = 0; items[nitems][1] = which % 3 + 1; items[nitems++][2] = 0;
    if (which >= 3) { items[nitems][0] = 2; items[nitems][1] = 1; items[nitems++][2] = 0; }
  }
  const int cap_s = (I->n_real * nr + 1) * nitems, cap_r = (I->n_img + 1) * nitems;
  plan_rec *snd = (plan_rec *)calloc((size_t)cap_s, sizeof(plan_rec)), *rcv = (plan_rec *)calloc((size_t)cap_r, sizeof(plan_rec));
  int ns = 0, nrv = 0;
  for (it = 0; it < nitems; it++) {
    int lo[3], len[3], p[3];
    for (q = 0; q < I->n_img; q++) {               /* what I receive */
      const int t = I->n_real + q;
      box_position(L, I->gid[t], p);
      if (!clip_region(L, p, I->lo, I->n, items[it][1], items[it][2], lo, len)) continue;
      plan_rec *R = &rcv[nrv++];
      R->peer = L->rank_of_box[I->gid[t]]; R->gid = I->gid[t]; R->item = it;
      R->e.box = t; R->e.vec = items[it][0]; R->e.deep = -1;
      R->e.i = lo[0]; R->e.j = lo[1]; R->e.k = lo[2]; R->e.ni = len[0]; R->e.nj = len[1]; R->e.nk = len[2];
    }
    for (b = 0; b < I->n_real; b++) for (r = 0; r < nr; r++) {      /* what I send */
      if (r == me) continue;
      box_position(L, I->gid[b], p);
      if (!clip_region(L, p, I->brick_lo + 3 * r, I->brick_n + 3 * r, items[it][1], items[it][2], lo, len)) continue;
      plan_rec *R = &snd[ns++];
      R->peer = r; R->gid = I->gid[b]; R->item = it;
      R->e.box = b; R->e.vec = items[it][0]; R->e.deep = -1;
      R->e.i = lo[0]; R->e.j = lo[1]; R->e.k = lo[2]; R->e.ni = len[0]; R->e.nj = len[1]; R->e.nk = len[2];
    }
  }
  const size_t ts = plan_finish(snd, ns, &P->d_send, &P->n_sp, &P->sp_rank, &P->sp_size, &P->sp_off);
  const size_t tr = plan_finish(rcv, nrv, &P->d_recv, &P->n_rp, &P->rp_rank, &P->rp_size, &P->rp_off);
  P->n_send = ns; P->n_recv = nrv; P->built = 1;
  free(snd); free(rcv);
  if (ts > I->send_cap) { if (I->sendbuf) { hpgmg_hip_sync(); hpgmg_hip_free(I->sendbuf); } I->sendbuf = hpgmg_vector_alloc(ts + 2); I->send_cap = ts; }
  if (tr > I->recv_cap) { if (I->recvbuf) { hpgmg_hip_sync(); hpgmg_hip_free(I->recvbuf); } I->recvbuf = hpgmg_vector_alloc(tr + 2); I->recv_cap = tr; }
  if (P->n_sp + P->n_rp > I->ptr_cap) { I->ptr_cap = P->n_sp + P->n_rp + 8; I->ptr_tmp = (double **)realloc(I->ptr_tmp, (size_t)I->ptr_cap * sizeof(double *)); }
}

static void plan_exchange(level_type *L, halo_images *I, int which, int scr, int id, int rhs_id) {
  const hpgmg_transport *T = hpgmg_get_transport();
  image_plan *P = &I->plan[which];
  int q;
  if (!P->built) plan_build(L, I, which);
  if (P->n_send + P->n_recv == 0) return;
  TICK(L, ghostZone_total, which == IMG_PLAN_COEF ? "images: coefficient vectors" : "images: refresh");
  HIP_OK(hpgmg_hip_pair_halo_pack(&I->dev, (double *const *)I->d_pair_base, scr, id, 0, 0, rhs_id, P->d_send, P->n_send, I->sendbuf));
  double **rp = I->ptr_tmp, **sp = I->ptr_tmp + P->n_rp;
  for (q = 0; q < P->n_rp; q++) rp[q] = I->recvbuf + P->rp_off[q];
  for (q = 0; q < P->n_sp; q++) sp[q] = I->sendbuf + P->sp_off[q];
  T->sendrecv(T->ctx, P->n_rp, rp, P->rp_size, P->rp_rank, P->n_sp, sp, P->sp_size, P->sp_rank, (L->tag << 4) | 0xC);
  HIP_OK(hpgmg_hip_pair_halo_unpack(&I->dev, (double *const *)I->d_pair_base, scr, id, 0, 0, rhs_id, P->d_recv, P->n_recv, I->recvbuf, NULL, NULL));
  TOCK();
}

/* ---- boundary entries: the own boxes' blocks of the stencil's shape (sources looked up in the table with the images) and, for every image,
 * the blocks level.c:367-465 would give that box -- clipped to the cells the image holds, cut into the reference's pieces of at most 16 x 16 */
static void entries_append(hpgmg_hip_bc_entry **h, int *n, int *cap, const hpgmg_hip_bc_entry *e) {
  if (*n == *cap) { *cap = *cap ? 2 * *cap : 256; *h = (hpgmg_hip_bc_entry *)realloc(*h, (size_t)*cap * sizeof(**h)); }
  (*h)[(*n)++] = *e;
}
static hpgmg_hip_bc_entry *entries_upload(const hpgmg_hip_bc_entry *h, int n) {
  hpgmg_hip_bc_entry *d = (hpgmg_hip_bc_entry *)hpgmg_hip_malloc((size_t)(n > 0 ? n : 1) * sizeof *h);
  if (!d) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
  if (n > 0) HIP_OK(hpgmg_hip_memcpy_h2d(d, h, (size_t)n * sizeof *h));
  return d;
}
static void bc_build(level_type *L, halo_images *I) {
  const int shape = I->shape, dim = L->box_dim, g = L->box_ghosts;
  const int nblk = L->boundary_condition.num_blocks[shape];
  const blockCopy_type *blocks = L->boundary_condition.blocks[shape];
  hpgmg_hip_bc_entry *all = NULL, *konly = NULL, e;
  int n_all = 0, cap_all = 0, n_k = 0, cap_k = 0, kind, q, dir;
  for (kind = 1; kind <= 3; kind++) for (q = 0; q < nblk; q++) {             /* own boxes, faces first */
    const blockCopy_type *bl = &blocks[q];
    const int d[3] = {bl->subtype % 3 - 1, (bl->subtype % 9) / 3 - 1, bl->subtype / 9 - 1};
    const int lo[3] = {bl->read.i, bl->read.j, bl->read.k}, len[3] = {bl->dim.i, bl->dim.j, bl->dim.k};
    if ((d[0] != 0) + (d[1] != 0) + (d[2] != 0) != kind) continue;
    int p[3];
    box_position(L, I->gid[bl->read.box], p);
    const int local = hp_bc_entry_from_block(L, bl->read.box, p, lo, len, bl->subtype, table_find, I, &e);
    /* a block that runs along the face of a box of ANOTHER rank fills ghost cells no kernel of this path reads: a place outside a box is looked
     * up through the in-domain directions first (gf_column), i.e. in the image's own ghost zone -- the image's entry below */
    if (!local || e.src_box >= I->n_real) continue;
    entries_append(&all, &n_all, &cap_all, &e);
    if (d[2]) entries_append(&konly, &n_k, &cap_k, &e);
  }
  I->n_bc_own = n_all;
  for (kind = 1; kind <= 3; kind++) for (q = 0; q < I->n_img; q++) for (dir = 0; dir < 27; dir++) {
    const int t = I->n_real + q;
    const int d[3] = { dir % 3 - 1, (dir / 3) % 3 - 1, dir / 9 - 1 };
    const int nbl[3] = { L->boxes_in.i, L->boxes_in.j, L->boxes_in.k };
    int p[3], a, outside_all = 1, order = 0, need_lo[3], need_len[3], lo[3], len[3], empty = 0;
    if (dir == 13) continue;
    box_position(L, I->gid[t], p);
    for (a = 0; a < 3; a++) if (d[a]) { order++; if (!(p[a] + d[a] < 0 || p[a] + d[a] >= nbl[a])) outside_all = 0; }
    if (order != kind || !outside_all) continue;              /* only regions whose every direction leaves the DOMAIN: the others run along a neighbouring box */
    if (shape == STENCIL_SHAPE_STAR && order > 1) continue;
    if (shape == STENCIL_SHAPE_NO_CORNERS && order > 2) continue;
    if (!clip_region(L, p, I->lo, I->n, I->depth_max, 0, need_lo, need_len)) continue;
    for (a = 0; a < 3; a++) {
      if (d[a] < 0) { lo[a] = -g; len[a] = g; }
      else if (d[a] > 0) { lo[a] = dim; len[a] = g; }
      else { lo[a] = need_lo[a]; len[a] = need_len[a]; if (len[a] <= 0) empty = 1; }
    }
    if (empty) continue;
    { /* pieces of at most (whole i) x 16 x 16 (level.c:430-432: a block never tiles smaller than the ghost depth) */
      const int tj = 16 < g ? g : 16, tk = tj;
      int j0, k0;
      for (k0 = 0; k0 < len[2]; k0 += tk) for (j0 = 0; j0 < len[1]; j0 += tj) {
        const int plo[3] = { lo[0], lo[1] + j0, lo[2] + k0 };
        const int plen[3] = { len[0], (len[1] - j0 < tj) ? len[1] - j0 : tj, (len[2] - k0 < tk) ? len[2] - k0 : tk };
        hp_bc_entry_from_block(L, t, p, plo, plen, dir, table_find, I, &e);
        entries_append(&all, &n_all, &cap_all, &e);
        if (d[2]) entries_append(&konly, &n_k, &cap_k, &e);
      }
    }
  }
  I->d_bc = entries_upload(all, n_all); I->n_bc = n_all;
  I->d_bc_k = entries_upload(konly, n_k); I->n_bc_k = n_k;
  free(all); free(konly);
}

const hpgmg_hip_bc_entry *hp_images_bc_k(level_type *L, backend_t *B, int *n_out) {
  if (!B->img->d_bc) bc_build(L, B->img);
  *n_out = B->img->n_bc_k;
  return B->img->d_bc_k;
}
/* apply_BCs of `order` (4: v4, 12: p2, 2: v2) to vector (scr, id): part 0 own boxes and images, 1 own boxes, 2 images */
void hp_images_bcs(level_type *L, backend_t *B, int scr, int id, int order, int part) {
  halo_images *I = B->img;
  if (!I->d_bc) bc_build(L, I);
  const hpgmg_hip_bc_entry *e = I->d_bc + (part == 2 ? I->n_bc_own : 0);
  const int n = part == 0 ? I->n_bc : (part == 1 ? I->n_bc_own : I->n_bc - I->n_bc_own);
  if (n <= 0) return;
  hpgmg_hip_level Ls = I->dev;
  if (scr) Ls.box_base = (double *const *)I->d_pair_base;
  TICK(L, boundary_conditions, "apply_BCs (own boxes and images)");
  HIP_OK(hpgmg_hip_exchange_and_bc(&Ls, id, NULL, 0, e, n, order));
  TOCK();
}

/* fv4 red + black: the cells whose intermediate value the one-pass kernel does not recompute (operators_hip.c: fv4_special_cells) -- on a face
 * between two boxes of the table, next to a domain wall in another direction -- listed for every box whose neighbour across that face is an own
 * box (the reader); for an image only the cells it holds. */
const int *hp_images_fv4_special(level_type *L, backend_t *B, int *n_out) {
  halo_images *I = B->img;
  if (I->n_special < 0) {
    int cap = 1024, n = 0, t, ax, side, u, v;
    int *h = (int *)malloc((size_t)cap * 4 * sizeof(int));
    const int dim = L->box_dim, N[3] = { L->dim.i, L->dim.j, L->dim.k };
    for (t = 0; t < I->n_all; t++) {
      int p[3], need_lo[3] = {0, 0, 0}, need_len[3] = {dim, dim, dim};
      box_position(L, I->gid[t], p);
      if (t >= I->n_real && !clip_region(L, p, I->lo, I->n, I->depth_max, 0, need_lo, need_len)) continue;
      for (ax = 0; ax < 3; ax++) for (side = 0; side < 2; side++) {
        const int nb = I->h_nbr[6 * t + 2 * ax + side];
        if (nb < 0 || nb >= I->n_real) continue;                            /* a wall, or nobody on this rank reads across that face */
        const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
        for (v = need_lo[a2]; v < need_lo[a2] + need_len[a2]; v++) for (u = need_lo[a1]; u < need_lo[a1] + need_len[a1]; u++) {
          const int g1 = p[a1] * dim + u, g2 = p[a2] * dim + v;
          if (!(g1 == 0 || g1 == N[a1] - 1 || g2 == 0 || g2 == N[a2] - 1)) continue;
          int c[3];
          c[ax] = side ? dim - 1 : 0; c[a1] = u; c[a2] = v;
          if (n == cap) { cap *= 2; h = (int *)realloc(h, (size_t)cap * 4 * sizeof(int)); }
          h[4 * n] = t; h[4 * n + 1] = c[0]; h[4 * n + 2] = c[1]; h[4 * n + 3] = c[2]; n++;
        }
      }
    }
    I->d_special = (int *)hpgmg_hip_malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(int));
    if (!I->d_special) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
    if (n > 0) HIP_OK(hpgmg_hip_memcpy_h2d(I->d_special, h, (size_t)n * 4 * sizeof(int)));
    free(h);
    I->n_special = n;
  }
  *n_out = I->n_special;
  return I->d_special;
}

static halo_images *images_build(level_type *L, backend_t *B, int *blo, int *bn) {
  const int me = L->my_rank, dim = L->box_dim;
  const size_t vol = (size_t)L->box_volume, nv = (size_t)L->numVectors;
  halo_images *I = (halo_images *)calloc(1, sizeof(*I));
  int a, b, bi, bj, bk, t, d;
  I->brick_lo = blo; I->brick_n = bn;
  for (a = 0; a < 3; a++) { I->lo[a] = blo[3 * me + a]; I->n[a] = bn[3 * me + a]; }
  I->shape = stencil_get_shape();
  I->depth_max = stencil_get_radius() + 1;
  I->n_real = L->num_my_boxes; I->n_special = -1;
  I->gid = (int *)malloc((size_t)(L->boxes_in.i * L->boxes_in.j * L->boxes_in.k) * sizeof(int));
  for (b = 0; b < I->n_real; b++) I->gid[b] = L->my_boxes[b].global_box_id;
  t = I->n_real;
  for (bk = I->lo[2] - 1; bk <= I->lo[2] + I->n[2]; bk++) for (bj = I->lo[1] - 1; bj <= I->lo[1] + I->n[1]; bj++) for (bi = I->lo[0] - 1; bi <= I->lo[0] + I->n[0]; bi++) {
    const int r = hp_box_rank_at(L, bi, bj, bk);
    if (r < 0 || r == me) continue;
    I->gid[t++] = bi + L->boxes_in.i * (bj + L->boxes_in.j * bk);
  }
  I->n_all = t; I->n_img = t - I->n_real;
  /* storage of the images: level vectors and the two private vectors, in the alignment class of the own boxes */
  const size_t pad = ((uintptr_t)L->my_boxes[0].vectors[0] % 16) / sizeof(double);
  I->storage = (double *)hpgmg_hip_malloc(((size_t)I->n_img * nv * vol + 2) * sizeof(double));
  I->scratch = (double *)hpgmg_hip_malloc(((size_t)I->n_img * 2 * vol + 2) * sizeof(double));
  if (!I->storage || !I->scratch) { fprintf(stderr, "hpgmg: no memory for %d images of neighbouring boxes: %s\n", I->n_img, hpgmg_hip_last_error()); abort(); }
  hp_ensure_pair_scratch(L, B);
  double **base = (double **)calloc((size_t)I->n_all, sizeof(double *)), **pbase = (double **)calloc((size_t)I->n_all, sizeof(double *));
  int *low = (int *)calloc((size_t)I->n_all * 3, sizeof(int));
  I->h_nbr = (int *)calloc((size_t)I->n_all * 6, sizeof(int));
  for (t = 0; t < I->n_all; t++) {
    int p[3];
    box_position(L, I->gid[t], p);
    if (t < I->n_real) { base[t] = L->my_boxes[t].vectors[0]; pbase[t] = B->pair_scratch + pad + (size_t)t * 2 * vol; }
    else { const size_t q = (size_t)(t - I->n_real); base[t] = I->storage + pad + q * nv * vol; pbase[t] = I->scratch + pad + q * 2 * vol; }
    for (a = 0; a < 3; a++) low[3 * t + a] = p[a] * dim;
    for (d = 0; d < 6; d++) {
      int np[3] = { p[0], p[1], p[2] };
      np[d / 2] += (d & 1) ? 1 : -1;
      const int r = hp_box_rank_at(L, np[0], np[1], np[2]);
      int code = -1;                                             /* the domain boundary */
      if (r >= 0) { code = table_find(I, np[0] + L->boxes_in.i * (np[1] + L->boxes_in.j * np[2])); if (code < 0) code = -2; }
      I->h_nbr[6 * t + d] = code;
    }
  }
  I->d_box_base = (double **)hpgmg_hip_malloc((size_t)I->n_all * sizeof(double *));
  I->d_pair_base = (double **)hpgmg_hip_malloc((size_t)I->n_all * sizeof(double *));
  I->d_box_low = (int *)hpgmg_hip_malloc((size_t)I->n_all * 3 * sizeof(int));
  I->d_box_nbr = (int *)hpgmg_hip_malloc((size_t)I->n_all * 6 * sizeof(int));
  if (!I->d_box_base || !I->d_pair_base || !I->d_box_low || !I->d_box_nbr) { fprintf(stderr, "hpgmg: device allocation failed: %s\n", hpgmg_hip_last_error()); abort(); }
  HIP_OK(hpgmg_hip_memcpy_h2d(I->d_box_base, base, (size_t)I->n_all * sizeof(double *)));
  HIP_OK(hpgmg_hip_memcpy_h2d(I->d_pair_base, pbase, (size_t)I->n_all * sizeof(double *)));
  HIP_OK(hpgmg_hip_memcpy_h2d(I->d_box_low, low, (size_t)I->n_all * 3 * sizeof(int)));
  HIP_OK(hpgmg_hip_memcpy_h2d(I->d_box_nbr, I->h_nbr, (size_t)I->n_all * 6 * sizeof(int)));
  free(base); free(pbase); free(low);
  I->dev = B->dev;
  I->dev.box_base = (double *const *)I->d_box_base; I->dev.box_low = I->d_box_low; I->dev.box_nbr = I->d_box_nbr;
  I->dev_all = I->dev; I->dev_all.num_boxes = I->n_all; I->dev_all.box_stride = 0;
  I->seen_v0 = B->seen_v0; I->seen_nv = B->seen_nv;
  return I;
}

void hp_images_release(backend_t *B) {
  halo_images *I = B->img;
  int q;
  if (!I) { B->img_state = 0; return; }
  for (q = 0; q < IMG_PLANS; q++) {
    image_plan *P = &I->plan[q];
    if (P->d_send) hpgmg_hip_free(P->d_send);
    if (P->d_recv) hpgmg_hip_free(P->d_recv);
    free(P->sp_rank); free(P->rp_rank); free(P->sp_size); free(P->rp_size); free(P->sp_off); free(P->rp_off);
  }
  if (I->sendbuf) hpgmg_hip_free(I->sendbuf);
  if (I->recvbuf) hpgmg_hip_free(I->recvbuf);
  if (I->storage) hpgmg_hip_free(I->storage);
  if (I->scratch) hpgmg_hip_free(I->scratch);
  if (I->d_box_base) hpgmg_hip_free(I->d_box_base);
  if (I->d_pair_base) hpgmg_hip_free(I->d_pair_base);
  if (I->d_box_low) hpgmg_hip_free(I->d_box_low);
  if (I->d_box_nbr) hpgmg_hip_free(I->d_box_nbr);
  if (I->d_bc) hpgmg_hip_free(I->d_bc);
  if (I->d_bc_k) hpgmg_hip_free(I->d_bc_k);
  if (I->d_special) hpgmg_hip_free(I->d_special);
  free(I->gid); free(I->h_nbr); free(I->brick_lo); free(I->brick_n); free(I->ptr_tmp);
  free(I);
  B->img = NULL; B->img_state = 0; B->img_active = 0;
}
void hp_images_invalidate_coefficients(backend_t *B) { if (B->img) B->img->coef_valid = 0; }

/* May the 27-point / fv4 stencil launches of this level read the neighbouring ranks' cells from images?  Every rank reaches the same
 * answer (it decides the message pattern): Dirichlet, every rank's share a brick, boxes large enough that an image only ever stands for a
 * direct neighbour. */
int hp_images_ready(level_type *L, backend_t *B) {
  if (B->img && (B->img->seen_v0 != B->seen_v0 || B->img->seen_nv != B->seen_nv)) hp_images_release(B);       /* create_vectors() re-allocated the boxes */
  if (B->img_state == 0) {
    const hpgmg_transport *T = hpgmg_get_transport();
    hpgmg_config cfg;
    hpgmg_get_config(&cfg);
    B->img_state = -1;
    int ok = hp_switch(SW_IMAGES) && T && T->size > 1 && hp_ghost_free_mode() && (cfg.op == HPGMG_OP_27PT || cfg.op == HPGMG_OP_FV4) &&
             L->boundary_condition.type == BC_DIRICHLET && L->num_my_boxes > 0 && !B->all_faces_local &&
             L->box_dim >= 8 && stencil_get_radius() + 1 + L->box_ghosts <= L->box_dim;
    if (ok) {
      int *blo = (int *)malloc((size_t)T->size * 3 * sizeof(int)), *bn = (int *)malloc((size_t)T->size * 3 * sizeof(int));
      if (all_bricks(L, T->size, blo, bn) && bn[3 * L->my_rank] > 0) { B->img = images_build(L, B, blo, bn); B->img_state = 1; }
      else { free(blo); free(bn); }
    }
  }
  return B->img_state > 0;
}

/* Make the images of vector (scr, id) current `depth` cells deep (and those of the right-hand side one cell deep when rhs_id >= 0), then
 * apply the boundary conditions of bc_order to the own boxes and the images: what exchange_boundary() + apply_BCs() are to a launch that
 * reads ghost zones (gsrb.c:30-33, chebyshev.c:45-46, residual.c:11-12).  The launch that follows must use hp_stencil_dev(B). */
void hp_images_refresh(level_type *L, backend_t *B, int scr, int id, int depth, int rhs_id, int bc_order) {
  halo_images *I = B->img;
  if (depth < 1 || depth > I->depth_max) { fprintf(stderr, "hpgmg: images hold %d cells, %d asked for\n", I->depth_max, depth); abort(); }
  if (!I->coef_valid) { plan_exchange(L, I, IMG_PLAN_COEF, 0, 0, 0); I->coef_valid = 1; }
  plan_exchange(L, I, depth - 1 + (rhs_id >= 0 ? 3 : 0), scr, id, rhs_id >= 0 ? rhs_id : 0);
  hp_images_bcs(L, B, scr, id, bc_order, 0);
  B->img_active = 1;
  hp_images_exchanges++;
}

/* The same with the message hidden behind computation (north_star: "ghost-zone exchange on RCCL over xGMI overlapped with interior smoothing";
 * the reference overlaps only its local copies with the messages, exchange_boundary.c:81-90):
 *     launch stream:    own boxes' boundary conditions | part 1 of the stencil launch: tiles that read no image | wait | part 2
 *     exchange stream:  wait for the vector | pack, grouped send / receive, unpack, the images' boundary conditions |
 * Returns 1 when set up that way -- the caller issues part 1, hp_images_refresh_end(), part 2 (hpgmg_hip_set_tile_part) -- and 0 when the
 * refresh was done in line (HPGMG_OVERLAP=0): one whole launch. */
static void *img_stream = NULL, *ev_ready = NULL, *ev_landed = NULL;
int hp_images_refresh_begin(level_type *L, backend_t *B, int scr, int id, int depth, int rhs_id, int bc_order) {
  halo_images *I = B->img;
  if (!hp_overlap_enabled() || hpgmg_get_timer_mode() == TIMERS_SYNC) { hp_images_refresh(L, B, scr, id, depth, rhs_id, bc_order); return 0; }
  if (depth < 1 || depth > I->depth_max) { fprintf(stderr, "hpgmg: images hold %d cells, %d asked for\n", I->depth_max, depth); abort(); }
  if (!img_stream) {
    img_stream = hpgmg_hip_stream_create(); ev_ready = hpgmg_hip_event_create(); ev_landed = hpgmg_hip_event_create();
    if (!img_stream || !ev_ready || !ev_landed) { fprintf(stderr, "hpgmg: cannot create the exchange stream\n"); abort(); }
  }
  void *launch_stream = hpgmg_hip_get_stream();
  HIP_OK(hpgmg_hip_event_record(ev_ready));                    /* the vector to be sent is complete once everything issued so far has run */
  hp_images_bcs(L, B, scr, id, bc_order, 1);
  hpgmg_hip_set_stream(img_stream);
  HIP_OK(hpgmg_hip_stream_wait_event(ev_ready));
  if (!I->coef_valid) { plan_exchange(L, I, IMG_PLAN_COEF, 0, 0, 0); I->coef_valid = 1; }
  plan_exchange(L, I, depth - 1 + (rhs_id >= 0 ? 3 : 0), scr, id, rhs_id >= 0 ? rhs_id : 0);
  hp_images_bcs(L, B, scr, id, bc_order, 2);
  HIP_OK(hpgmg_hip_event_record(ev_landed));
  hpgmg_hip_set_stream(launch_stream);
  B->img_active = 1;
  hp_images_exchanges++;
  hp_overlap_counted();
  return 1;
}
void hp_images_refresh_end(void) { HIP_OK(hpgmg_hip_stream_wait_event(ev_landed)); }

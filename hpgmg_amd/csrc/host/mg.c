/*
 * mg.c -- multigrid hierarchy construction and the F-cycle / V-cycle drivers.
 *
 * Behavioural reference: finite-volume/source/mg.c
 *   build_interpolation :181-475   build_restriction :484-831   MGBuild :842-1022
 *   richardson_error :1113-1131   MGVCycle :1135-1164   MGSolve :1168-1233
 *   FMGSolve :1237-1344   MGPrintTiming :54-161
 * Only operators.h functions touch vector data; this file passes vector ids.
 * The ORDER of operator calls in MGVCycle/FMGSolve is kept exactly, since that
 * order is what the reference's printed residual norms pin down.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <stdint.h>
#include "hpgmg_level.h"
#include "hpgmg_operators.h"
#include "hpgmg_mg.h"

hpgmg_solve_record hpgmg_last_solve;
int hpgmg_verbose = 1;

void hpgmg_communicator_free(communicator_type *C); /* level.c */

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
#define SAY(rank, ...) do { if ((rank) == 0 && hpgmg_verbose) { fprintf(stdout, __VA_ARGS__); fflush(stdout); } } while (0)

/* ------------------------------------------------------------------ inter-level lists */
typedef struct { int send_rank, send_id, send_box, recv_rank, recv_id, recv_box, i, j, k; } xfer_t;

static int xfer_cmp(const void *pa, const void *pb) {
  const xfer_t *a = (const xfer_t *)pa, *b = (const xfer_t *)pb;
  if (a->send_rank != b->send_rank) return a->send_rank < b->send_rank ? -1 : 1;
  if (a->send_id   != b->send_id)   return a->send_id   < b->send_id   ? -1 : 1;
  if (a->recv_id   != b->recv_id)   return a->recv_id   < b->recv_id   ? -1 : 1; /* makes ties explicit */
  return 0;
}
static int int_cmp(const void *pa, const void *pb) { int a = *(const int *)pa, b = *(const int *)pb; return (a > b) - (a < b); }
static int sort_unique(int *v, int n) {
  int i, m = 0;
  qsort(v, (size_t)n, sizeof(int), int_cmp);
  for (i = 0; i < n; i++) if (m == 0 || v[m - 1] != v[i]) v[m++] = v[i];
  return m;
}
static int local_index_of(const level_type *L, int global_id) {
  int b;
  for (b = 0; b < L->num_my_boxes; b++) if (L->my_boxes[b].global_box_id == global_id) return b;
  return -1;
}
static void box_coords(const level_type *L, const box_type *B, int c[3]) {
  c[0] = B->low.i / L->box_dim; c[1] = B->low.j / L->box_dim; c[2] = B->low.k / L->box_dim;
}
static int box_id(const level_type *L, int bi, int bj, int bk) { return bi + L->boxes_in.i * (bj + L->boxes_in.j * bk); }

/* Enumerate every (fine box, coarse box) pair in which this rank owns at least
 * one end.  own_fine: pairs generated from my fine boxes; else from my coarse boxes. */
static int enumerate_pairs(const level_type *F, const level_type *C, int from_fine, xfer_t **out) {
  const int ri = F->boxes_in.i / C->boxes_in.i, rj = F->boxes_in.j / C->boxes_in.j, rk = F->boxes_in.k / C->boxes_in.k;
  const int half = F->box_dim / 2;
  int n = 0, b;
  if (from_fine) {
    xfer_t *v = (xfer_t *)malloc((size_t)(F->num_my_boxes + 1) * sizeof(xfer_t));
    for (b = 0; b < F->num_my_boxes; b++) {
      int fc[3]; box_coords(F, &F->my_boxes[b], fc);
      int cid = box_id(C, fc[0] / ri, fc[1] / rj, fc[2] / rk);
      xfer_t t = { F->rank_of_box[F->my_boxes[b].global_box_id], F->my_boxes[b].global_box_id, b,
                   C->rank_of_box[cid], cid, local_index_of(C, cid),
                   half * (fc[0] % ri), half * (fc[1] % rj), half * (fc[2] % rk) };
      v[n++] = t;
    }
    *out = v;
  } else {
    xfer_t *v = (xfer_t *)malloc((size_t)(C->num_my_boxes * ri * rj * rk + 1) * sizeof(xfer_t));
    int bi, bj, bk;
    for (b = 0; b < C->num_my_boxes; b++) {
      int cc[3]; box_coords(C, &C->my_boxes[b], cc);
      for (bk = 0; bk < rk; bk++) for (bj = 0; bj < rj; bj++) for (bi = 0; bi < ri; bi++) {
        int fid = box_id(F, ri * cc[0] + bi, rj * cc[1] + bj, rk * cc[2] + bk);
        xfer_t t = { F->rank_of_box[fid], fid, local_index_of(F, fid),
                     C->rank_of_box[C->my_boxes[b].global_box_id], C->my_boxes[b].global_box_id, b,
                     half * bi, half * bj, half * bk };
        v[n++] = t;
      }
    }
    *out = v;
  }
  return n;
}

static void alloc_peers(int n, int **ranks, int **sizes, double ***bufs, const int *peers) {
  *ranks = (int *)malloc((size_t)n * sizeof(int));
  *sizes = (int *)calloc((size_t)n, sizeof(int));
  *bufs  = (double **)calloc((size_t)n, sizeof(double *));
  memcpy(*ranks, peers, (size_t)n * sizeof(int));
}

/* restriction fine level l -> coarse level l+1.  Send side (pack [0] + local [1])
 * hangs off the FINE level, receive side (unpack [2]) off the COARSE level.
 * Inside one message regions are ordered by fine global box id. */
static void build_restriction(mg_type *G, int type) {
  int l, n, p;
  for (l = 0; l < G->num_levels; l++) memset(&G->levels[l]->restriction[type], 0, sizeof(communicator_type));
  for (l = 0; l + 1 < G->num_levels; l++) {
    level_type *F = G->levels[l], *C = G->levels[l + 1];
    const int me = F->my_rank, half = F->box_dim / 2;
    const int ni = half + (type == RESTRICT_FACE_I), nj = half + (type == RESTRICT_FACE_J), nk = half + (type == RESTRICT_FACE_K);
    const int elem = ni * nj * nk;
    /* ---- fine side ---- */
    if (F->num_my_boxes > 0) {
      communicator_type *S = &F->restriction[type];
      xfer_t *x; int nx = enumerate_pairs(F, C, 1, &x);
      int *peers = (int *)malloc((size_t)(nx + 1) * sizeof(int)), npeers = 0, nremote = 0;
      qsort(x, (size_t)nx, sizeof(xfer_t), xfer_cmp);
      for (n = 0; n < nx; n++) if (x[n].recv_rank != me) { peers[npeers++] = x[n].recv_rank; nremote++; }
      npeers = sort_unique(peers, npeers);
      S->num_sends = npeers;
      if (npeers) alloc_peers(npeers, &S->send_ranks, &S->send_sizes, &S->send_buffers, peers);
      double *bulk = nremote ? hpgmg_vector_alloc((size_t)nremote * (size_t)elem) : NULL;
      for (p = 0; p < npeers; p++) {
        S->send_buffers[p] = bulk;
        for (n = 0; n < nx; n++) if (x[n].recv_rank == peers[p]) {
          const box_type *B = &F->my_boxes[x[n].send_box];
          append_block_to_list(&S->blocks[0], &S->allocated_blocks[0], &S->num_blocks[0], ni, nj, nk,
                               x[n].send_box, NULL, 0, 0, 0, B->jStride, B->kStride, 2,
                               -1, S->send_buffers[p], S->send_sizes[p], 0, 0, ni, ni * nj, 1,
                               BLOCKCOPY_TILE_I, BLOCKCOPY_TILE_J, BLOCKCOPY_TILE_K, 0);
          S->send_sizes[p] += elem;
        }
        bulk += S->send_sizes[p];
      }
      for (n = 0; n < nx; n++) if (x[n].recv_rank == me) {
        const box_type *B = &F->my_boxes[x[n].send_box], *D = &C->my_boxes[x[n].recv_box];
        append_block_to_list(&S->blocks[1], &S->allocated_blocks[1], &S->num_blocks[1], ni, nj, nk,
                             x[n].send_box, NULL, 0, 0, 0, B->jStride, B->kStride, 2,
                             x[n].recv_box, NULL, x[n].i, x[n].j, x[n].k, D->jStride, D->kStride, 1,
                             BLOCKCOPY_TILE_I, BLOCKCOPY_TILE_J, BLOCKCOPY_TILE_K, 0);
      }
      free(x); free(peers);
    }
    /* ---- coarse side ---- */
    if (C->num_my_boxes > 0) {
      communicator_type *R = &C->restriction[type];
      xfer_t *x; int nx = enumerate_pairs(F, C, 0, &x);
      int *peers = (int *)malloc((size_t)(nx + 1) * sizeof(int)), npeers = 0, nremote = 0;
      qsort(x, (size_t)nx, sizeof(xfer_t), xfer_cmp);
      for (n = 0; n < nx; n++) if (x[n].send_rank != C->my_rank) { peers[npeers++] = x[n].send_rank; nremote++; }
      npeers = sort_unique(peers, npeers);
      R->num_recvs = npeers;
      if (npeers) alloc_peers(npeers, &R->recv_ranks, &R->recv_sizes, &R->recv_buffers, peers);
      double *bulk = nremote ? hpgmg_vector_alloc((size_t)nremote * (size_t)elem) : NULL;
      for (p = 0; p < npeers; p++) {
        R->recv_buffers[p] = bulk;
        for (n = 0; n < nx; n++) if (x[n].send_rank == peers[p]) {
          const box_type *D = &C->my_boxes[x[n].recv_box];
          append_block_to_list(&R->blocks[2], &R->allocated_blocks[2], &R->num_blocks[2], ni, nj, nk,
                               -1, R->recv_buffers[p], R->recv_sizes[p], 0, 0, ni, ni * nj, 1,
                               x[n].recv_box, NULL, x[n].i, x[n].j, x[n].k, D->jStride, D->kStride, 1,
                               BLOCKCOPY_TILE_I, BLOCKCOPY_TILE_J, BLOCKCOPY_TILE_K, 0);
          R->recv_sizes[p] += elem;
        }
        bulk += R->recv_sizes[p];
      }
      free(x); free(peers);
    }
  }
}

/* interpolation coarse level l+1 -> fine level l.  Send side (pack [0] + local
 * [1]) hangs off the COARSE level, receive side (unpack/increment [2]) off the
 * FINE level.  A message carries whole interpolated fine boxes, ordered by
 * coarse global id then fine global id. */
static void build_interpolation(mg_type *G) {
  int l, n, p;
  for (l = 0; l < G->num_levels; l++) memset(&G->levels[l]->interpolation, 0, sizeof(communicator_type));
  for (l = 0; l + 1 < G->num_levels; l++) {
    level_type *F = G->levels[l], *C = G->levels[l + 1];
    const int half = F->box_dim / 2, fd = F->box_dim, elem = fd * fd * fd;
    /* ---- coarse side ---- */
    if (C->num_my_boxes > 0) {
      communicator_type *S = &C->interpolation;
      xfer_t *x; int nx = enumerate_pairs(F, C, 0, &x);
      /* here the coarse box is the sender: swap roles so the sort key is (coarse owner, coarse id, fine id) */
      for (n = 0; n < nx; n++) { xfer_t t = x[n];
        x[n].send_rank = t.recv_rank; x[n].send_id = t.recv_id; x[n].send_box = t.recv_box;
        x[n].recv_rank = t.send_rank; x[n].recv_id = t.send_id; x[n].recv_box = t.send_box; }
      int *peers = (int *)malloc((size_t)(nx + 1) * sizeof(int)), npeers = 0, nremote = 0;
      qsort(x, (size_t)nx, sizeof(xfer_t), xfer_cmp);
      for (n = 0; n < nx; n++) if (x[n].recv_rank != C->my_rank) { peers[npeers++] = x[n].recv_rank; nremote++; }
      npeers = sort_unique(peers, npeers);
      S->num_sends = npeers;
      if (npeers) alloc_peers(npeers, &S->send_ranks, &S->send_sizes, &S->send_buffers, peers);
      double *bulk = nremote ? hpgmg_vector_alloc((size_t)nremote * (size_t)elem) : NULL;
      for (p = 0; p < npeers; p++) {
        S->send_buffers[p] = bulk;
        for (n = 0; n < nx; n++) if (x[n].recv_rank == peers[p]) {
          const box_type *B = &C->my_boxes[x[n].send_box];
          append_block_to_list(&S->blocks[0], &S->allocated_blocks[0], &S->num_blocks[0], half, half, half,
                               x[n].send_box, NULL, x[n].i, x[n].j, x[n].k, B->jStride, B->kStride, 1,
                               -1, S->send_buffers[p], S->send_sizes[p], 0, 0, fd, fd * fd, 2,
                               BLOCKCOPY_TILE_I, BLOCKCOPY_TILE_J, BLOCKCOPY_TILE_K, 0);
          S->send_sizes[p] += elem;
        }
        bulk += S->send_sizes[p];
      }
      for (n = 0; n < nx; n++) if (x[n].recv_rank == C->my_rank) {
        const box_type *B = &C->my_boxes[x[n].send_box], *D = &F->my_boxes[x[n].recv_box];
        append_block_to_list(&S->blocks[1], &S->allocated_blocks[1], &S->num_blocks[1], half, half, half,
                             x[n].send_box, NULL, x[n].i, x[n].j, x[n].k, B->jStride, B->kStride, 1,
                             x[n].recv_box, NULL, 0, 0, 0, D->jStride, D->kStride, 2,
                             BLOCKCOPY_TILE_I, BLOCKCOPY_TILE_J, BLOCKCOPY_TILE_K, 0);
      }
      free(x); free(peers);
    }
    /* ---- fine side ---- */
    if (F->num_my_boxes > 0) {
      communicator_type *R = &F->interpolation;
      xfer_t *x; int nx = enumerate_pairs(F, C, 1, &x);
      for (n = 0; n < nx; n++) { xfer_t t = x[n];
        x[n].send_rank = t.recv_rank; x[n].send_id = t.recv_id; x[n].send_box = t.recv_box;
        x[n].recv_rank = t.send_rank; x[n].recv_id = t.send_id; x[n].recv_box = t.send_box; }
      int *peers = (int *)malloc((size_t)(nx + 1) * sizeof(int)), npeers = 0, nremote = 0;
      qsort(x, (size_t)nx, sizeof(xfer_t), xfer_cmp);
      for (n = 0; n < nx; n++) if (x[n].send_rank != F->my_rank) { peers[npeers++] = x[n].send_rank; nremote++; }
      npeers = sort_unique(peers, npeers);
      R->num_recvs = npeers;
      if (npeers) alloc_peers(npeers, &R->recv_ranks, &R->recv_sizes, &R->recv_buffers, peers);
      double *bulk = nremote ? hpgmg_vector_alloc((size_t)nremote * (size_t)elem) : NULL;
      for (p = 0; p < npeers; p++) {
        R->recv_buffers[p] = bulk;
        for (n = 0; n < nx; n++) if (x[n].send_rank == peers[p]) {
          const box_type *D = &F->my_boxes[x[n].recv_box];
          append_block_to_list(&R->blocks[2], &R->allocated_blocks[2], &R->num_blocks[2], fd, fd, fd,
                               -1, R->recv_buffers[p], R->recv_sizes[p], 0, 0, fd, fd * fd, 1,
                               x[n].recv_box, NULL, 0, 0, 0, D->jStride, D->kStride, 1,
                               BLOCKCOPY_TILE_I, BLOCKCOPY_TILE_J, BLOCKCOPY_TILE_K, 0);
          R->recv_sizes[p] += elem;
        }
        bulk += R->recv_sizes[p];
      }
      free(x); free(peers);
    }
  }
}

/* ------------------------------------------------------------------ hierarchy */
typedef struct { int procs, dim, boxes, box_dim, ghosts; } plan_t;

/* Coarsening ladder of the reference's true-V-cycle build (mg.c:895-952):
 * halve boxes while they are bigger than 8^3, then merge 8 boxes into 1, then
 * (non power-of-two domains) collapse onto few ranks, finally halve again. */
static int plan_next_reference(const plan_t *f, int coarsest_dim, plan_t *c);
/* MI355X deviation from the reference ladder (rank map only; results do not depend on which rank owns a box):
 * a level of <= hpgmg_gather_dim^3 cells is owned entirely by rank 0.  The reference keeps such levels spread over
 * all ranks until its boxes reach 8^3 (e.g. 8 ranks x 8 boxes of 8^3 for a 32^3 level), where every smoother sweep
 * is a latency-bound message exchange; one GPU runs the whole <= 64^3 tail of a V-cycle in well under a
 * millisecond with no messages (hipGraph segments, fused tail kernel).  0 = the reference's rank map. */
int hpgmg_gather_dim = -1;
void hpgmg_set_gather_dim(int dim) { hpgmg_gather_dim = dim; }
/* The reference's -DUSE_UCYCLES ladder (mg.c:878-893): boxes are halved as long as they can be and never merged, so the bottom level keeps
 * every box (a truncated V-cycle; the bottom solver then works on a level of many small boxes). */
static int hpgmg_ucycles = 0;
void hpgmg_set_ucycles(int on) { hpgmg_ucycles = on ? 1 : 0; }
static int plan_next(const plan_t *f, int coarsest_dim, plan_t *c) {
  if (hpgmg_ucycles) {
    if (f->box_dim % 2) return 0;
    *c = *f; c->dim = f->dim / 2; c->box_dim = f->box_dim / 2;
    return c->box_dim >= c->ghosts;
  }
  if (hpgmg_gather_dim < 0) { const char *e = getenv("HPGMG_GATHER_DIM"); hpgmg_gather_dim = (e && *e) ? atoi(e) : 64; }
  if (!plan_next_reference(f, coarsest_dim, c)) return 0;
  if (f->procs == 1) c->procs = 1;                       /* once gathered, stay gathered */
  if (hpgmg_gather_dim > 0 && c->dim <= hpgmg_gather_dim) c->procs = 1;
  return 1;
}
static int plan_next_reference(const plan_t *f, int coarsest_dim, plan_t *c) {
  const int r = stencil_get_radius();
  *c = *f;
  c->dim = f->dim / 2;
  if ((f->box_dim % 2 == 0) && (f->box_dim > MG_AGGLOMERATION_START) && (f->box_dim / 2 >= r)) { c->box_dim = f->box_dim / 2; return 1; }
  if ((f->boxes % 2 == 0) && (f->box_dim >= r)) { c->boxes = f->boxes / 2; return 1; }
  if (coarsest_dim != 1 && f->dim == 2 * coarsest_dim && f->dim / 2 >= r) { c->procs = 1; c->box_dim = f->dim / 2; c->boxes = 1; return 1; }
  if (coarsest_dim != 1 && f->dim == 4 * coarsest_dim && f->box_dim / 2 >= r) {
    c->procs = coarsest_dim < f->procs ? coarsest_dim : f->procs; c->box_dim = f->box_dim / 2; return 1; }
  if (coarsest_dim != 1 && f->dim == 8 * coarsest_dim && f->box_dim / 2 >= r) {
    c->procs = coarsest_dim * coarsest_dim < f->procs ? coarsest_dim * coarsest_dim : f->procs; c->box_dim = f->box_dim / 2; return 1; }
  if ((f->box_dim % 2 == 0) && (f->box_dim / 2 >= r)) { c->box_dim = f->box_dim / 2; return 1; }
  return 0;
}

void MGBuild(mg_type *G, level_type *fine, double a, double b, int minCoarseGridDim) {
  const double t0 = now();
  plan_t plan[64];
  int l, depth = 1, coarsest = fine->dim.i;
  G->my_rank = fine->my_rank;
  G->timers.MGBuild = 0; G->timers.MGSolve = 0; G->MGSolves_performed = 0;
  while (coarsest >= 2 * minCoarseGridDim && (coarsest & 1) == 0) { depth++; coarsest /= 2; }
  if (depth > 64) depth = 64;

  plan[0].procs = fine->num_ranks; plan[0].dim = fine->dim.i; plan[0].boxes = fine->boxes_in.i;
  plan[0].box_dim = fine->box_dim; plan[0].ghosts = fine->box_ghosts;
  G->levels = (level_type **)calloc((size_t)depth, sizeof(level_type *));
  G->levels[0] = fine;
  G->num_levels = 1;
  while (G->num_levels < depth) {
    plan_t next;
    if (!plan_next(&plan[G->num_levels - 1], coarsest, &next)) break;
    if (next.dim < minCoarseGridDim) break;
    plan[G->num_levels++] = next;
  }

  for (l = 1; l < G->num_levels; l++) {
    int nv = G->levels[l - 1]->numVectors;
    if (l == G->num_levels - 1) nv += IterativeSolver_NumVectors(); /* Krylov work vectors live on the bottom level */
    G->levels[l] = (level_type *)malloc(sizeof(level_type));
    create_level(G->levels[l], plan[l].boxes, plan[l].box_dim, plan[l].ghosts, nv,
                 G->levels[l - 1]->boundary_condition.type, G->levels[l - 1]->my_rank, plan[l].procs);
    G->levels[l]->h = 2.0 * G->levels[l - 1]->h;
  }

  SAY(G->my_rank, "\n  Building restriction and interpolation lists... ");
  build_restriction(G, RESTRICT_CELL);
  build_restriction(G, RESTRICT_FACE_I);
  build_restriction(G, RESTRICT_FACE_J);
  build_restriction(G, RESTRICT_FACE_K);
  build_interpolation(G);
  SAY(G->my_rank, "done\n");

  /* a rank is active on level l if it owns a box there or on any coarser level
   * (the reference's -DUSE_SUBCOMM rule, mg.c:985-989); reductions on level l
   * involve exactly those ranks */
  for (l = 1; l < G->num_levels; l++) {
    const hpgmg_transport *T = hpgmg_get_transport();
    int ll, r, nranks = (T ? T->size : 1), n = 0;
    hpgmg_level_ext *X = hpgmg_level_ext_get(G->levels[l]);
    int *act = (int *)calloc((size_t)nranks, sizeof(int));
    for (ll = l; ll < G->num_levels; ll++) {
      const level_type *L = G->levels[ll];
      int nb = L->boxes_in.i * L->boxes_in.j * L->boxes_in.k, q;
      for (q = 0; q < nb; q++) if (L->rank_of_box[q] >= 0 && L->rank_of_box[q] < nranks) act[L->rank_of_box[q]] = 1;
    }
    G->levels[l]->active = (G->my_rank < nranks) ? act[G->my_rank] : 0;
    free(X->active_ranks);
    X->active_ranks = (int *)malloc((size_t)nranks * sizeof(int));
    for (r = 0; r < nranks; r++) if (act[r]) X->active_ranks[n++] = r;
    X->num_active_ranks = n;
    free(act);
  }

  { /* sub-communicators for the levels that reduce over a proper subset of the ranks: every rank passes here with the same lists in the same order */
    const hpgmg_transport *T = hpgmg_get_transport();
    if (T && T->size > 1 && T->prepare_subset)
      for (l = 0; l < G->num_levels; l++) {
        hpgmg_level_ext *X = hpgmg_level_ext_get(G->levels[l]);
        if (X->num_active_ranks > 1 && X->num_active_ranks < T->size) T->prepare_subset(T->ctx, X->active_ranks, X->num_active_ranks);
      }
  }
  SAY(G->my_rank, "\n");
  for (l = 1; l < G->num_levels; l++) rebuild_operator(G->levels[l], G->levels[l - 1], a, b);
  SAY(G->my_rank, "\n");

  for (l = 0; l < G->num_levels; l++) {
    level_type *L = G->levels[l];
    int alpha_is_zero = 1;
    L->must_subtract_mean = 0;
    if (hpgmg_vectors_reserved() > VECTOR_ALPHA && L->active) alpha_is_zero = (dot(L, VECTOR_ALPHA, VECTOR_ALPHA) == 0.0);
    if (L->boundary_condition.type == BC_PERIODIC && (a == 0 || alpha_is_zero)) L->must_subtract_mean = 1;
  }
  G->timers.MGBuild += now() - t0;
}

void MGDestroy(mg_type *G) {
  int l, t;
  SAY(G->my_rank, "attempting to free the restriction and interpolation lists... ");
  for (l = G->num_levels - 1; l >= 0; l--) {
    hpgmg_communicator_free(&G->levels[l]->interpolation);
    for (t = 3; t >= 0; t--) hpgmg_communicator_free(&G->levels[l]->restriction[t]);
  }
  SAY(G->my_rank, "done\n");
  for (l = G->num_levels - 1; l > 0; l--) { destroy_level(G->levels[l]); free(G->levels[l]); }
  free(G->levels);
  G->levels = NULL; G->num_levels = 0;
}

void MGResetTimers(mg_type *G) {
  int l;
  for (l = 0; l < G->num_levels; l++) { hpgmg_level_sync_counters(G->levels[l]); reset_level_timers(G->levels[l]); }
  G->timers.MGSolve = 0;
  G->MGSolves_performed = 0;
}

/* ------------------------------------------------------------------ launch-bound segments
 * Work on levels of <= SEGMENT_MAX_DIM^3 cells is a long chain of tiny launches.  The driver
 * names each stretch of such work between two host synchronisations (bottom solves, norms) with
 * a key that is identical in every solve; the plugin may capture and replay it (HIP: hipGraph).
 * Nothing here changes WHAT is executed or in which order. */
#define SEGMENT_MAX_DIM 64
static long long seg_base = 0;   /* identifies (hierarchy, start level) of the running solve */
static int seg_counter = 0, seg_open_now = 0;
static int is_small(const mg_type *G, int l) { return G->levels[l]->dim.i <= SEGMENT_MAX_DIM; }
/* A captured segment bakes in the vector ids, a, b and (through the Chebyshev coefficients) every level's eigenvalue bound: all of them
 * are part of the key, so another solve on the same hierarchy with other vectors or a rebuilt operator never replays a stale graph. */
static void seg_reset(const mg_type *G, int onLevel, int u_id, int F_id, double a, double b) {
  unsigned long long h = 1469598103934665603ULL;
  int l;
  /* every value goes through a full 64-bit avalanche (splitmix64 finaliser): small integers as doubles differ only in their top 12 bits,
   * which a multiplicative hash never carries downwards -- and the top bits are the ones the key drops below */
#define MIX(v) do { unsigned long long t_; double d_ = (double)(v); memcpy(&t_, &d_, sizeof t_); h ^= t_; \
    h ^= h >> 30; h *= 0xbf58476d1ce4e5b9ULL; h ^= h >> 27; h *= 0x94d049bb133111ebULL; h ^= h >> 31; } while (0)
  MIX(u_id); MIX(F_id); MIX(a); MIX(b); MIX(onLevel);
  for (l = 0; l < G->num_levels; l++) MIX(G->levels[l]->dominant_eigenvalue_of_DinvA);
#undef MIX
  seg_base = (long long)(((((unsigned long long)(uintptr_t)G) << 20) ^ (h << 12)) & ~0xfffULL);   /* low 12 bits: the segment counter */
  seg_counter = 0; seg_open_now = 0;
}
static void seg_open(void) { if (!seg_open_now) { hpgmg_segment_begin(seg_base + (seg_counter++)); seg_open_now = 1; } }
static void seg_close(void) { if (seg_open_now) { hpgmg_segment_end(); seg_open_now = 0; } }

/* ------------------------------------------------------------------ cycles */
void richardson_error(mg_type *G, int lh, int u_id) {
  /* || u^2h - R u^h ||_inf estimates the error at h; the ratio of two such
   * differences estimates the order (reference mg.c:1113-1131) */
  restriction(G->levels[lh + 1], VECTOR_TEMP, G->levels[lh], u_id, RESTRICT_CELL);
  restriction(G->levels[lh + 2], VECTOR_TEMP, G->levels[lh + 1], u_id, RESTRICT_CELL);
  add_vectors(G->levels[lh + 1], VECTOR_TEMP, 1.0, u_id, -1.0, VECTOR_TEMP);
  add_vectors(G->levels[lh + 2], VECTOR_TEMP, 1.0, u_id, -1.0, VECTOR_TEMP);
  double d21 = norm(G->levels[lh + 1], VECTOR_TEMP);
  double d42 = norm(G->levels[lh + 2], VECTOR_TEMP);
  hpgmg_last_solve.richardson_error = d21;
  hpgmg_last_solve.richardson_order = log(d42 / d21) / log(2);
  SAY(G->my_rank, "  h=%0.15e  ||error||=%0.15e\n", G->levels[lh]->h, d21);
  SAY(G->my_rank, "  order=%0.3f\n", hpgmg_last_solve.richardson_order);
}

void MGVCycle(mg_type *G, int e_id, int R_id, double a, double b, int l) {
  level_type *L = G->levels[l];
  hpgmg_tick t;
  if (!L->active) return;
  if (l == G->num_levels - 1) {
    t = hpgmg_tick_begin(L, &L->timers.Total, "bottom solve");
    if (!hpgmg_vcycle_legs_fused(&G->levels[l], 1, e_id, R_id, a, b, HPGMG_LEG_BOTTOM)) {
      seg_close();                                 /* the host-driven Krylov solver synchronises */
      IterativeSolver(L, e_id, R_id, a, b, MG_DEFAULT_BOTTOM_NORM);
    }
    hpgmg_tick_end(t);
    return;
  }
  const int opened_here = is_small(G, l) && !seg_open_now;
  if (is_small(G, l)) seg_open();
  /* tiny levels: the plugin may run the rest of this V-cycle (or each of its legs) as one fused operation */
  const int maybe_tail = is_small(G, l);         /* only small levels can be fused: do not pay for a tick around a refusal on the big ones */
  if (maybe_tail) t = hpgmg_tick_begin(L, &L->timers.Total, "V-cycle tail (fused)");
  if (maybe_tail && hpgmg_vcycle_legs_fused(&G->levels[l], G->num_levels - l, e_id, R_id, a, b, HPGMG_LEG_VCYCLE)) {
    hpgmg_tick_end(t);
    if (opened_here) seg_close();
    return;
  }
  if (maybe_tail && hpgmg_vcycle_legs_fused(&G->levels[l], G->num_levels - l, e_id, R_id, a, b, HPGMG_LEG_DOWN)) {
    hpgmg_tick_end(t);
    MGVCycle(G, e_id, R_id, a, b, G->num_levels - 1);          /* bottom solve (closes the segment) */
    seg_open();
    t = hpgmg_tick_begin(L, &L->timers.Total, "V-cycle tail, up leg (fused)");
    if (!hpgmg_vcycle_legs_fused(&G->levels[l], G->num_levels - l, e_id, R_id, a, b, HPGMG_LEG_UP)) { fprintf(stderr, "fused V-cycle leg refused after being accepted\n"); exit(1); }
    hpgmg_tick_end(t);
    if (opened_here) seg_close();
    return;
  }
  if (maybe_tail) hpgmg_tick_end(t);          /* nothing was launched: adds (next to) nothing */
  t = hpgmg_tick_begin(L, &L->timers.Total, "V-cycle down leg");
  if (!hpgmg_smooth_in_cycle(L, e_id, R_id, a, b)) smooth(L, e_id, R_id, a, b);
  if (!hpgmg_residual_restrict_zero_fused(G->levels[l + 1], R_id, L, e_id, R_id, a, b, e_id)) {
    residual(L, VECTOR_TEMP, e_id, R_id, a, b);
    if (!hpgmg_restrict_zero_fused(G->levels[l + 1], R_id, L, VECTOR_TEMP, e_id)) {
      restriction(G->levels[l + 1], R_id, L, VECTOR_TEMP, RESTRICT_CELL);
      zero_vector(G->levels[l + 1], e_id);
    }
  }
  hpgmg_tick_end(t);

  if (!is_small(G, l) && is_small(G, l + 1)) seg_open();   /* everything below this point is launch bound */
  MGVCycle(G, e_id, R_id, a, b, l + 1);
  if (is_small(G, l)) seg_open();                           /* re-open after the bottom solve */
  else seg_close();                                         /* back on a bandwidth-bound level */

  t = hpgmg_tick_begin(L, &L->timers.Total, "V-cycle up leg");
  if (!hpgmg_interp_smooth_fused(L, e_id, R_id, G->levels[l + 1], a, b)) {
    interpolation_vcycle(L, e_id, 1.0, G->levels[l + 1], e_id);
    if (!hpgmg_smooth_in_cycle(L, e_id, R_id, a, b)) smooth(L, e_id, R_id, a, b);
  }
  hpgmg_tick_end(t);
  if (opened_here) seg_close();
}

/* residual check shared by MGSolve and FMGSolve; returns 1 when converged */
static int check_residual(mg_type *G, int l, int e_id, int F_id, double a, double b, double norm_of_F,
                          double rtol, const char *label) {
  level_type *L = G->levels[l];
  hpgmg_tick t = hpgmg_tick_begin(L, &L->timers.Total, "residual check");
  if (L->must_subtract_mean == 1) {
    double m = mean(L, e_id);
    shift_vector(L, e_id, e_id, -m);
  }
  double r;
  if (!hpgmg_residual_norm_fused(L, -1, e_id, F_id, a, b, &r)) {   /* -1: VECTOR_TEMP is scratch here, nothing reads the residual after its norm */
    residual(L, VECTOR_TEMP, e_id, F_id, a, b);
    r = norm(L, VECTOR_TEMP);
  }
  hpgmg_tick_end(t);
  hpgmg_last_solve.norm_of_F = norm_of_F;
  hpgmg_last_solve.norm_of_residual = r;
  SAY(L->my_rank, "%s  norm=%1.15e  rel=%1.15e  ", label, r, r / norm_of_F);
  return (r / norm_of_F < rtol);
}

void MGSolve(mg_type *G, int onLevel, int u_id, int F_id, double a, double b, double rtol) {
  level_type *L = G->levels[onLevel];
  const int e_id = u_id, R_id = VECTOR_R, maxVCycles = 20;
  char label[64];
  int v;
  G->MGSolves_performed++;
  if (!L->active) return;
  SAY(L->my_rank, "MGSolve... ");
  double t0 = now();
  seg_reset(G, onLevel + 64, u_id, F_id, a, b);
  double norm_of_F = norm(L, F_id);
  zero_vector(L, e_id);
  scale_vector(L, R_id, 1.0, F_id);
  for (v = 0; v < maxVCycles; v++) {
    L->vcycles_from_this_level++;
    MGVCycle(G, e_id, R_id, a, b, onLevel);
    seg_close();
    snprintf(label, sizeof label, v > 0 ? "\n           v-cycle=%2d" : "v-cycle=%2d", v + 1);
    hpgmg_last_solve.vcycles = v + 1;
    if (check_residual(G, onLevel, e_id, F_id, a, b, norm_of_F, rtol, label)) break;
  }
  G->timers.MGSolve += now() - t0;
  SAY(L->my_rank, "done (%f seconds)\n", now() - t0);
}

/* Conjugate gradients preconditioned with one V-cycle per iteration (reference mg.c:1500-1605; Saad, Iterative Methods for Sparse Linear
 * Systems, algorithm 9.1 with M^-1 = MGVCycle).  Every level gets three more vectors (p, Ap, z) the first time; the residual test uses the
 * TRUE residual F - A x after every update (mg.c:1579-1580), what is printed is what the reference prints.  The operator calls and their
 * order are the reference's, so the iterates -- which hang on dot products over the fine level -- are too. */
void MGPCG(mg_type *G, int onLevel, int x_id, int F_id, double a, double b, double rtol) {
  level_type *L = G->levels[onLevel];
  const int r_id = VECTOR_R, p_id = hpgmg_vectors_reserved(), Ap_id = p_id + 1, z_id = p_id + 2, jMax = 20;
  int l, j = 0, failed = 0, converged = 0;
  if (!L->active) return;
  for (l = 0; l < G->num_levels; l++) create_vectors(G->levels[l], hpgmg_vectors_reserved() + 3);
  SAY(L->my_rank, "MGPCG...  ");
  const double t0 = now();
  seg_reset(G, onLevel + 128, x_id, F_id, a, b);
  G->MGSolves_performed++;
  zero_vector(L, x_id);
  residual(L, r_id, x_id, F_id, a, b);
  if (L->must_subtract_mean == 1) { const double m = mean(L, r_id); shift_vector(L, r_id, r_id, -m); }
  const double norm_of_r0 = norm(L, r_id);
  if (norm_of_r0 == 0.0) converged = 1;                                   /* entered with the exact solution */
  L->vcycles_from_this_level++;
  zero_vector(L, z_id);
  MGVCycle(G, z_id, r_id, a, b, onLevel);                                 /* z = M^-1 r */
  seg_close();
  scale_vector(L, p_id, 1.0, z_id);
  double r_dot_z = dot(L, r_id, z_id);
  while (j < jMax && !failed && !converged) {
    j++; L->Krylov_iterations++;
    apply_op(L, Ap_id, p_id, a, b);
    const double Ap_dot_p = dot(L, Ap_id, p_id);
    if (Ap_dot_p == 0.0) { failed = 1; break; }                           /* pivot breakdown */
    const double alpha = r_dot_z / Ap_dot_p;
    if (isinf(alpha)) { failed = 1; break; }
    add_vectors(L, x_id, 1.0, x_id, alpha, p_id);
    add_vectors(L, r_id, 1.0, r_id, -alpha, Ap_id);
    if (L->must_subtract_mean == 1) { const double m = mean(L, r_id); shift_vector(L, r_id, r_id, -m); }
    residual(L, VECTOR_TEMP, x_id, F_id, a, b);                           /* the true residual decides */
    const double norm_of_r = norm(L, VECTOR_TEMP);
    if (norm_of_r == 0.0) { converged = 1; break; }
    if (j > 1) SAY(L->my_rank, "\n          ");
    SAY(L->my_rank, "iter=%3d  norm=%1.15e  rel=%1.15e  ", j, norm_of_r, norm_of_r / norm_of_r0);
    hpgmg_last_solve.norm_of_F = norm_of_r0; hpgmg_last_solve.norm_of_residual = norm_of_r; hpgmg_last_solve.vcycles = j;
    if (norm_of_r / norm_of_r0 < rtol) break;
    L->vcycles_from_this_level++;
    zero_vector(L, z_id);
    MGVCycle(G, z_id, r_id, a, b, onLevel);
    seg_close();
    const double r_dot_z_new = dot(L, r_id, z_id);
    if (r_dot_z_new == 0.0) { failed = 1; break; }                        /* Lanczos breakdown */
    const double beta = r_dot_z_new / r_dot_z;
    if (isinf(beta)) { failed = 1; break; }
    add_vectors(L, p_id, 1.0, z_id, beta, p_id);
    r_dot_z = r_dot_z_new;
  }
  G->timers.MGSolve += now() - t0;
  SAY(L->my_rank, "done (%f seconds)\n", now() - t0);
}

/* V-cycles FMGSolve may add after its F-cycle until the residual has dropped by rtol: 0, or 20 = the reference built with -DUNLIMIT_FMG_ITERATIONS (mg.c:1239-1247) */
static int hpgmg_fmg_vcycles = 0;
void hpgmg_set_fmg_vcycles(int n) { hpgmg_fmg_vcycles = n > 0 ? n : 0; }
/* The benchmark step is zero_vector(u) followed by FMGSolve (hpgmg-fv.c:77-85).  FMGSolve's first access to u on its level is the write of
 * interpolation_fcycle (mg.c:1295), so the zeroing may wait until then and go out WITH it (hpgmg_zero_interpolation_fcycle_fused): the caller
 * says "u is to be zeroed first" instead of calling zero_vector, and FMGSolve zeroes it where it would first be touched. */
static int fmg_zero_u_first = 0;
void hpgmg_fmg_zero_u_first(void) { fmg_zero_u_first = 1; }
static void fmg_solve_once(mg_type *G, int onLevel, int u_id, int F_id, double a, double b, double rtol);
/* FMGSolve reads F and overwrites u (its first access to u is interpolation_fcycle's write, prescale 0.0): a solve that a plugin reports as failed as a
 * whole (hpgmg_solve_attempt_end: a fused launch of the HIP plugin did not get all its workgroups running) is repeated from the same inputs -- the
 * plugin has switched the failing form off, so the second attempt is the launch-by-launch path.  The reference has no such failure mode; its operators
 * complete (chebyshev.c:9-12 is its only exit). */
void FMGSolve(mg_type *G, int onLevel, int u_id, int F_id, double a, double b, double rtol) {
  int attempt, l, booked[64];
  for (attempt = 0; attempt < 2; attempt++) {
    for (l = 0; l < G->num_levels && l < 64; l++) booked[l] = G->levels[l]->vcycles_from_this_level;
    hpgmg_solve_attempt_begin();
    fmg_solve_once(G, onLevel, u_id, F_id, a, b, rtol);
    if (!hpgmg_solve_attempt_end()) return;
    /* what the failed attempt booked: the solve counts once; u is cleared so that 0.0 * u is 0.0 again */
    G->MGSolves_performed--;
    for (l = 0; l < G->num_levels && l < 64; l++) G->levels[l]->vcycles_from_this_level = booked[l];
    if (G->levels[onLevel]->active) zero_vector(G->levels[onLevel], u_id);
  }
  fprintf(stderr, "hpgmg: FMGSolve failed twice\n");
  exit(1);
}
static void fmg_solve_once(mg_type *G, int onLevel, int u_id, int F_id, double a, double b, double rtol) {
  /* one F-cycle; further V-cycles only on request (hpgmg_set_fmg_vcycles) */
  const int e_id = u_id, R_id = VECTOR_R, maxVCycles = hpgmg_fmg_vcycles;
  const int bottom = G->num_levels - 1;
  level_type *L = G->levels[onLevel];
  char label[64];
  int l, v;
  int u_to_zero = fmg_zero_u_first;
  hpgmg_tick t;
  fmg_zero_u_first = 0;
  G->MGSolves_performed++;
  if (!L->active) return;
  SAY(L->my_rank, "FMGSolve... ");
  const double t0 = now();
  seg_reset(G, onLevel, u_id, F_id, a, b);

  t = hpgmg_tick_begin(L, &L->timers.Total, "norm(F), R = F");
  double norm_of_F = 0.0;
  int first_restriction = onLevel;               /* the plugin may do norm, copy and the first restriction in one pass over F */
  int norm_deferred = 0;                         /* norm(F) is used only by the check at the end: the plugin may let the host run on and hand it over there */
  if (onLevel < bottom && hpgmg_norm_scale_restrict_fused_deferred(L, F_id, R_id, G->levels[onLevel + 1])) { first_restriction = onLevel + 1; norm_deferred = 1; }
  else if (onLevel < bottom && hpgmg_norm_scale_restrict_fused(L, F_id, R_id, G->levels[onLevel + 1], &norm_of_F)) first_restriction = onLevel + 1;
  else {
    norm_of_F = norm(L, F_id);
    scale_vector(L, R_id, 1.0, F_id);
  }
  hpgmg_tick_end(t);

  /* the plugin may run everything below some small level -- the rest of the restrictions, the bottom solve and the climb back up to
   * that level, interpolation_fcycle + V-cycle per level -- as one fused operation */
  int ftail = bottom;
  for (l = onLevel; l < bottom; l++) if (is_small(G, l) && hpgmg_vcycle_legs_fused(&G->levels[l], G->num_levels - l, e_id, R_id, a, b, HPGMG_LEG_FCYCLE_TAIL_ASK)) { ftail = l; break; }

  if (u_to_zero && ftail <= onLevel) { zero_vector(L, u_id); u_to_zero = 0; }      /* no interpolation onto this level will come: zero it now, as the caller would have */
  for (l = first_restriction; l < ftail; l++) {           /* carry the right-hand side down */
    if (is_small(G, l)) seg_open();
    t = hpgmg_tick_begin(G->levels[l], &G->levels[l]->timers.Total, "restrict R");
    restriction(G->levels[l + 1], R_id, G->levels[l], R_id, RESTRICT_CELL);
    hpgmg_tick_end(t);
  }

  if (ftail < bottom) {
    if (is_small(G, ftail)) seg_open();
    t = hpgmg_tick_begin(G->levels[ftail], &G->levels[ftail]->timers.Total, "F-cycle tail (fused)");
    if (!hpgmg_vcycle_legs_fused(&G->levels[ftail], G->num_levels - ftail, e_id, R_id, a, b, HPGMG_LEG_FCYCLE_TAIL)) { fprintf(stderr, "fused F-cycle tail refused after being accepted\n"); exit(1); }
    hpgmg_tick_end(t);
    for (l = bottom - 1; l >= ftail; l--) G->levels[l]->vcycles_from_this_level++;
    seg_close();
  } else {
    if (bottom > onLevel) { t = hpgmg_tick_begin(G->levels[bottom], &G->levels[bottom]->timers.Total, "zero e (bottom)"); zero_vector(G->levels[bottom], e_id); hpgmg_tick_end(t); }
    if (is_small(G, bottom)) seg_open();
    MGVCycle(G, e_id, R_id, a, b, bottom);          /* the bottom solve (times itself) */
  }

  for (l = ftail - 1; l >= onLevel; l--) {       /* climb: prolong the solution, then one V-cycle */
    if (is_small(G, l)) seg_open();
    t = hpgmg_tick_begin(G->levels[l], &G->levels[l]->timers.Total, "interpolation_fcycle");
    if (l == onLevel && u_to_zero) {
      if (!hpgmg_zero_interpolation_fcycle_fused(G->levels[l], e_id, G->levels[l + 1], e_id)) { zero_vector(G->levels[l], e_id); interpolation_fcycle(G->levels[l], e_id, 0.0, G->levels[l + 1], e_id); }
      u_to_zero = 0;
    } else if (is_small(G, l) && hpgmg_vcycle_legs_fused(&G->levels[l], G->num_levels - l, e_id, R_id, a, b, HPGMG_LEG_FCYCLE_STEP)) {      /* the plugin ran this step whole: interpolation + V-cycle */
      hpgmg_tick_end(t);
      G->levels[l]->vcycles_from_this_level++;
      seg_close();
      continue;
    } else interpolation_fcycle(G->levels[l], e_id, 0.0, G->levels[l + 1], e_id);
    hpgmg_tick_end(t);
    G->levels[l]->vcycles_from_this_level++;
    MGVCycle(G, e_id, R_id, a, b, l);
    seg_close();
  }

  if (norm_deferred) norm_of_F = hpgmg_norm_deferred_fetch(L);
  hpgmg_last_solve.vcycles = 0;
  for (v = -1; v < maxVCycles; v++) {
    if (v >= 0) { L->vcycles_from_this_level++; MGVCycle(G, e_id, R_id, a, b, onLevel); hpgmg_last_solve.vcycles = v + 1; }
    if (v >= 0) snprintf(label, sizeof label, "\n            v-cycle=%2d", v + 1);
    else        snprintf(label, sizeof label, "f-cycle   ");
    if (check_residual(G, onLevel, e_id, F_id, a, b, norm_of_F, rtol, label)) break;
  }
  G->timers.MGSolve += now() - t0;
  SAY(L->my_rank, "done (%f seconds)\n", now() - t0);
}

/* ------------------------------------------------------------------ timing table */
void MGPrintTiming(mg_type *G, int fromLevel) {
  const int nl = G->num_levels;
  { int c; for (c = 0; c < nl; c++) hpgmg_level_sync_counters(G->levels[c]); }
  if (G->my_rank != 0 || !hpgmg_verbose) return;
  const double scale = 1.0 / (double)(G->MGSolves_performed ? G->MGSolves_performed : 1);
  int l;
  printf("\n\n");
  printf("level                     "); for (l = fromLevel; l < nl; l++) printf("%12d ", l - fromLevel); printf("\n");
  printf("level dimension           "); for (l = fromLevel; l < nl; l++) printf("%10d^3 ", G->levels[l]->dim.i); printf("\n");
  printf("box dimension             "); for (l = fromLevel; l < nl; l++) printf("%10d^3 ", G->levels[l]->box_dim); printf("       total\n");
  printf("------------------        "); for (l = fromLevel; l < nl + 1; l++) printf("------------ "); printf("\n");
#define ROW(title, field) do { double tot = 0; printf("%-26s", title); \
    for (l = fromLevel; l < nl; l++) { double v = scale * G->levels[l]->timers.field; tot += v; printf("%12.6f ", v); } \
    printf("%12.6f\n", tot); } while (0)
  ROW("smooth", smooth);
  ROW("  max", smooth);
  ROW("  min", smooth);
  ROW("residual", residual);
  ROW("applyOp", apply_op);
  ROW("BLAS1", blas1);
  ROW("BLAS3", blas3);
  ROW("Boundary Conditions", boundary_conditions);
  ROW("Restriction", restriction_total);
  ROW("  local restriction", restriction_local);
  ROW("Interpolation", interpolation_total);
  ROW("  local interpolation", interpolation_local);
  ROW("Ghost Zone Exchange", ghostZone_total);
  ROW("  local exchange", ghostZone_local);
  printf("------------------        "); for (l = fromLevel; l < nl + 1; l++) printf("------------ "); printf("\n");
  ROW("Total by level", Total);
#undef ROW
  printf("\n");
  printf("   Total time in MGBuild  %12.6f seconds\n", G->timers.MGBuild);
  printf("   Total time in MGSolve  %12.6f seconds\n", scale * G->timers.MGSolve);
  printf("      number of v-cycles  %12d\n", G->levels[fromLevel]->vcycles_from_this_level / (G->MGSolves_performed ? G->MGSolves_performed : 1));
  printf("Bottom solver iterations  %12d\n", G->levels[nl - 1]->Krylov_iterations / (G->MGSolves_performed ? G->MGSolves_performed : 1));
  printf("\n\n");
  fflush(stdout);
}

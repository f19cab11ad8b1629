// blas1.hip -- interior-only vector operations, reductions and the 7-pt operator
// rebuild (reference finite-volume/source/operators/misc.c and operators.7pt.c:158-227).
//
// Work unit = one unit-stride row (j,k) of one box per wave: 64 lanes stride
// along i, so every load/store instruction of a wave covers 512 contiguous
// bytes.  All of these are pure streaming kernels (8-24 B per cell), HBM-bound.
// Sums (dot, mean) reproduce the reference's single-thread order: one partial
// per dim x 8 x 8 tile accumulated in k,j,i order, partials added in tile-list
// order (misc.c:261-269).  They are only ever called on the coarsest level
// (BiCGStab) and in untimed setup, so one lane per tile is enough.
#include "common.hpp"

namespace hpgmg {

constexpr int kRowsPerBlock = 4;   // 256 threads = 4 waves = 4 rows

struct RowIter {                   // decode "row id" -> box, j, k over a (rows_per_side)^2 * num_boxes space
  int box, j, k;
};
__device__ __forceinline__ bool row_of(int row, int side, int num_boxes, RowIter &r) {
  const int per_box = side * side;
  if (row >= per_box * num_boxes) return false;
  r.box = row / per_box;
  const int rem = row - r.box * per_box;
  r.k = rem / side;
  r.j = rem - r.k * side;
  return true;
}
static inline int rows_grid(int rows) { return (rows + kRowsPerBlock - 1) / kRowsPerBlock; }

enum { OP_AXPBY = 0, OP_MUL, OP_INVERT, OP_SCALE, OP_SHIFT, OP_COLOR, OP_RANDOM };
struct EwArgs { int id_c, id_a, id_b; double sa, sb; int colors, ic, jc, kc; };

template <int OP>
__global__ __launch_bounds__(256) void elementwise_kernel(const hpgmg_hip_level L, const EwArgs A) {
  RowIter r;
  const int row = blockIdx.x * kRowsPerBlock + threadIdx.x / 64, lane = threadIdx.x % 64;
  if (!row_of(row, L.dim, L.num_boxes, r)) return;
  const int base = r.j * L.jStride + r.k * L.kStride;
  double *c = vec_origin(L, r.box, A.id_c) + base;
  const double *pa = (OP == OP_COLOR || OP == OP_RANDOM) ? nullptr : vec_origin(L, r.box, A.id_a) + base;
  const double *pb = (OP == OP_AXPBY || OP == OP_MUL) ? vec_origin(L, r.box, A.id_b) + base : nullptr;
  for (int i = lane; i < L.dim; i += 64) {
    double v;
    if (OP == OP_AXPBY)       v = A.sa * pa[i] + A.sb * pb[i];
    else if (OP == OP_MUL)    v = A.sa * pa[i] * pb[i];
    else if (OP == OP_INVERT) v = A.sa / pa[i];
    else if (OP == OP_SCALE)  v = A.sa * pa[i];
    else if (OP == OP_SHIFT)  v = pa[i] + A.sa;
    else if (OP == OP_COLOR) {
      const double si = ((i + L.box_low[3 * r.box] + A.ic) % A.colors == 0) ? 1.0 : 0.0;
      const double sj = ((r.j + L.box_low[3 * r.box + 1] + A.jc) % A.colors == 0) ? 1.0 : 0.0;
      const double sk = ((r.k + L.box_low[3 * r.box + 2] + A.kc) % A.colors == 0) ? 1.0 : 0.0;
      v = si * sj * sk;
    } else v = -1.000 + 2.0 * (double)(i ^ r.j ^ r.k ^ 0x1);   // misc.c:500, literally
    c[i] = v;
  }
}

// zero_vector / init_vector: the whole padded box including ghosts; ghosts := 0 (misc.c:26-41, :68-86)
__global__ __launch_bounds__(256) void fill_kernel(const hpgmg_hip_level L, int id, double inside) {
  RowIter r;
  const int side = L.dim + 2 * L.ghosts;
  const int row = blockIdx.x * kRowsPerBlock + threadIdx.x / 64, lane = threadIdx.x % 64;
  if (!row_of(row, side, L.num_boxes, r)) return;
  double *v = L.box_base[r.box] + (size_t)id * (size_t)L.volume + r.j * L.jStride + r.k * L.kStride;
  const bool ghost_row = (r.j < L.ghosts) || (r.k < L.ghosts) || (r.j >= L.dim + L.ghosts) || (r.k >= L.dim + L.ghosts);
  for (int i = lane; i < side; i += 64) {
    const bool ghost = ghost_row || (i < L.ghosts) || (i >= L.dim + L.ghosts);
    v[i] = ghost ? 0.0 : inside;
  }
}

// ---- max reductions ------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v) {
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(v, off, 64); v = (o > v) ? o : v; }
  return v;
}
__device__ __forceinline__ void block_max_store_at(double v, double *partials, int slot) {
  __shared__ double smem[kRowsPerBlock];
  v = wave_max(v);
  if (threadIdx.x % 64 == 0) smem[threadIdx.x / 64] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = smem[0];
    for (int w = 1; w < kRowsPerBlock; w++) m = (smem[w] > m) ? smem[w] : m;
    partials[slot] = m;
  }
}
__device__ __forceinline__ void block_max_store(double v, double *partials) {
  __shared__ double smem[kRowsPerBlock];
  v = wave_max(v);
  if (threadIdx.x % 64 == 0) smem[threadIdx.x / 64] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = smem[0];
    for (int w = 1; w < kRowsPerBlock; w++) m = (smem[w] > m) ? smem[w] : m;
    partials[blockIdx.x] = m;
  }
}

// grid = (workgroups per box, boxes); each wave strides over the rows of its box with four rows in flight;
// block maxima go to `partials` (the last workgroup to finish folds them and publishes the result)
__device__ __forceinline__ double absmax_rows(const hpgmg_hip_level &L, int id) {
  const int dim = L.dim, rows = dim * dim, lane = threadIdx.x % 64;
  const int wave = blockIdx.x * kRowsPerBlock + threadIdx.x / 64, waves = gridDim.x * kRowsPerBlock;
  const int sh = ((dim & (dim - 1)) == 0) ? __builtin_ctz(dim) : -1;
  const double *base = vec_origin(L, blockIdx.y, id);
  double m = 0.0;
  for (int row0 = wave; row0 < rows; row0 += 4 * waves) {
    const double *p[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int row = row0 + u * waves;
      row = (row < rows) ? row : row0;                          // a repeated row does not change a maximum
      const int k = (sh >= 0) ? (row >> sh) : (row / dim), j = row - k * dim;
      p[u] = base + j * L.jStride + k * L.kStride;
    }
    for (int i = lane; i < dim; i += 64) {
      double f[4];
#pragma unroll
      for (int u = 0; u < 4; u++) f[u] = fabs(p[u][i]);
#pragma unroll
      for (int u = 0; u < 4; u++) m = (f[u] > m) ? f[u] : m;
    }
  }
  return m;
}

// Scalar results go straight to a pinned host slot {value, sequence}: the last lane stores the value,
// fences at system scope, then stores the launch's sequence number; the host polls the sequence
// instead of paying a full stream synchronisation (reductions gate the host-driven BiCGStab).
// (ResultSlot and publish(): common.hpp)

// small levels: one workgroup walks every row and stores the final max itself
__global__ __launch_bounds__(256) void absmax_small_kernel(const hpgmg_hip_level L, int id, ResultSlot *result, unsigned long long seq) {
  __shared__ double smem[4];
  const int rows = L.num_boxes * L.dim * L.dim, lane = threadIdx.x % 64;
  double m = 0.0;
  for (int row = threadIdx.x / 64; row < rows; row += 4) {
    RowIter r;
    row_of(row, L.dim, L.num_boxes, r);
    const double *p = vec_origin(L, r.box, id) + r.j * L.jStride + r.k * L.kStride;
    for (int i = lane; i < L.dim; i += 64) { const double f = fabs(p[i]); m = (f > m) ? f : m; }
  }
  m = wave_max(m);
  if (lane == 0) smem[threadIdx.x / 64] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int w = 1; w < 4; w++) m = (smem[w] > m) ? smem[w] : m; publish(result, m, seq); }
}

// (a single-launch "last workgroup folds" variant was measured 4x slower: the agent-scope fence each
// workgroup needs before signalling costs an L2 write-back on a multi-XCD part; a second tiny launch is cheaper)
__global__ __launch_bounds__(256) void absmax_kernel(const hpgmg_hip_level L, int id, double *partials) {
  block_max_store_at(absmax_rows(L, id), partials, blockIdx.x + gridDim.x * blockIdx.y);
}

__global__ __launch_bounds__(256) void final_max_kernel(const double *partials, int n, double init, ResultSlot *result, unsigned long long seq) {
  __shared__ double smem[4];
  double m = init;
  // four loads in flight per lane: the chain of one load per iteration made 8192 partials a 10 us launch (a maximum is exact under any order)
  double m1 = init, m2 = init, m3 = init;
  int t = threadIdx.x;
  for (; t + 768 < n; t += 1024) {
    const double a0 = partials[t], a1 = partials[t + 256], a2 = partials[t + 512], a3 = partials[t + 768];
    m = (a0 > m) ? a0 : m; m1 = (a1 > m1) ? a1 : m1; m2 = (a2 > m2) ? a2 : m2; m3 = (a3 > m3) ? a3 : m3;
  }
  for (; t < n; t += 256) m = (partials[t] > m) ? partials[t] : m;
  m = (m1 > m) ? m1 : m; m2 = (m3 > m2) ? m3 : m2; m = (m2 > m) ? m2 : m;
  m = wave_max(m);
  if (threadIdx.x % 64 == 0) smem[threadIdx.x / 64] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int w = 1; w < 4; w++) m = (smem[w] > m) ? smem[w] : m; publish(result, (smem[0] > m) ? smem[0] : m, seq); }
}

// norm(F) ; scale_vector(R, 1.0, F) ; restriction(coarse R <- R, RESTRICT_CELL) -- the first three operators of FMGSolve (mg.c:1262-1270)
// -- as one pass over F: a wave owns one COARSE row of a box (its four fine rows, a lane the 2 x 2 x 2 children of a coarse cell,
// 16-byte accesses), stores R = 1.0 * F, forms the restricted value in the reference's order (restriction.c:54-57) and the maximum of |F|.
// NT: R is written past the caches (a level larger than the infinity cache: the first reader of R comes a whole cycle later)
typedef double __attribute__((ext_vector_type(2))) pair_v;
template <bool NT> __device__ __forceinline__ void store_pair(double *p, double2 v) {
  pair_v w; w.x = v.x; w.y = v.y;
  if (NT) __builtin_nontemporal_store(w, (pair_v __attribute__((address_space(1))) *)as_global(p));
  else *(pair_v __attribute__((address_space(1))) *)as_global(p) = w;
}
template <bool NT>
__global__ __launch_bounds__(256) void norm_copy_restrict_kernel(const hpgmg_hip_level L, int f_id, int r_id, const hpgmg_hip_level Lc, int rc_id,
                                                                 const int *__restrict__ map, double *partials) {
  const int half = L.dim >> 1, rows_per_box = half * half, lane = threadIdx.x % 64;
  const int row = blockIdx.x * kRowsPerBlock + threadIdx.x / 64;
  double m = 0.0;
  if (row < rows_per_box * L.num_boxes) {
    const int box = row / rows_per_box, rem = row - box * rows_per_box, ck = rem / half, cj = rem - ck * half;
    const int jS = L.jStride, kS = L.kStride, base = 2 * cj * jS + 2 * ck * kS;
    const double *f = vec_origin(L, box, f_id) + base;
    double *r = vec_origin(L, box, r_id) + base;
    const int *mp = map + 4 * box;
    double *c = vec_origin(Lc, mp[0], rc_id) + mp[1] + (mp[2] + cj) * Lc.jStride + (mp[3] + ck) * Lc.kStride;
    for (int ci = lane; ci < half; ci += 64) {
      const double2 a = *reinterpret_cast<const double2 *>(f + 2 * ci), b = *reinterpret_cast<const double2 *>(f + 2 * ci + jS);
      const double2 cc = *reinterpret_cast<const double2 *>(f + 2 * ci + kS), d = *reinterpret_cast<const double2 *>(f + 2 * ci + jS + kS);
      const double2 ra = make_double2(1.0 * a.x, 1.0 * a.y), rb = make_double2(1.0 * b.x, 1.0 * b.y);
      const double2 rc = make_double2(1.0 * cc.x, 1.0 * cc.y), rd = make_double2(1.0 * d.x, 1.0 * d.y);
      store_pair<NT>(r + 2 * ci, ra); store_pair<NT>(r + 2 * ci + jS, rb);
      store_pair<NT>(r + 2 * ci + kS, rc); store_pair<NT>(r + 2 * ci + jS + kS, rd);
      double v = ra.x + ra.y; v = v + rb.x; v = v + rb.y; v = v + rc.x; v = v + rc.y; v = v + rd.x; v = v + rd.y;
      c[ci] = v * 0.125;
      double q;
      q = fabs(a.x); m = (q > m) ? q : m;  q = fabs(a.y); m = (q > m) ? q : m;  q = fabs(b.x); m = (q > m) ? q : m;  q = fabs(b.y); m = (q > m) ? q : m;
      q = fabs(cc.x); m = (q > m) ? q : m; q = fabs(cc.y); m = (q > m) ? q : m; q = fabs(d.x); m = (q > m) ? q : m;  q = fabs(d.y); m = (q > m) ? q : m;
    }
  }
  block_max_store(m, partials);
}

// ---- ordered sums ----------------------------------------------------------------------
// one lane per dim x 8 x 8 tile, tiles numbered as level.c:1184-1210 builds my_blocks
// kFinish: the whole level fits one 64-lane workgroup, which then also adds the partials in tile
// order and stores the result (to pinned host memory) -- one launch per reduction on small levels.
template <bool kFinish>
__global__ __launch_bounds__(64) void tile_sum_kernel(const hpgmg_hip_level L, int id_a, int id_b, double *partials, ResultSlot *result, unsigned long long seq) {
  __shared__ double part[64];
  const int tiles_side = (L.dim + BLOCKCOPY_TILE_J - 1) / BLOCKCOPY_TILE_J;
  const int tiles_per_box = tiles_side * tiles_side;
  const int ntiles = tiles_per_box * L.num_boxes;
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (kFinish) part[threadIdx.x] = 0.0;
  if (t >= ntiles) { if (kFinish) __syncthreads(); return; }
  const int box = t / tiles_per_box, rem = t % tiles_per_box;
  const int k0 = (rem / tiles_side) * BLOCKCOPY_TILE_K, j0 = (rem % tiles_side) * BLOCKCOPY_TILE_J;
  const int k1 = min(k0 + BLOCKCOPY_TILE_K, L.dim), j1 = min(j0 + BLOCKCOPY_TILE_J, L.dim);
  const double *pa = vec_origin(L, box, id_a);
  const double *pb = (id_b >= 0) ? vec_origin(L, box, id_b) : nullptr;
  double acc = 0.0;
  for (int k = k0; k < k1; k++) for (int j = j0; j < j1; j++) {
    const int base = j * L.jStride + k * L.kStride;
    for (int i = 0; i < L.dim; i++) acc += pb ? pa[base + i] * pb[base + i] : pa[base + i];
  }
  if (!kFinish) { partials[t] = acc; return; }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0.0; for (int q = 0; q < ntiles; q++) s += part[q]; publish(result, s, seq); }
}
// The same sum on a level of many tiles: ONE WAVE per tile.  A tile's partial is a chain of dim x 8 x 8 dependent additions in k, j, i order (misc.c:261-269)
// -- that order is the contract, so the chain itself cannot be cut: ~8 cycles per addition, 8192 of them for a tile of a 128^3 box = ~30 us, which with every
// tile of the level on a wave of its own is also the time of the launch.  What can be taken out of the chain is memory: with a LANE per tile (above) every load
// of the chain was a 64-byte sector per lane and a round trip per element (1.35 ms at 256^3).  Here the wave's 64 lanes fetch EIGHT rows (dim cells each, coalesced)
// ahead of the chain into LDS -- for a dot product they form the products -- and lane 0 adds the values in order from LDS (one row ahead: 153 us, a memory round
// trip per row is longer than a row's chain).
constexpr int kSumMaxChunks = 8;      // rows of up to 512 cells
constexpr int kSumGroup = 8;          // rows fetched together (one k plane of a tile): enough loads in flight to hide a memory round trip behind a group's chain
template <int CHUNKS>
__global__ __launch_bounds__(64) void tile_sum_wave_kernel(const hpgmg_hip_level L, int id_a, int id_b, double *partials) {
  extern __shared__ double sum_rows[];      // [2][kSumGroup][dim]
  typedef const double __attribute__((address_space(3))) *ldsrow;
  const int tiles_side = (L.dim + BLOCKCOPY_TILE_J - 1) / BLOCKCOPY_TILE_J, tiles_per_box = tiles_side * tiles_side;
  const int t = (int)blockIdx.x, lane = (int)threadIdx.x, dim = L.dim;
  const int box = t / tiles_per_box, rem = t % tiles_per_box;
  const int k0 = (rem / tiles_side) * BLOCKCOPY_TILE_K, j0 = (rem % tiles_side) * BLOCKCOPY_TILE_J;
  const int k1 = min(k0 + BLOCKCOPY_TILE_K, dim), j1 = min(j0 + BLOCKCOPY_TILE_J, dim), nj = j1 - j0, nrows = (k1 - k0) * nj;
  const gcptr pa = as_global(vec_origin(L, box, id_a));
  const gcptr pb = (id_b >= 0) ? as_global(vec_origin(L, box, id_b)) : pa;
  const bool dot = id_b >= 0;
  const int ngroups = (nrows + kSumGroup - 1) / kSumGroup;
  double v[kSumGroup][CHUNKS];
  auto fetch = [&](int g) {      // rows g * 8 .. g * 8 + 7 of the tile (k, j order), a row = dim cells, lane l takes cells l, l + 64, ...
#pragma unroll
    for (int q = 0; q < kSumGroup; q++) {
      // (every load is issued, at an address clamped into the tile, and the value discarded where there is no cell: a branch per load made each one a
      // round trip of its own -- 158 us at 256^3)
      const int r = g * kSumGroup + q, rc = min(r, nrows - 1);
      const int base = (j0 + rc % nj) * L.jStride + (k0 + rc / nj) * L.kStride;
#pragma unroll
      for (int c = 0; c < CHUNKS; c++) {
        const int i = lane + 64 * c, ic = min(i, dim - 1);
        const double a = pa[base + ic], bb = pb[base + ic];
        const double w = dot ? a * bb : a;
        v[q][c] = (r < nrows && i < dim) ? w : 0.0;
      }
    }
  };
  fetch(0);
  double acc = 0.0;
  for (int g = 0; g < ngroups; g++) {
    double *buf = sum_rows + (size_t)(g & 1) * kSumGroup * dim;
#pragma unroll
    for (int q = 0; q < kSumGroup; q++)
#pragma unroll
      for (int c = 0; c < CHUNKS; c++) { const int i = lane + 64 * c; if (i < dim) buf[q * dim + i] = v[q][c]; }
    if (g + 1 < ngroups) fetch(g + 1);      // in flight under the chain below
    __syncthreads();                        // (one wave: orders its LDS writes before lane 0's reads; the buffer written next was read two groups ago)
    if (lane == 0) {
      const int rows_here = min(kSumGroup, nrows - g * kSumGroup), n = rows_here * dim;
      const ldsrow rr = (ldsrow)buf;
#pragma unroll 16
      for (int i = 0; i < n; i++) acc += rr[i];      // the chain: k, j, i order
    }
  }
  if (lane == 0) partials[t] = acc;
}
// the partials in tile order: a wave fetches 64 of them at a time, lane 0's chain takes them from the lanes' registers
__global__ __launch_bounds__(64) void ordered_sum_kernel(const double *partials, int n, ResultSlot *result, unsigned long long seq) {
  const int lane = (int)threadIdx.x;
  double s = 0.0;
  double next = (lane < n) ? partials[lane] : 0.0;
  for (int t0 = 0; t0 < n; t0 += 64) {
    const double cur = next;
    if (t0 + 64 < n) next = (t0 + 64 + lane < n) ? partials[t0 + 64 + lane] : 0.0;
    const int m = (n - t0 < 64) ? n - t0 : 64;
    if (m == 64) {
#pragma unroll
      for (int q = 0; q < 64; q++) s += __shfl(cur, q, 64);
    } else {
      for (int q = 0; q < m; q++) s += __shfl(cur, q, 64);
    }
  }
  if (lane == 0) publish(result, s, seq);
}

// ---- operators.7pt.c:158-227 --------------------------------------------------------------
struct RebuildArgs { int variable_coeff, alpha_id, l1inv_id; double a, b, h2inv; };
__global__ __launch_bounds__(256) void rebuild7_kernel(const hpgmg_hip_level L, const RebuildArgs A, double *partials) {
  RowIter r;
  const int row = blockIdx.x * kRowsPerBlock + threadIdx.x / 64, lane = threadIdx.x % 64;
  double best = -1e9;
  if (row_of(row, L.dim, L.num_boxes, r)) {
    const int base = r.j * L.jStride + r.k * L.kStride, jS = L.jStride, kS = L.kStride;
    const double *beta_i = vec_origin(L, r.box, VECTOR_BETA_I) + base;
    const double *beta_j = vec_origin(L, r.box, VECTOR_BETA_J) + base;
    const double *beta_k = vec_origin(L, r.box, VECTOR_BETA_K) + base;
    const double *alpha = (A.alpha_id >= 0) ? vec_origin(L, r.box, A.alpha_id) + base : nullptr;
    double *dinv = vec_origin(L, r.box, VECTOR_DINV) + base;
    double *l1inv = (A.l1inv_id >= 0) ? vec_origin(L, r.box, A.l1inv_id) + base : nullptr;
    const int gi0 = L.box_low[3 * r.box], gj = L.box_low[3 * r.box + 1] + r.j, gk = L.box_low[3 * r.box + 2] + r.k;
    double jlo = 1.0, jhi = 1.0, klo = 1.0, khi = 1.0;
    if (!L.periodic) {
      if (gj - 1 < 0) jlo = 0.0;  if (gj + 1 >= L.dim_j) jhi = 0.0;
      if (gk - 1 < 0) klo = 0.0;  if (gk + 1 >= L.dim_k) khi = 0.0;
    }
    for (int i = lane; i < L.dim; i += 64) {
      double ilo = 1.0, ihi = 1.0;
      if (!L.periodic) { if (gi0 + i - 1 < 0) ilo = 0.0; if (gi0 + i + 1 >= L.dim_i) ihi = 0.0; }
      double sumAbsAij, Aii;
      if (A.variable_coeff) {
        double s = fabs(beta_i[i] * ilo);
        s = s + fabs(beta_j[i] * jlo);
        s = s + fabs(beta_k[i] * klo);
        s = s + fabs(beta_i[i + 1] * ihi);
        s = s + fabs(beta_j[i + jS] * jhi);
        s = s + fabs(beta_k[i + kS] * khi);
        sumAbsAij = fabs(A.b * A.h2inv) * s;
        double d = beta_i[i] * (ilo - 2.0);
        d = d + beta_j[i] * (jlo - 2.0);
        d = d + beta_k[i] * (klo - 2.0);
        d = d + beta_i[i + 1] * (ihi - 2.0);
        d = d + beta_j[i + jS] * (jhi - 2.0);
        d = d + beta_k[i + kS] * (khi - 2.0);
        Aii = ((-A.b) * A.h2inv) * d;
        if (alpha) Aii = Aii + A.a * alpha[i];
      } else {
        double s = ilo + jlo; s = s + klo; s = s + ihi; s = s + jhi; s = s + khi;
        sumAbsAij = fabs(A.b * A.h2inv) * s;
        Aii = A.a - (A.b * A.h2inv) * (s - 12.0);
      }
      dinv[i] = 1.0 / Aii;
      const double Di = (Aii + sumAbsAij) / Aii;
      best = (Di > best) ? Di : best;
      if (l1inv) l1inv[i] = (Aii >= 1.5 * sumAbsAij) ? 1.0 / (Aii) : 1.0 / (Aii + 0.5 * sumAbsAij);
    }
  }
  block_max_store(best, partials);
}

// operators/rebuild.c:139-181: from the accumulated Aii and sum|Aij| form Dinv, L1inv and the Gershgorin bound
__global__ __launch_bounds__(256) void blackbox_finalize_kernel(const hpgmg_hip_level L, int Aii_id, int sum_id, double a, double b, double h2inv, double *partials) {
  RowIter r;
  const int row = blockIdx.x * kRowsPerBlock + threadIdx.x / 64, lane = threadIdx.x % 64;
  double best = -1e9;
  if (row_of(row, L.dim, L.num_boxes, r)) {
    const int base = r.j * L.jStride + r.k * L.kStride;
    double *Aii = vec_origin(L, r.box, Aii_id) + base, *sumAbs = vec_origin(L, r.box, sum_id) + base;
    for (int i = lane; i < L.dim; i += 64) {
      double d = Aii[i];
      if (d == 0.0) d = a + b * h2inv;                       // the reference's "FIX !!!" branch (rebuild.c:160-163)
      const double sa = sumAbs[i];
      const double Di = (d + sa) / d;
      best = (Di > best) ? Di : best;
      sumAbs[i] = (d >= 1.5 * sa) ? 1.0 / (d) : 1.0 / (d + 0.5 * sa);
      Aii[i] = 1.0 / d;
    }
  }
  block_max_store(best, partials);
}

// scratch for partial results + a pinned host word the final kernels write to
static double *g_scratch = nullptr;  static int g_scratch_len = 0;
static ResultSlot *g_result_dev = nullptr;   // pinned, device-visible host slot: kernels publish the scalar here directly
static unsigned long long g_seq = 0;
static int ensure_scratch(int n) {
  if (!g_result_dev) { HPGMG_CHECK(hipHostMalloc((void **)&g_result_dev, 64, hipHostMallocDefault)); for (int q_ = 0; q_ < 4; q_++) { g_result_dev[q_].value = 0.0; g_result_dev[q_].seq = 0; } }      // [0]: the slot (+ a second value behind it); [2]: the slot of a deferred scalar
  if (n > g_scratch_len) {
    if (g_scratch) { hipStreamSynchronize(g_stream); (void)hipFree(g_scratch); }
    int want = n < 65536 ? 65536 : n;
    HPGMG_CHECK(hipMalloc((void **)&g_scratch, (size_t)want * sizeof(double)));
    g_scratch_len = want;
  }
  return 0;
}
static int fetch_result(double *out) {
  volatile ResultSlot *slot = g_result_dev;
  for (long spins = 0; slot->seq != g_seq; spins++) {
    __builtin_ia32_pause();
    if (spins > 20000000L) { HPGMG_CHECK(hipStreamSynchronize(g_stream)); if (slot->seq != g_seq) return record_error(hipErrorUnknown, "reduction result never arrived"); }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  *out = slot->value;
  return 0;
}

// for kernels of other translation units that publish a scalar themselves (stencil.hip: small_ops_kernel)
ResultSlot *reduction_slot_next(unsigned long long *seq_out) { if (ensure_scratch(1)) return nullptr; *seq_out = ++g_seq; return g_result_dev; }
int reduction_fetch(double *out) { return fetch_result(out); }
double reduction_second_value(void) { return reinterpret_cast<volatile double *>(g_result_dev)[2]; }
// for kernels of other translation units that leave one partial maximum per workgroup (stencil.hip: residual + norm fused)
double *reduction_scratch(int n) { return ensure_scratch(n) ? nullptr : g_scratch; }
int finish_max_reduction(int n, double init, double *out) {
  hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(256), 0, g_stream, (const double *)g_scratch, n, init, g_result_dev, ++g_seq);
  HPGMG_LAUNCH_CHECK("final_max_kernel");
  return fetch_result(out);
}

template <int OP>
static int launch_ew(const hpgmg_hip_level *L, const EwArgs &A) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  const int rows = L->num_boxes * L->dim * L->dim;
  hipLaunchKernelGGL((elementwise_kernel<OP>), dim3(rows_grid(rows)), dim3(256), 0, g_stream, *L, A);
  HPGMG_LAUNCH_CHECK("elementwise_kernel");
  return 0;
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {
int hpgmg_hip_graph_flush(void);

int hpgmg_hip_fill(const hpgmg_hip_level *L, int id, double v) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  const int side = L->dim + 2 * L->ghosts, rows = L->num_boxes * side * side;
  hipLaunchKernelGGL(fill_kernel, dim3(rows_grid(rows)), dim3(256), 0, g_stream, *L, id, v);
  HPGMG_LAUNCH_CHECK("fill_kernel");
  return 0;
}
int hpgmg_hip_axpby(const hpgmg_hip_level *L, int id_c, double sa, int id_a, double sb, int id_b) {
  EwArgs A = {}; A.id_c = id_c; A.id_a = id_a; A.id_b = id_b; A.sa = sa; A.sb = sb; return launch_ew<OP_AXPBY>(L, A);
}
int hpgmg_hip_mul(const hpgmg_hip_level *L, int id_c, double s, int id_a, int id_b) {
  EwArgs A = {}; A.id_c = id_c; A.id_a = id_a; A.id_b = id_b; A.sa = s; return launch_ew<OP_MUL>(L, A);
}
int hpgmg_hip_invert(const hpgmg_hip_level *L, int id_c, double s, int id_a) {
  EwArgs A = {}; A.id_c = id_c; A.id_a = id_a; A.sa = s; return launch_ew<OP_INVERT>(L, A);
}
int hpgmg_hip_scale(const hpgmg_hip_level *L, int id_c, double s, int id_a) {
  EwArgs A = {}; A.id_c = id_c; A.id_a = id_a; A.sa = s; return launch_ew<OP_SCALE>(L, A);
}
int hpgmg_hip_shift(const hpgmg_hip_level *L, int id_c, int id_a, double shift) {
  EwArgs A = {}; A.id_c = id_c; A.id_a = id_a; A.sa = shift; return launch_ew<OP_SHIFT>(L, A);
}
int hpgmg_hip_color(const hpgmg_hip_level *L, int id, int colors, int ic, int jc, int kc) {
  EwArgs A = {}; A.id_c = id; A.colors = colors; A.ic = ic; A.jc = jc; A.kc = kc; return launch_ew<OP_COLOR>(L, A);
}
int hpgmg_hip_random(const hpgmg_hip_level *L, int id) {
  EwArgs A = {}; A.id_c = id; return launch_ew<OP_RANDOM>(L, A);
}

// A scalar the caller needs only LATER (norm(F) at the start of FMGSolve is used in the convergence check at its end, mg.c:1262,1323) goes to a slot of
// its own, so the host does not wait for the pass that forms it and keeps the stream full behind it; hpgmg_hip_deferred_fetch() collects it.
static unsigned long long g_seq_deferred = 0;
static int norm_copy_restrict(const hpgmg_hip_level *L, int f_id, int r_id, const hpgmg_hip_level *Lc, int rc_id, const int *map, double *norm_out, bool deferred);
int hpgmg_hip_norm_copy_restrict_deferred(const hpgmg_hip_level *L, int f_id, int r_id, const hpgmg_hip_level *Lc, int rc_id, const int *map) {
  double unused = 0.0;
  return norm_copy_restrict(L, f_id, r_id, Lc, rc_id, map, &unused, true);
}
int hpgmg_hip_deferred_fetch(double *out) {
  if (!g_result_dev || g_seq_deferred == 0) return record_error(hipErrorInvalidValue, "deferred_fetch: nothing was deferred");
  volatile ResultSlot *slot = g_result_dev + 2;
  for (long spins = 0; slot->seq != g_seq_deferred; spins++) {
    __builtin_ia32_pause();
    if (spins > 20000000L) { HPGMG_CHECK(hipStreamSynchronize(g_stream)); if (slot->seq != g_seq_deferred) return record_error(hipErrorUnknown, "deferred reduction result never arrived"); }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  *out = slot->value;
  return 0;
}
int hpgmg_hip_norm_copy_restrict(const hpgmg_hip_level *L, int f_id, int r_id, const hpgmg_hip_level *Lc, int rc_id, const int *map, double *norm_out) {
  return norm_copy_restrict(L, f_id, r_id, Lc, rc_id, map, norm_out, false);
}
static int norm_copy_restrict(const hpgmg_hip_level *L, int f_id, int r_id, const hpgmg_hip_level *Lc, int rc_id, const int *map, double *norm_out, bool deferred) {
  if (int e = hpgmg_hip_graph_flush()) return e;
  if (norm_out) *norm_out = 0.0;
  if (L->num_boxes <= 0 || Lc->num_boxes <= 0 || (L->dim & 1) || !(L->flags & 1) || (L->jStride & 1) || (L->kStride & 1) || (L->volume & 1) || f_id == r_id)
    return record_error(hipErrorInvalidValue, "norm_copy_restrict: level not supported");
  const int nblk = rows_grid(L->num_boxes * (L->dim / 2) * (L->dim / 2));
  if (int e = ensure_scratch(nblk)) return e;
  static const bool nt_allowed = [] { const char *e = getenv("HPGMG_TUNE_COPY_NT"); return !(e && *e == '0'); }();
  if (nt_allowed && (double)L->num_boxes * L->dim * L->dim * L->dim * sizeof(double) > 256e6)
    hipLaunchKernelGGL(norm_copy_restrict_kernel<true>, dim3(nblk), dim3(256), 0, g_stream, *L, f_id, r_id, *Lc, rc_id, map, g_scratch);
  else hipLaunchKernelGGL(norm_copy_restrict_kernel<false>, dim3(nblk), dim3(256), 0, g_stream, *L, f_id, r_id, *Lc, rc_id, map, g_scratch);
  if (!norm_out) { HPGMG_LAUNCH_CHECK("norm_copy_restrict (copy + restriction only)"); return 0; }     // the norm is not wanted: nothing to reduce, nothing to wait for
  if (deferred) {
    hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(256), 0, g_stream, (const double *)g_scratch, nblk, 0.0, g_result_dev + 2, ++g_seq_deferred);
    HPGMG_LAUNCH_CHECK("norm_copy_restrict (norm deferred)");
    return 0;
  }
  hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(256), 0, g_stream, (const double *)g_scratch, nblk, 0.0, g_result_dev, ++g_seq);
  HPGMG_LAUNCH_CHECK("norm_copy_restrict");
  return fetch_result(norm_out);
}
int hpgmg_hip_norm_max(const hpgmg_hip_level *L, int id, double *out) {
  if (int e = hpgmg_hip_graph_flush()) return e;
  *out = 0.0;
  if (L->num_boxes <= 0) return 0;
  const int nblk = rows_grid(L->num_boxes * L->dim * L->dim);
  if (int e = ensure_scratch(nblk)) return e;
  if (nblk <= 64) {
    hipLaunchKernelGGL(absmax_small_kernel, dim3(1), dim3(256), 0, g_stream, *L, id, g_result_dev, ++g_seq);
  } else {
    // ~16 workgroups per CU over the whole level, each wave streaming several rows of one box
    int per_box = rows_grid(L->dim * L->dim);
    while (per_box > 1 && (long long)per_box * L->num_boxes > 4096) per_box = (per_box + 1) / 2;
    hipLaunchKernelGGL(absmax_kernel, dim3(per_box, L->num_boxes), dim3(256), 0, g_stream, *L, id, g_scratch);
    hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(256), 0, g_stream, (const double *)g_scratch, per_box * L->num_boxes, 0.0, g_result_dev, ++g_seq);
  }
  HPGMG_LAUNCH_CHECK("norm_max");
  return fetch_result(out);
}
static int ordered_sum(const hpgmg_hip_level *L, int id_a, int id_b, double *out) {
  if (int e = hpgmg_hip_graph_flush()) return e;
  *out = 0.0;
  if (L->num_boxes <= 0) return 0;
  const int side = (L->dim + BLOCKCOPY_TILE_J - 1) / BLOCKCOPY_TILE_J, ntiles = side * side * L->num_boxes;
  if (int e = ensure_scratch(ntiles)) return e;
  if (ntiles <= 64) {
    hipLaunchKernelGGL((tile_sum_kernel<true>), dim3(1), dim3(64), 0, g_stream, *L, id_a, id_b, g_scratch, g_result_dev, ++g_seq);
  } else {
    const size_t lds = (size_t)2 * kSumGroup * L->dim * sizeof(double);
    if (L->dim <= 128)      hipLaunchKernelGGL((tile_sum_wave_kernel<2>), dim3(ntiles), dim3(64), lds, g_stream, *L, id_a, id_b, g_scratch);
    else if (L->dim <= 256) hipLaunchKernelGGL((tile_sum_wave_kernel<4>), dim3(ntiles), dim3(64), lds, g_stream, *L, id_a, id_b, g_scratch);
    else if (L->dim <= 64 * kSumMaxChunks) {
      static bool once = false;
      if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)tile_sum_wave_kernel<kSumMaxChunks>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)2 * kSumGroup * 64 * kSumMaxChunks * sizeof(double)))); once = true; }
      hipLaunchKernelGGL((tile_sum_wave_kernel<kSumMaxChunks>), dim3(ntiles), dim3(64), lds, g_stream, *L, id_a, id_b, g_scratch);
    }
    else hipLaunchKernelGGL((tile_sum_kernel<false>), dim3((ntiles + 63) / 64), dim3(64), 0, g_stream, *L, id_a, id_b, g_scratch, g_result_dev, 0ULL);
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(64), 0, g_stream, (const double *)g_scratch, ntiles, g_result_dev, ++g_seq);
  }
  HPGMG_LAUNCH_CHECK("ordered_sum");
  return fetch_result(out);
}
int hpgmg_hip_dot(const hpgmg_hip_level *L, int id_a, int id_b, double *out) { return ordered_sum(L, id_a, id_b, out); }
int hpgmg_hip_sum(const hpgmg_hip_level *L, int id, double *out) { return ordered_sum(L, id, -1, out); }

int hpgmg_hip_blackbox_finalize(const hpgmg_hip_level *L, int Aii_id, int sumAbs_id, double a, double b, double h2inv, double *lambda_max_out) {
  if (int e = hpgmg_hip_graph_flush()) return e;
  *lambda_max_out = -1e9;
  if (L->num_boxes <= 0) return 0;
  const int nblk = rows_grid(L->num_boxes * L->dim * L->dim);
  if (int e = ensure_scratch(nblk)) return e;
  hipLaunchKernelGGL(blackbox_finalize_kernel, dim3(nblk), dim3(256), 0, g_stream, *L, Aii_id, sumAbs_id, a, b, h2inv, g_scratch);
  hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(256), 0, g_stream, (const double *)g_scratch, nblk, -1e9, g_result_dev, ++g_seq);
  HPGMG_LAUNCH_CHECK("blackbox_finalize");
  return fetch_result(lambda_max_out);
}

int hpgmg_hip_rebuild_7pt(const hpgmg_hip_level *L, int variable_coeff, int alpha_id, int l1inv_id,
                          double a, double b, double h2inv, double *lambda_max_out) {
  if (int e = hpgmg_hip_graph_flush()) return e;
  *lambda_max_out = -1e9;
  if (L->num_boxes <= 0) return 0;
  const int nblk = rows_grid(L->num_boxes * L->dim * L->dim);
  if (int e = ensure_scratch(nblk)) return e;
  RebuildArgs A = { variable_coeff, alpha_id, l1inv_id, a, b, h2inv };
  hipLaunchKernelGGL(rebuild7_kernel, dim3(nblk), dim3(256), 0, g_stream, *L, A, g_scratch);
  hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(256), 0, g_stream, (const double *)g_scratch, nblk, -1e9, g_result_dev, ++g_seq);
  HPGMG_LAUNCH_CHECK("rebuild_7pt");
  return fetch_result(lambda_max_out);
}

}  // extern "C"

// stencil.hip -- the hot path: Chebyshev / GSRB / Jacobi smoother sweeps and the
// residual / apply_op stencil of the 7-point operator, written for gfx950.
//
// Reference semantics (paths relative to finite-volume/source/):
//   apply_op_ijk      operators.7pt.c:49-89 (VC Helmholtz :51-62, VC Poisson :64-74, CC :77-88)
//   Chebyshev update  operators/chebyshev.c:86-95      GSRB update operators/gsrb.c:90-105
//   Jacobi update     operators/jacobi.c:53-60         residual    operators/residual.c:42-48
// Every floating-point expression keeps the reference's association and the
// file is compiled with -ffp-contract=off, so results are bit-identical to the
// reference's gcc -O2 build (no FMA, no reassociation): see DESIGN.md "parity".
//
// Bound: HBM bandwidth (about 25 flop per 64-72 B per cell).  Kernels in this file:
//   * stencil7_kernel (any box size; the mid-size, cache-resident levels): one lane per (i,j) column, 64 lanes
//     along the unit-stride i direction, the block marches in +k and keeps x[k-1], x[k], x[k+1] and beta_k[k],
//     beta_k[k+1] in registers; the +-i / +-j neighbours of x are re-read through the vector L1;
//   * stencil7_wide_kernel (boxes of side 128 m; the bandwidth-bound fine level): 2 x 2 cells per lane with
//     16-byte loads, j neighbours through an LDS copy of the plane, i neighbours by wave shuffle -- every cell
//     of every stream is fetched from memory once per workgroup;
//   * (cheby_pair_kernel, two sweeps per pass on the fine level: cheby_pair.hpp, launched from pair.hip; the single-workgroup kernels of
//     the small levels: small_levels.hip; shared declarations: stencil_direct.hpp)
//   * stencil7_shell_kernel: the cells next to faces owned by another rank, after the overlapped exchange;
//   * stencil27_kernel (27-point) and stencil_direct_kernel (4th-order fv4, black-box probes);
//   * logical tiles are ordered box, k, j, i and dealt to XCDs in contiguous ranges (common.hpp) so halo
//     planes shared by adjacent tiles hit the same L2.
#include "stencil_direct.hpp"
#include "stencil27_rb.hpp"
#include "stencil27_rb_box.hpp"
#include "stencil7_tile.hpp"

namespace hpgmg {

// RR (MODE_RESIDUAL on the launch-bound levels, boxes of side <= 32): residual + restriction + zero_vector of MGVCycle's down leg (mg.c:1150-1153) in
// ONE launch -- a wave holds whole PAIRS of rows (blockDim.x <= 32), so the lane of an even (i, j) gathers its 2 x 2 patch of residuals from its
// neighbours' registers, sums it in restriction.c:54-57's order and carries the sum over the plane pair; the coarse zero_vector rides along as
// extra workgroups; the residual itself is stored only when asked for (FA.store_res: the exact state of the three operators).
template <int V, int MODE, bool IP = false, bool RR = false>
__global__ __launch_bounds__(256) void stencil7_kernel(const hpgmg_hip_level L, const StencilArgs P, const InterpFold F = InterpFold{}, const FusedArgs FA = FusedArgs{}) {
  if (RR && (int)blockIdx.x >= kXcds * P.per_xcd) {      // zero_vector(coarse, zero_id): whole padded boxes, ghosts included
    const int z = (int)blockIdx.x - kXcds * P.per_xcd, zbox = z / FA.zero_chunks_per_box, chunk = z - zbox * FA.zero_chunks_per_box;
    if (zbox >= FA.Lc.num_boxes) return;
    double *v = FA.Lc.box_base[zbox] + (size_t)FA.zero_id * (size_t)FA.Lc.volume;
    const int lo = chunk * 4096, hi = (lo + 4096 < FA.Lc.volume) ? lo + 4096 : FA.Lc.volume, nth = (int)(blockDim.x * blockDim.y);
    for (int q = lo + (int)(threadIdx.y * blockDim.x + threadIdx.x); q < hi; q += nth) v[q] = 0.0;
    return;
  }
  const int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (logical >= P.total_blocks) return;
  // logical id -> (box, k chunk, j tile, i tile), i fastest
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;

  const int i = ti * (int)blockDim.x + (int)threadIdx.x;
  const int j = tj * (int)blockDim.y + (int)threadIdx.y;
  if (i >= L.dim || j >= L.dim) return;
  const int k0 = ck * P.kchunk;
  const int k1 = (k0 + P.kchunk < L.dim) ? k0 + P.kchunk : L.dim;
  const int jS = L.jStride, kS = L.kStride;

  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  constexpr bool kSmooth = (MODE == MODE_CHEBY || MODE == MODE_GSRB || MODE == MODE_JACOBI);

  typename src_ptr<MODE == MODE_GSRB>::type x = vec_origin(L, box, P.xn_id);
  double *out = vec_origin(L, box, P.xout_id);   // may alias x (in-place GSRB): no __restrict__
  const double *__restrict__ rhs = (MODE == MODE_APPLY) ? nullptr : vec_origin(L, box, P.rhs_id);
  const double *__restrict__ dinv = kSmooth ? vec_origin(L, box, VECTOR_DINV) : nullptr;
  const double *__restrict__ alpha = kHelm ? vec_origin(L, box, VECTOR_ALPHA) : nullptr;
  const double *__restrict__ beta_i = kVC ? vec_origin(L, box, VECTOR_BETA_I) : nullptr;
  const double *__restrict__ beta_j = kVC ? vec_origin(L, box, VECTOR_BETA_J) : nullptr;
  const double *__restrict__ beta_k = kVC ? vec_origin(L, box, VECTOR_BETA_K) : nullptr;

  int colour000 = 0;
  if (MODE == MODE_GSRB) colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;

  // Value of x just outside this box across face `dir` (0..5 = -i,+i,-j,+j,-k,+k): from the
  // neighbouring local box, from the Dirichlet condition (ghost = -centre), or from the ghost zone.
  const int last = L.dim - 1;
  auto outside = [&](int dir, int idx_in_neighbour, int idx_ghost, double centre) -> double {
    const int nb = L.box_nbr[6 * box + dir];
    if (nb >= 0) return vec_origin(L, nb, P.xn_id)[idx_in_neighbour];
    if (nb == -1) return -centre;
    return x[idx_ghost];
  };
  const bool gf = P.ghost_free != 0;
  // cells whose stencil reaches into a ghost zone another rank fills (box_nbr == -2)
  bool defer_ij = false, defer_klo = false, defer_khi = false;
  if (P.defer) {
    const int *nb = L.box_nbr + 6 * box;
    defer_ij = (i == 0 && nb[0] == -2) || (i == last && nb[1] == -2) || (j == 0 && nb[2] == -2) || (j == last && nb[3] == -2);
    defer_klo = (nb[4] == -2); defer_khi = (nb[5] == -2);
  }

  // IP, sweep 0 of a smooth() with the interpolation folded in: cell (ii, jj, kk) of box bx as stored + the coarse value above it; one step across a
  // face of the box lands in the neighbouring box (every box local) or on the Dirichlet rule applied to the folded centre
  const bool fold_x = IP && F.which == 1;
  auto folded = [&](int bx, int ii, int jj, int kk) -> double { return vec_origin(L, bx, P.xn_id)[ii + jj * jS + kk * kS] + fold_coarse(F, bx, ii, jj, kk); };
  auto folded_step = [&](int dir, int ii, int jj, int kk, double centre) -> double {      // (ii, jj, kk): one step outside this box across face dir
    const int nb = L.box_nbr[6 * box + dir];
    if (nb < 0) return -centre;
    if (dir == 0) ii = last; else if (dir == 1) ii = 0; else if (dir == 2) jj = last; else if (dir == 3) jj = 0; else if (dir == 4) kk = last; else kk = 0;
    return folded(nb, ii, jj, kk);
  };
  double *coarse = nullptr; double racc = 0.0;
  if (RR) {
    const int *m = FA.map + 4 * box;
    coarse = vec_origin(FA.Lc, m[0], FA.coarse_id) + (m[1] + (i >> 1)) + (m[2] + (j >> 1)) * FA.Lc.jStride + (m[3] + (k0 >> 1)) * FA.Lc.kStride;
  }
  int ijk = i + j * jS + k0 * kS;
  double xc = fold_x ? folded(box, i, j, k0) : x[ijk];
  double xm = fold_x ? (k0 == 0 ? folded_step(4, i, j, -1, xc) : folded(box, i, j, k0 - 1))
                     : ((gf && k0 == 0) ? outside(4, i + j * jS + last * kS, ijk - kS, xc) : x[ijk - kS]);
  double bk0 = kVC ? beta_k[ijk] : 0.0;

  for (int k = k0; k < k1; k++, ijk += kS) {
    const double xp = fold_x ? (k == last ? folded_step(5, i, j, k + 1, xc) : folded(box, i, j, k + 1))
                             : ((gf && k == last) ? outside(5, i + j * jS, ijk + kS, xc) : x[ijk + kS]);
    const double bk1 = kVC ? beta_k[ijk + kS] : 0.0;
    bool update = true;
    if (MODE == MODE_GSRB) update = (((i ^ j ^ k ^ colour000) & 1) == 0);
    if (P.defer && (defer_ij || (k == 0 && defer_klo) || (k == last && defer_khi))) update = false;
    if (update) {
      double xim, xip, xjm, xjp;
      if (fold_x) {
        xim = (i == 0)    ? folded_step(0, i - 1, j, k, xc) : folded(box, i - 1, j, k);
        xip = (i == last) ? folded_step(1, i + 1, j, k, xc) : folded(box, i + 1, j, k);
        xjm = (j == 0)    ? folded_step(2, i, j - 1, k, xc) : folded(box, i, j - 1, k);
        xjp = (j == last) ? folded_step(3, i, j + 1, k, xc) : folded(box, i, j + 1, k);
      } else {
        xim = (gf && i == 0)    ? outside(0, last + j * jS + k * kS, ijk - 1, xc)  : x[ijk - 1];
        xip = (gf && i == last) ? outside(1, j * jS + k * kS, ijk + 1, xc)          : x[ijk + 1];
        xjm = (gf && j == 0)    ? outside(2, i + last * jS + k * kS, ijk - jS, xc) : x[ijk - jS];
        xjp = (gf && j == last) ? outside(3, i + k * kS, ijk + jS, xc)              : x[ijk + jS];
      }
      double bi0 = 0.0, bi1 = 0.0, bj0 = 0.0, bj1 = 0.0, al = 0.0;
      if (kVC) { bi0 = beta_i[ijk]; bi1 = beta_i[ijk + 1]; bj0 = beta_j[ijk]; bj1 = beta_j[ijk + jS]; }
      if (kHelm) al = alpha[ijk];
      const double Ax = apply_op_7pt<V>(xc, xim, xip, xjm, xjp, xm, xp, bi0, bi1, bj0, bj1, bk0, bk1, al, P.a, P.b, P.h2inv);
      if (MODE == MODE_CHEBY) {
        double xnm1 = out[ijk];
        if (IP && F.which == 2) xnm1 = xnm1 + fold_coarse(F, box, i, j, k);
        out[ijk] = xc + P.c1 * (xc - xnm1) + P.c2 * dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_GSRB) {
        out[ijk] = xc + dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_JACOBI) {
        out[ijk] = xc + P.c2 * dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_RESIDUAL) {
        const double r = rhs[ijk] - Ax;
        if (!RR || FA.store_res) out[ijk] = r;
        if (RR) {      // the children (i, i+1) of rows j, j+1 of plane k, then of plane k+1; times 0.125 (restriction.c:54-57)
          const int w = (int)blockDim.x;
          const double r10 = __shfl_down(r, 1, 64), r01 = __shfl_down(r, w, 64), r11 = __shfl_down(r, w + 1, 64);
          if (((k - k0) & 1) == 0) { racc = r + r10; racc = racc + r01; racc = racc + r11; }
          else {
            racc = racc + r; racc = racc + r10; racc = racc + r01; racc = racc + r11;
            if (((i | j) & 1) == 0) coarse[((k - k0) >> 1) * FA.Lc.kStride] = racc * 0.125;
          }
        }
      } else {
        out[ijk] = Ax;
      }
    } else if (P.copy_other_colour) {
      out[ijk] = xc;
    }
    xm = xc; xc = xp; bk0 = bk1;
  }
}


// ---------------------------------------------------------------------------------------------
// Wide kernel for boxes whose side is a multiple of 128 (the bandwidth-critical fine levels).
// Each lane owns a 2 (i) x 2 (j) patch: a wave covers two full 128-cell rows with 16-byte loads
// (1 KiB per wave-instruction, the coalescing sweet spot); a 64 x WJ workgroup covers 2*WJ rows
// and marches in +k.  Every cell of x is fetched from memory ONCE per workgroup:
//   * k neighbours: the plane k+1 loaded this step is the centre of the next (registers);
//   * j neighbours: the rows owned by the waves above / below come from an LDS copy of the plane
//     (each wave deposits the plane k+1 rows it just loaded; double buffered, one barrier per step);
//     only the two rows outside the workgroup's slab are read from memory;
//   * i neighbours: the neighbouring lane's registers (wave shuffle); lanes 0 / 63 sit on the tile edge.
// (Measured on MI355X: re-reading those neighbours through L1/L2, as the first version did, made the
// L2 miss traffic 10.1 streams per cell instead of 8.4 -- 32 KiB of L1 and 4 MiB of L2 per XCD do not
// hold a plane step of 128 resident workgroups.)
// Interior rows are 16-byte aligned because jStride and kStride are even and the first interior cell
// is 32-byte aligned (create_vectors); the launcher checks this and otherwise uses the generic kernel.
// GSRB: the two cells of a row always have different colours, so every lane updates exactly one
// cell per row (no idle lanes) and stores 8 bytes.
struct alignas(16) d2 { double x, y; };
__device__ __forceinline__ d2 ld2(const double *p) { return *reinterpret_cast<const d2 *>(p); }
__device__ __forceinline__ void st2(double *p, d2 v) { *reinterpret_cast<d2 *>(p) = v; }

template <int V, int MODE, int WJ>
__global__ __launch_bounds__(64 * WJ) void stencil7_wide_kernel(const hpgmg_hip_level L, const StencilArgs P, const FusedArgs F) {
  __shared__ d2 slab[2][2 * WJ][64];                          // plane copy: [buffer][row of the slab][i pair]
  __shared__ double wg_max[WJ];
  // zero_vector(coarse, zero_id) -- whole padded boxes, ghosts included -- by the workgroups appended after the stencil grid (physical
  // order, so they spread over all XCDs instead of landing on the last ones)
  if (MODE == MODE_RESIDUAL_RESTRICT && (int)blockIdx.x >= kXcds * P.per_xcd) {
    const int z = (int)blockIdx.x - kXcds * P.per_xcd, zbox = z / F.zero_chunks_per_box, chunk = z - zbox * F.zero_chunks_per_box;
    if (zbox >= F.Lc.num_boxes) return;
    double *v = F.Lc.box_base[zbox] + (size_t)F.zero_id * (size_t)F.Lc.volume;
    const int lo = chunk * 4096, hi = (lo + 4096 < F.Lc.volume) ? lo + 4096 : F.Lc.volume;
    for (int q = lo + (int)(threadIdx.y * 64 + threadIdx.x); q < hi; q += 64 * WJ) v[q] = 0.0;
    return;
  }
  int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (P.order) logical = P.order[logical];
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;

  const int lane = (int)threadIdx.x, ty = (int)threadIdx.y;
  const int i = ti * 128 + 2 * lane;                         // cells i, i+1
  const int ja = tj * (2 * WJ) + 2 * ty;                     // rows ja, ja+1
  const int k0 = ck * P.kchunk, k1 = (k0 + P.kchunk < L.dim) ? k0 + P.kchunk : L.dim;
  const int jS = L.jStride, kS = L.kStride, last = L.dim - 1;

  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  constexpr bool kSmooth = (MODE == MODE_CHEBY || MODE == MODE_GSRB || MODE == MODE_JACOBI);
  constexpr bool kRestrict = (MODE == MODE_RESIDUAL_RESTRICT), kNorm = (MODE == MODE_RESIDUAL_NORM);

  typename src_ptr<MODE == MODE_GSRB>::type x = vec_origin(L, box, P.xn_id);
  double *out = ((kRestrict && !F.store_res) || (kNorm && F.no_store)) ? nullptr : vec_origin(L, box, P.xout_id);
  const double *__restrict__ rhs = (MODE == MODE_APPLY) ? nullptr : vec_origin(L, box, P.rhs_id);
  const double *__restrict__ dinv = kSmooth ? vec_origin(L, box, VECTOR_DINV) : nullptr;
  const double *__restrict__ alpha = kHelm ? vec_origin(L, box, VECTOR_ALPHA) : nullptr;
  const double *__restrict__ beta_i = kVC ? vec_origin(L, box, VECTOR_BETA_I) : nullptr;
  const double *__restrict__ beta_j = kVC ? vec_origin(L, box, VECTOR_BETA_J) : nullptr;
  const double *__restrict__ beta_k = kVC ? vec_origin(L, box, VECTOR_BETA_K) : nullptr;
  // fused forms: where this lane's 2 x 2 (x 2 planes) patch lands in the coarse level; running sum / maximum
  double *coarse = nullptr; double racc = 0.0, lane_max = 0.0;
  if (kRestrict) {
    const int *m = F.map + 4 * box;
    coarse = vec_origin(F.Lc, m[0], F.coarse_id) + (m[1] + (i >> 1)) + (m[2] + (ja >> 1)) * F.Lc.jStride + (m[3] + (k0 >> 1)) * F.Lc.kStride;
  }

  const bool gf = P.ghost_free != 0;
  // pair of values just outside this box across face `dir`, for the pair whose centres are `c`
  auto outside2 = [&](int dir, int idx_in_neighbour, int idx_ghost, d2 c) -> d2 {
    const int nb = L.box_nbr[6 * box + dir];
    if (nb >= 0) return ld2(vec_origin(L, nb, P.xn_id) + idx_in_neighbour);
    if (nb == -1) return d2{-c.x, -c.y};
    return ld2(x + idx_ghost);
  };
  auto outside1 = [&](int dir, int idx_in_neighbour, int idx_ghost, double c) -> double {
    const int nb = L.box_nbr[6 * box + dir];
    if (nb >= 0) return vec_origin(L, nb, P.xn_id)[idx_in_neighbour];
    if (nb == -1) return -c;
    return x[idx_ghost];
  };

  int colour000 = 0;
  if (MODE == MODE_GSRB) colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;

  // cells next to a face another rank owns (P.defer): bit 0 = the pair's first cell, bit 1 = its second cell
  int dm_a = 0, dm_b = 0; bool defer_klo = false, defer_khi = false;
  if (P.defer) {
    const int *nb = L.box_nbr + 6 * box;
    const int di = ((i == 0 && nb[0] == -2) ? 1 : 0) | ((i + 1 == last && nb[1] == -2) ? 2 : 0);
    dm_a = (ja == 0 && nb[2] == -2) ? 3 : di;
    dm_b = (ja + 1 == last && nb[3] == -2) ? 3 : di;
    defer_klo = (nb[4] == -2); defer_khi = (nb[5] == -2);
  }

  int ia = i + ja * jS + k0 * kS;          // index of (i, ja, k); row b is ia + jS
  d2 xc_a = ld2(x + ia), xc_b = ld2(x + ia + jS);
  d2 xm_a = (gf && k0 == 0) ? outside2(4, i + ja * jS + last * kS, ia - kS, xc_a) : ld2(x + ia - kS);
  d2 xm_b = (gf && k0 == 0) ? outside2(4, i + (ja + 1) * jS + last * kS, ia + jS - kS, xc_b) : ld2(x + ia + jS - kS);
  d2 bk0_a = {0, 0}, bk0_b = {0, 0};
  if (kVC) { bk0_a = ld2(beta_k + ia); bk0_b = ld2(beta_k + ia + jS); }
  slab[k0 & 1][2 * ty][lane] = xc_a;
  slab[k0 & 1][2 * ty + 1][lane] = xc_b;
  __syncthreads();

  for (int k = k0; k < k1; k++, ia += kS) {
    const int ib = ia + jS;
    const d2 xp_a = (gf && k == last) ? outside2(5, i + ja * jS, ia + kS, xc_a) : ld2(x + ia + kS);
    const d2 xp_b = (gf && k == last) ? outside2(5, i + (ja + 1) * jS, ib + kS, xc_b) : ld2(x + ib + kS);
    d2 bk1_a = {0, 0}, bk1_b = {0, 0};
    if (kVC) { bk1_a = ld2(beta_k + ia + kS); bk1_b = ld2(beta_k + ib + kS); }
    // rows above and below the 2-row patch: the neighbouring wave's rows (LDS) or, at the slab edge, memory
    d2 xjm_a, xjp_b;
    if (ty > 0) xjm_a = slab[k & 1][2 * ty - 1][lane];
    else        xjm_a = (gf && ja == 0) ? outside2(2, i + last * jS + k * kS, ia - jS, xc_a) : ld2(x + ia - jS);
    if (ty < WJ - 1) xjp_b = slab[k & 1][2 * ty + 2][lane];
    else             xjp_b = (gf && ja + 1 == last) ? outside2(3, i + k * kS, ib + jS, xc_b) : ld2(x + ib + jS);
    // i neighbours of the pair: the cell left of .x and the cell right of .y live in the adjacent lanes
    double xl_a = __shfl_up(xc_a.y, 1, 64), xl_b = __shfl_up(xc_b.y, 1, 64);
    double xr_a = __shfl_down(xc_a.x, 1, 64), xr_b = __shfl_down(xc_b.x, 1, 64);
    if (lane == 0) {
      xl_a = (gf && i == 0) ? outside1(0, last + ja * jS + k * kS, ia - 1, xc_a.x)       : x[ia - 1];
      xl_b = (gf && i == 0) ? outside1(0, last + (ja + 1) * jS + k * kS, ib - 1, xc_b.x) : x[ib - 1];
    }
    if (lane == 63) {
      xr_a = (gf && i + 1 == last) ? outside1(1, ja * jS + k * kS, ia + 2, xc_a.y)       : x[ia + 2];
      xr_b = (gf && i + 1 == last) ? outside1(1, (ja + 1) * jS + k * kS, ib + 2, xc_b.y) : x[ib + 2];
    }
    // beta_j on the face between the two rows serves both (upper face of row a, lower face of row b)
    d2 bj_lo = {0, 0}, bj_mid = {0, 0}, bj_hi = {0, 0};
    if (kVC) { bj_lo = ld2(beta_j + ia); bj_mid = ld2(beta_j + ib); bj_hi = ld2(beta_j + ib + jS); }

    const bool defer_plane = P.defer && ((k == 0 && defer_klo) || (k == last && defer_khi));
#define HPGMG_ROW(IDX, XC, XM, XP, XJM, XJP, XL, XR, BK0, BK1, BJL, BJH, JROW, DM)                                          \
    {                                                                                                                        \
      const int dm = defer_plane ? 3 : DM;                                                                                   \
      d2 bi = {0, 0}, al = {0, 0}; double bir = 0;                                                                           \
      if (kVC) { bi = ld2(beta_i + IDX); bir = __shfl_down(bi.x, 1, 64); if (lane == 63) bir = beta_i[IDX + 2]; }            \
      if (kHelm) al = ld2(alpha + IDX);                                                                                     \
      if (MODE == MODE_GSRB) {                                                                                              \
        const int pp = (JROW ^ k ^ colour000) & 1;              /* the cell of this pair whose colour is swept */           \
        const double c = pp ? XC.y : XC.x;                                                                                  \
        const double Ax = apply_op_7pt<V>(c, pp ? XC.x : XL, pp ? XR : XC.y, pp ? XJM.y : XJM.x, pp ? XJP.y : XJP.x,        \
                                          pp ? XM.y : XM.x, pp ? XP.y : XP.x, pp ? bi.y : bi.x, pp ? bir : bi.y,            \
                                          pp ? BJL.y : BJL.x, pp ? BJH.y : BJH.x, pp ? BK0.y : BK0.x, pp ? BK1.y : BK1.x,   \
                                          pp ? al.y : al.x, P.a, P.b, P.h2inv);                                             \
        const d2 r2 = ld2(rhs + IDX), dv = ld2(dinv + IDX);                                                                 \
        const double xn = c + (pp ? dv.y : dv.x) * ((pp ? r2.y : r2.x) - Ax);                                              \
        if (P.copy_other_colour) st2(out + IDX, pp ? d2{XC.x, xn} : d2{xn, XC.y}); else if (!((dm >> pp) & 1)) out[IDX + pp] = xn; \
      } else {                                                                                                              \
        const double Ax0 = apply_op_7pt<V>(XC.x, XL, XC.y, XJM.x, XJP.x, XM.x, XP.x, bi.x, bi.y, BJL.x, BJH.x, BK0.x, BK1.x, al.x, P.a, P.b, P.h2inv); \
        const double Ax1 = apply_op_7pt<V>(XC.y, XC.x, XR, XJM.y, XJP.y, XM.y, XP.y, bi.y, bir, BJL.y, BJH.y, BK0.y, BK1.y, al.y, P.a, P.b, P.h2inv);  \
        d2 o;                                                                                                               \
        if (MODE == MODE_APPLY) { o.x = Ax0; o.y = Ax1; }                                                                   \
        else {                                                                                                              \
          const d2 r2 = ld2(rhs + IDX);                                                                                     \
          if (MODE == MODE_RESIDUAL || kRestrict || kNorm) { o.x = r2.x - Ax0; o.y = r2.y - Ax1; }                          \
          else {                                                                                                            \
            const d2 dv = ld2(dinv + IDX);                                                                                  \
            if (MODE == MODE_CHEBY) {                                                                                       \
              const d2 old = ld2(out + IDX);                                                                                \
              o.x = XC.x + P.c1 * (XC.x - old.x) + P.c2 * dv.x * (r2.x - Ax0);                                              \
              o.y = XC.y + P.c1 * (XC.y - old.y) + P.c2 * dv.y * (r2.y - Ax1);                                              \
            } else {                                                                                                        \
              o.x = XC.x + P.c2 * dv.x * (r2.x - Ax0);                                                                      \
              o.y = XC.y + P.c2 * dv.y * (r2.y - Ax1);                                                                      \
            }                                                                                                               \
          }                                                                                                                 \
        }                                                                                                                   \
        RES = o;                                                                                                            \
        if ((kRestrict && !F.store_res) || (kNorm && F.no_store)) { }                                                       \
        else if (dm == 0) st2(out + IDX, o);                                                                                \
        else { if (!(dm & 1)) out[IDX] = o.x; if (!(dm & 2)) out[IDX + 1] = o.y; }                                          \
      }                                                                                                                     \
    }
    d2 res_a = {0, 0}, res_b = {0, 0};
#define RES res_a
    HPGMG_ROW(ia, xc_a, xm_a, xp_a, xjm_a, xc_b, xl_a, xr_a, bk0_a, bk1_a, bj_lo, bj_mid, ja, dm_a)
#undef RES
#define RES res_b
    HPGMG_ROW(ib, xc_b, xm_b, xp_b, xc_a, xjp_b, xl_b, xr_b, bk0_b, bk1_b, bj_mid, bj_hi, (ja + 1), dm_b)
#undef RES
#undef HPGMG_ROW
    if (kRestrict) {      // restriction.c:54-57: the eight children in the order (i, i+1) of rows j, j+1 of plane k, then of plane k+1; times 0.125
      if (((k - k0) & 1) == 0) { racc = res_a.x + res_a.y; racc = racc + res_b.x; racc = racc + res_b.y; }
      else {
        racc = racc + res_a.x; racc = racc + res_a.y; racc = racc + res_b.x; racc = racc + res_b.y;
        coarse[((k - k0) >> 1) * F.Lc.kStride] = racc * 0.125;
      }
    }
    if (kNorm) {
      double f;
      f = fabs(res_a.x); lane_max = (f > lane_max) ? f : lane_max;  f = fabs(res_a.y); lane_max = (f > lane_max) ? f : lane_max;
      f = fabs(res_b.x); lane_max = (f > lane_max) ? f : lane_max;  f = fabs(res_b.y); lane_max = (f > lane_max) ? f : lane_max;
    }
    // hand the plane k+1 rows to the neighbouring waves for the next step
    slab[(k + 1) & 1][2 * ty][lane] = xp_a;
    slab[(k + 1) & 1][2 * ty + 1][lane] = xp_b;
    __syncthreads();
    xm_a = xc_a; xc_a = xp_a; bk0_a = bk1_a;
    xm_b = xc_b; xc_b = xp_b; bk0_b = bk1_b;
  }
  if (kNorm) {                                                 // a maximum is exact under any order (misc.c:307-317)
    for (int off = 32; off > 0; off >>= 1) { const double o2 = __shfl_down(lane_max, off, 64); lane_max = (o2 > lane_max) ? o2 : lane_max; }
    if (lane == 0) wg_max[ty] = lane_max;
    __syncthreads();
    if (lane == 0 && ty == 0) { double m = wg_max[0]; for (int q = 1; q < WJ; q++) m = (wg_max[q] > m) ? wg_max[q] : m; F.partials[logical] = m; }
  }
}


// ---------------------------------------------------------------------------------------------
// The cells a deferred launch (StencilArgs.defer) skipped: one lane per cell of every box face whose neighbour
// box lives on another rank, run after that rank's ghost data has been unpacked.  A cell on an edge or corner
// shared by several such faces is computed by the lowest-numbered face only (in-place GSRB must not update twice).
template <int V, int MODE>
__global__ __launch_bounds__(256) void stencil7_shell_kernel(const hpgmg_hip_level L, const StencilArgs P) {
  const int box = blockIdx.y / 6, dir = blockIdx.y % 6;
  const int *nb = L.box_nbr + 6 * box;
  if (nb[dir] != -2) return;
  const int dim = L.dim, last = dim - 1, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= dim * dim) return;
  const int u = t % dim, v = t / dim, side = (dir & 1) ? last : 0;
  int i, j, k;
  if (dir < 2) { i = side; j = u; k = v; } else if (dir < 4) { j = side; i = u; k = v; } else { k = side; i = u; j = v; }
  const int on_face[6] = { i == 0, i == last, j == 0, j == last, k == 0, k == last };
  for (int d = 0; d < dir; d++) if (on_face[d] && nb[d] == -2) return;
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  const int jS = L.jStride, kS = L.kStride, ijk = i + j * jS + k * kS;
  if (MODE == MODE_GSRB) {
    const int colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;
    if (((i ^ j ^ k ^ colour000) & 1) != 0) return;
  }
  const double *x = vec_origin(L, box, P.xn_id);
  double *out = vec_origin(L, box, P.xout_id);
  const double xc = x[ijk];
  auto outside = [&](int d, int idx_in_neighbour, int idx_ghost) -> double {
    const int n = nb[d];
    if (n >= 0) return vec_origin(L, n, P.xn_id)[idx_in_neighbour];
    if (n == -1) return -xc;
    return x[idx_ghost];
  };
  const bool gf = P.ghost_free != 0;
  const double xim = (gf && i == 0)    ? outside(0, last + j * jS + k * kS, ijk - 1)  : x[ijk - 1];
  const double xip = (gf && i == last) ? outside(1, j * jS + k * kS, ijk + 1)          : x[ijk + 1];
  const double xjm = (gf && j == 0)    ? outside(2, i + last * jS + k * kS, ijk - jS) : x[ijk - jS];
  const double xjp = (gf && j == last) ? outside(3, i + k * kS, ijk + jS)              : x[ijk + jS];
  const double xkm = (gf && k == 0)    ? outside(4, i + j * jS + last * kS, ijk - kS) : x[ijk - kS];
  const double xkp = (gf && k == last) ? outside(5, i + j * jS, ijk + kS)              : x[ijk + kS];
  double bi0 = 0, bi1 = 0, bj0 = 0, bj1 = 0, bk0 = 0, bk1 = 0, al = 0;
  if (kVC) {
    const double *bi = vec_origin(L, box, VECTOR_BETA_I), *bj = vec_origin(L, box, VECTOR_BETA_J), *bk = vec_origin(L, box, VECTOR_BETA_K);
    bi0 = bi[ijk]; bi1 = bi[ijk + 1]; bj0 = bj[ijk]; bj1 = bj[ijk + jS]; bk0 = bk[ijk]; bk1 = bk[ijk + kS];
  }
  if (kHelm) al = vec_origin(L, box, VECTOR_ALPHA)[ijk];
  const double Ax = apply_op_7pt<V>(xc, xim, xip, xjm, xjp, xkm, xkp, bi0, bi1, bj0, bj1, bk0, bk1, al, P.a, P.b, P.h2inv);
  if (MODE == MODE_APPLY) { out[ijk] = Ax; return; }
  const double rhs = vec_origin(L, box, P.rhs_id)[ijk];
  if (MODE == MODE_RESIDUAL) { out[ijk] = rhs - Ax; return; }
  const double dinv = vec_origin(L, box, VECTOR_DINV)[ijk];
  if (MODE == MODE_CHEBY)      { const double xnm1 = out[ijk]; out[ijk] = xc + P.c1 * (xc - xnm1) + P.c2 * dinv * (rhs - Ax); }
  else if (MODE == MODE_GSRB)  { out[ijk] = xc + dinv * (rhs - Ax); }
  else                         { out[ijk] = xc + P.c2 * dinv * (rhs - Ax); }
}

// ---------------------------------------------------------------------------------------------
// 27-point constant-coefficient operator (reference operators.27pt.c:48-51,60-91).
// One lane per (i,j) column marching in +k with the three 3x3 planes around the cell held in
// registers (27 values, 9 new loads per step instead of 27); streams: x, x_nm1, rhs, Dinv =
// 40 B per cell for Chebyshev.  The weighted partial sums are formed in the reference's order:
// ((C3*corners + C2*edges) + C1*faces) + C0*centre, each group summed left to right as listed.
// MODE_BLACKBOX is the probe of operators/rebuild.c:126-132 (out = Aii, rhs slot = sum|Aij|).
template <int MODE>
__global__ __launch_bounds__(256) void stencil27_kernel(const hpgmg_hip_level L, const StencilArgs P) {
  const int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;
  const int i = ti * (int)blockDim.x + (int)threadIdx.x;
  const int j = tj * (int)blockDim.y + (int)threadIdx.y;
  if (i >= L.dim || j >= L.dim) return;
  const int k0 = ck * P.kchunk, k1 = (k0 + P.kchunk < L.dim) ? k0 + P.kchunk : L.dim;
  const int jS = L.jStride, kS = L.kStride;
  constexpr bool kSmooth = (MODE == MODE_CHEBY || MODE == MODE_GSRB || MODE == MODE_JACOBI);

  const double *__restrict__ x = vec_origin(L, box, P.xn_id);      // 27-pt GSRB is always out of place
  double *__restrict__ out = vec_origin(L, box, P.xout_id);
  double *rhs = (MODE == MODE_APPLY) ? nullptr : vec_origin(L, box, P.rhs_id);   // BLACKBOX: the sum|Aij| accumulator
  const double *__restrict__ dinv = kSmooth ? vec_origin(L, box, VECTOR_DINV) : nullptr;
  int colour000 = 0;
  if (MODE == MODE_GSRB) colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;

  int ijk = i + j * jS + k0 * kS;
  plane9 m = load_plane(x + ijk - kS, jS), c = load_plane(x + ijk, jS);
  for (int k = k0; k < k1; k++, ijk += kS) {
    const plane9 p = load_plane(x + ijk + kS, jS);
    const double xc = c.v[1][1];
    bool update = true;
    if (MODE == MODE_GSRB) update = (((i ^ j ^ k ^ colour000) & 1) == 0);
    if (update) {
      const double Ax = apply_op_27pt(m, c, p, P.a, P.b, P.h2inv);
      if (MODE == MODE_CHEBY)         { const double xnm1 = out[ijk]; out[ijk] = xc + P.c1 * (xc - xnm1) + P.c2 * dinv[ijk] * (rhs[ijk] - Ax); }
      else if (MODE == MODE_GSRB)     { out[ijk] = xc + dinv[ijk] * (rhs[ijk] - Ax); }
      else if (MODE == MODE_JACOBI)   { out[ijk] = xc + P.c2 * dinv[ijk] * (rhs[ijk] - Ax); }
      else if (MODE == MODE_RESIDUAL) { out[ijk] = rhs[ijk] - Ax; }
      else if (MODE == MODE_APPLY)    { out[ijk] = Ax; }
      else { out[ijk] += (xc) * Ax; rhs[ijk] += fabs((1.0 - xc) * Ax); }
    } else {
      out[ijk] = xc;   // out-of-place GSRB copies the other colour (gsrb.c:94-98)
    }
    m = c; c = p;
  }
}


template <int V, int MODE>
__global__ __launch_bounds__(256) void stencil_direct_kernel(const hpgmg_hip_level L, const StencilArgs P) {
  const int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;
  // GSRB: a lane owns the PAIR of cells (2 ip, 2 ip + 1) and sweeps whichever of the two has this half sweep's colour, so no
  // lane idles on the other colour (the launcher halves the tiles along i); every other mode: one cell per lane
  const int ip = ti * (int)blockDim.x + (int)threadIdx.x;
  const int i = (MODE == MODE_GSRB) ? 2 * ip : ip;
  const int j = tj * (int)blockDim.y + (int)threadIdx.y;
  if (i >= L.dim || j >= L.dim) return;
  const int k0 = ck * P.kchunk, k1 = (k0 + P.kchunk < L.dim) ? k0 + P.kchunk : L.dim;
  const int jS = L.jStride, kS = L.kStride;
  constexpr bool kSmooth = (MODE == MODE_CHEBY || MODE == MODE_GSRB || MODE == MODE_JACOBI);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);

  const double *x = vec_origin(L, box, P.xn_id);
  double *out = vec_origin(L, box, P.xout_id);           // in-place GSRB (fv2) aliases x
  double *rhs = (MODE == MODE_APPLY) ? nullptr : vec_origin(L, box, P.rhs_id);   // BLACKBOX: the sum|Aij| accumulator
  const double *dinv = kSmooth ? vec_origin(L, box, VECTOR_DINV) : nullptr;
  const double *alpha = kHelm ? vec_origin(L, box, VECTOR_ALPHA) : nullptr;
  const double *bi = kVC ? vec_origin(L, box, VECTOR_BETA_I) : nullptr;
  const double *bj = kVC ? vec_origin(L, box, VECTOR_BETA_J) : nullptr;
  const double *bk = kVC ? vec_origin(L, box, VECTOR_BETA_K) : nullptr;
  int colour000 = 0;
  if (MODE == MODE_GSRB) colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;

  int row = j * jS + k0 * kS;                            // (0, j, k)
  for (int k = k0; k < k1; k++, row += kS) {
    if (MODE == MODE_GSRB) {
      const int sel = (j ^ k ^ colour000) & 1;             // cell i is swept when (i ^ j ^ k ^ colour000) is even: of the pair, the one with i & 1 == sel
      const int iu = i + sel, io = i + 1 - sel;
      if (iu < L.dim) {
        const int ijk = iu + row;
        const double xc = x[ijk];
        const double Ax = apply_op_direct<V>(x, alpha, bi, bj, bk, ijk, jS, kS, P.a, P.b, P.h2inv);
        out[ijk] = xc + dinv[ijk] * (rhs[ijk] - Ax);
      }
      if (P.copy_other_colour && io < L.dim) out[io + row] = x[io + row];
    } else {
      const int ijk = i + row;
      const double xc = x[ijk];
      const double Ax = apply_op_direct<V>(x, alpha, bi, bj, bk, ijk, jS, kS, P.a, P.b, P.h2inv);
      if (MODE == MODE_CHEBY)         { const double xnm1 = out[ijk]; out[ijk] = xc + P.c1 * (xc - xnm1) + P.c2 * dinv[ijk] * (rhs[ijk] - Ax); }
      else if (MODE == MODE_JACOBI)   { out[ijk] = xc + P.c2 * dinv[ijk] * (rhs[ijk] - Ax); }
      else if (MODE == MODE_RESIDUAL) { out[ijk] = rhs[ijk] - Ax; }
      else if (MODE == MODE_APPLY)    { out[ijk] = Ax; }
      else { out[ijk] += (xc) * Ax; rhs[ijk] += fabs((1.0 - xc) * Ax); }
    }
  }
}

// ---- smoother-kernel profiling: hipEvent pair around every smoother launch ----
static bool g_profile = false;
static long long g_profile_min_cells = 0;   // only launches covering at least this many cells are timed
static const int kMaxPairs = 8192;
static hipEvent_t g_ev[2 * kMaxPairs];
static int g_pairs_alloc = 0, g_pairs_used = 0;
static long long g_prof_cells = 0, g_prof_launches = 0;
static double g_prof_ms_flushed = 0.0;
static bool g_profile_skipped_part1 = false;

static void profile_flush() {
  for (int p = 0; p < g_pairs_used; p++) {
    float ms = 0.f;
    hipEventSynchronize(g_ev[2 * p + 1]);
    if (hipEventElapsedTime(&ms, g_ev[2 * p], g_ev[2 * p + 1]) == hipSuccess) g_prof_ms_flushed += ms;
  }
  g_pairs_used = 0;
}
extern "C" int hpgmg_hip_graph_is_open(void);
extern "C" int hpgmg_hip_graph_flush(void);
// An event is a marker packet, and the GPU idles ~5 us around each (measured: 8 per config-2 solve = 1.2 % of it; events attached to the dispatches
// themselves, hipExtLaunchKernelGGL, idle it longer).  So only every g_profile_stride-th eligible launch is timed: a stride coprime with the launches
// per solve (4 sweep pairs, 6 red + black passes, 8 sweeps) visits every position of the cycle in turn.
static int g_profile_stride = 1, g_profile_seen = 0;
int profile_begin(long long cells) {
  if (!g_profile || cells < g_profile_min_cells || hpgmg_hip_graph_is_open()) return -1;
  if (g_tile_part != 2 && (g_profile_seen++ % g_profile_stride) != 0) { g_profile_skipped_part1 = (g_tile_part == 1); return -1; }
  if (g_tile_part == 2 && g_profile_skipped_part1) return -1;          // the second part of a launch whose first part was not timed
  if (g_pairs_used == kMaxPairs) profile_flush();
  if (g_pairs_used == g_pairs_alloc) { hipEventCreate(&g_ev[2 * g_pairs_alloc]); hipEventCreate(&g_ev[2 * g_pairs_alloc + 1]); g_pairs_alloc++; }
  int p = g_pairs_used++;
  hipEventRecord(g_ev[2 * p], g_stream);
  return p;
}
// first_part: part 1 of a two-part launch (hpgmg_hip_set_tile_part): its time and cells are added to the launch that part 2 completes
void profile_end(int p, long long cells, bool first_part) {
  if (p < 0) return;
  hipEventRecord(g_ev[2 * p + 1], g_stream);
  g_prof_cells += cells;
  if (!first_part) g_prof_launches++;
}

static InterpFold g_fold = {};            // set by hpgmg_hip_stencil_fold_interpolation for the NEXT hpgmg_hip_smooth_cheby call, which moves it to g_fold_now whatever path it takes
static InterpFold g_fold_now = {};        // the fold of the hpgmg_hip_smooth_cheby call in progress
static int g_ghost_free = 0;
static int g_tile_ghost_free = 0;
static TileFused g_tile_fused = {};      // consumed by the next tiled residual launch (27-point / fv4): what becomes of the residual
static int g_tile_last_blocks = 0;       // workgroups of that launch (= partial maxima written)   // the LDS-tiled fv4 / 27-point kernels read x outside a box from the neighbouring box (all boxes local)
static int g_defer_mode = 0;   // 0: whole boxes; 1: skip the cells next to remote faces; 2: only those cells (stencil7_shell_kernel)

// A fused residual form was requested (g_tile_fused) but the launch is about to take a kernel that cannot honour it -- it would STORE the
// residual over the vector the caller passed as a dummy output.  Refuse loudly; never leave the request pending for a later launch.
static int refuse_unfused(const char *kernel) {
  if (!g_tile_fused.kind) return 0;
  g_tile_fused = TileFused{};
  return record_error(hipErrorInvalidValue, kernel);
}

static void plan(const hpgmg_hip_level *L, StencilArgs &P, dim3 &block, int &grid) {
  static const int tune_ty = env_int("HPGMG_TUNE_TY", 0), tune_kchunk = env_int("HPGMG_TUNE_KCHUNK", 0);   // experiments only
  P.ghost_free = (g_ghost_free && L->box_nbr) ? 1 : 0;
  int tx = 64;
  while (tx > 1 && tx / 2 >= L->dim) tx /= 2;           // smallest power of two >= dim, capped at one wave
  int ty = 256 / tx;
  if (tune_ty > 0 && tx == 64) ty = tune_ty;
  while (ty > 1 && ty / 2 >= L->dim) ty /= 2;
  block = dim3(tx, ty, 1);
  P.tiles_i = (L->dim + tx - 1) / tx;
  P.tiles_j = (L->dim + ty - 1) / ty;
  // k chunk: as long as possible (plane reuse in registers) while still exposing ~16 workgroups per CU; on the
  // mid-size levels (<= 128^3) parallelism wins over reuse -- their planes are L2 resident anyway (measured:
  // 7.07 -> 5.70 ms per 256^3 F-cycle going from a 16-plane minimum to this rule)
  int kchunk = L->dim;
  static const int min_kchunk = env_int("HPGMG_TUNE_MIN_KCHUNK", 1), want_blocks = env_int("HPGMG_TUNE_WANT_BLOCKS", 4096);
  while (kchunk > min_kchunk && (long long)L->num_boxes * P.tiles_i * P.tiles_j * ((L->dim + kchunk - 1) / kchunk) < want_blocks) kchunk /= 2;
  if (tune_kchunk > 0 && L->dim >= tune_kchunk) kchunk = tune_kchunk;
  P.kchunk = kchunk;
  P.chunks_k = (L->dim + kchunk - 1) / kchunk;
  P.total_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i;
  grid = grid_for(P.total_blocks, &P.per_xcd);
}

// smallest box side the tiled 27-point kernel takes: 64.  The 32 x 16 tiles for boxes of 32^3 work (HPGMG_TUNE_27PT_TILE32=1, covered by the
// tests) but measured 0.35 ms per `7 64` F-cycle SLOWER than the register kernel on that level (the 128^3 level is cache resident)
static int g_s27_tile32 = -1;
static int s27_tile_granule() { if (g_s27_tile32 < 0) g_s27_tile32 = env_int("HPGMG_TUNE_27PT_TILE32", 0) ? 1 : 0; return g_s27_tile32 ? 32 : 64; }
template <int MODE>
static int launch27(const hpgmg_hip_level *L, StencilArgs P, bool is_smoother) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  static const int no_tile27 = env_int("HPGMG_TUNE_27PT_DIRECT", 0);
  if (MODE != MODE_BLACKBOX && !no_tile27 && L->dim % s27_tile_granule() == 0 && P.xn_id != P.xout_id) {       // LDS-staged kernel (stencil27_tile.hpp)
    constexpr int TM = (MODE == MODE_BLACKBOX) ? MODE_APPLY : MODE;
    const bool narrow = (L->dim % 64 != 0);               // boxes of 32^3: 32 x 16 tiles
    const int TI = narrow ? 32 : 64, TJ = narrow ? 16 : 8;
    S27TileArgs A = {};
    A.xn_id = P.xn_id; A.xout_id = P.xout_id; A.rhs_id = P.rhs_id; A.mode = TM; A.a = P.a; A.b = P.b; A.h2inv = P.h2inv; A.c1 = P.c1; A.c2 = P.c2; A.sweep = P.sweep;
    A.ghost_free = (g_tile_ghost_free && L->box_nbr) ? 1 : 0;
    A.fused = g_tile_fused; g_tile_fused = TileFused{};
    A.tiles_i = L->dim / TI; A.tiles_j = L->dim / TJ;
    int kchunk = L->dim;
    while (kchunk > 32 && (long long)L->num_boxes * A.tiles_i * A.tiles_j * (L->dim / kchunk) < 8192) kchunk /= 2;   // measured at 512^3: 32-plane chunks 929 us, whole boxes 952
    static const int tune_kc = env_int("HPGMG_TUNE_27PT_KCHUNK", 0);
    if (tune_kc > 0 && L->dim % tune_kc == 0) kchunk = tune_kc;
    A.kchunk = kchunk; A.chunks_k = (L->dim + kchunk - 1) / kchunk;
    A.total_blocks = L->num_boxes * A.chunks_k * A.tiles_j * A.tiles_i;
    g_tile_last_blocks = A.total_blocks;
    int tgrid = grid_for(A.total_blocks, &A.per_xcd);
    long long tcells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
    const long long whole_cells = tcells;
    const bool first_part = g_tile_part == 1 && A.fused.kind == 0;
    if (g_tile_part && A.fused.kind == 0) {      // one part of the launch (hpgmg_hip_set_tile_part): part 1 = the tiles that read nothing of an image of another rank's box
      int count = 0;
      A.order = tile_part_order(L, A.tiles_i, A.tiles_j, A.chunks_k, g_tile_part, false, &tgrid, &A.per_xcd, &count);
      if (tgrid == 0) return 0;
      if (!A.order) return record_error(hipErrorOutOfMemory, "stencil27_tile: dispatch list of a partial launch");
      tcells = tcells * count / A.total_blocks;
    }
    const int tprof = is_smoother ? profile_begin(whole_cells) : -1;
    if (narrow) hipLaunchKernelGGL((stencil27_tile_kernel<TM, 16, 32>), dim3(tgrid), dim3(32, 16), 0, g_stream, *L, A);
    else        hipLaunchKernelGGL((stencil27_tile_kernel<TM, 8, 64>), dim3(tgrid), dim3(64, 8), 0, g_stream, *L, A);
    profile_end(tprof, tcells, first_part);
    HPGMG_LAUNCH_CHECK("stencil27_tile_kernel");
    return 0;
  }
  if (int e = refuse_unfused("fused residual form requested, but this level runs stencil27_kernel")) return e;
  dim3 block; int grid;
  plan(L, P, block, grid);
  const long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  int prof = is_smoother ? profile_begin(cells) : -1;
  hipLaunchKernelGGL((stencil27_kernel<MODE>), dim3(grid), block, 0, g_stream, *L, P);
  profile_end(prof, cells);
  HPGMG_LAUNCH_CHECK("stencil27_kernel");
  return 0;
}

static int fv4_tile_granule() { static const int g = env_int("HPGMG_TUNE_FV4_TILE32", 1) ? 32 : 64; return g; }   // smallest box side the tiled fv4 kernel takes
// the LDS-tiled 4th-order kernel (fv4_tile.hpp): boxes whose side is a multiple of 32, out of place
template <int MODE, int TJ, int TI>
static int launch_fv4_tile_tj(const hpgmg_hip_level *L, int variant, const StencilArgs &S, bool is_smoother) {
  Fv4TileArgs P = {};
  P.xn_id = S.xn_id; P.xout_id = S.xout_id; P.rhs_id = S.rhs_id; P.a = S.a; P.b = S.b; P.h2inv = S.h2inv; P.c1 = S.c1; P.c2 = S.c2;
  P.sweep = S.sweep; P.copy_other_colour = S.copy_other_colour; P.ghost_free = (g_tile_ghost_free && L->box_nbr) ? 1 : 0;
  P.fused = g_tile_fused; g_tile_fused = TileFused{};
  P.tiles_i = L->dim / TI; P.tiles_j = L->dim / TJ;
  int kchunk = L->dim;                                   // enough workgroups to fill the chip, as few chunk prologues as possible
  const int want = (TJ * TI >= 1024) ? 512 : 1024;
  static const int kc_min = env_int("HPGMG_TUNE_FV4_KCHUNK_MIN", 8);      // (a chunk costs four extra planes of loads: chunks of 4 planes -- what filling 1024 slots asked for on the 128^3 level of 7 64 -- fetch every plane twice; 8: 29 vs 34 us per half sweep there, tools/ab_fv4_kcmin.sh)
  while (kchunk > kc_min && (long long)L->num_boxes * P.tiles_i * P.tiles_j * (L->dim / kchunk) < want) kchunk /= 2;
  static const int tune_kc = env_int("HPGMG_TUNE_FV4_KCHUNK", 0);
  if (tune_kc > 0 && L->dim % tune_kc == 0) kchunk = tune_kc;
  P.kchunk = kchunk; P.chunks_k = (L->dim + kchunk - 1) / kchunk;
  P.total_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i;
  g_tile_last_blocks = P.total_blocks;
  int grid = grid_for(P.total_blocks, &P.per_xcd);
  const size_t lds = (size_t)11 * (TI + 4) * (TJ + 4) * sizeof(double);
  long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  const long long whole_cells = cells;
  const bool first_part = g_tile_part == 1 && P.fused.kind == 0;
  if (g_tile_part && P.fused.kind == 0) {        // one part of the launch (hpgmg_hip_set_tile_part): part 1 = the tiles that read nothing of an image of another rank's box
    int count = 0;
    P.order = tile_part_order(L, P.tiles_i, P.tiles_j, P.chunks_k, g_tile_part, false, &grid, &P.per_xcd, &count);
    if (grid == 0) return 0;
    if (!P.order) return record_error(hipErrorOutOfMemory, "fv4_tile: dispatch list of a partial launch");
    cells = cells * count / P.total_blocks;
  }
  const int prof = is_smoother ? profile_begin(whole_cells) : -1;
#define FV4_TILE_CASE(VAR) { \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)fv4_tile_kernel<VAR, MODE, TJ, TI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; } \
    hipLaunchKernelGGL((fv4_tile_kernel<VAR, MODE, TJ, TI>), dim3(grid), dim3(TI, TJ), lds, g_stream, *L, P); }
  if (variant == HPGMG_HIP_FV4_VC_HELMHOLTZ) FV4_TILE_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ)
  else FV4_TILE_CASE(HPGMG_HIP_FV4_VC_POISSON)
#undef FV4_TILE_CASE
  profile_end(prof, cells, first_part);
  HPGMG_LAUNCH_CHECK("fv4_tile_kernel");
  return 0;
}
// Tile height 8 (two workgroups per CU; 1.59 values loaded per cell and array) measured 1.85 ms per half sweep at 512^3, height 16 (one
// workgroup of 16 waves and 120 KB of LDS per CU; 1.33 values) 1.92 ms: the second resident workgroup hides more latency than the smaller
// halo saves.  Boxes of 32^3 (the 128^3 level of `7 64`): 32 x 16 tiles (a wave = two rows of 32 cells, still conflict free in LDS).
template <int MODE>
static int launch_fv4_tile(const hpgmg_hip_level *L, int variant, const StencilArgs &S, bool is_smoother) {
  static const int tj = env_int("HPGMG_TUNE_FV4_TJ", 8);
  if (L->dim % 64 != 0) return launch_fv4_tile_tj<MODE, 16, 32>(L, variant, S, is_smoother);
  if (tj == 8) return launch_fv4_tile_tj<MODE, 8, 64>(L, variant, S, is_smoother);
  return launch_fv4_tile_tj<MODE, 16, 64>(L, variant, S, is_smoother);
}

template <int MODE>
static int launch_direct(const hpgmg_hip_level *L, int variant, StencilArgs P, bool is_smoother) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  if (MODE != MODE_BLACKBOX && (variant == HPGMG_HIP_FV4_VC_HELMHOLTZ || variant == HPGMG_HIP_FV4_VC_POISSON)) {
    static const int no_tile = env_int("HPGMG_TUNE_FV4_DIRECT", 0);
    if (!no_tile && L->dim % fv4_tile_granule() == 0 && L->ghosts >= 2 && P.xn_id != P.xout_id) return launch_fv4_tile<(MODE == MODE_BLACKBOX) ? MODE_APPLY : MODE>(L, variant, P, is_smoother);
  }
  if (int e = refuse_unfused("fused residual form requested, but this level runs stencil_direct_kernel")) return e;
  dim3 block; int grid;
  plan(L, P, block, grid);
  P.ghost_free = 0;
  if (MODE == MODE_GSRB) {                       // a lane owns two cells along i (see the kernel)
    P.tiles_i = ((L->dim + 1) / 2 + (int)block.x - 1) / (int)block.x;
    P.total_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i;
    grid = grid_for(P.total_blocks, &P.per_xcd);
  }
  const long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  int prof = is_smoother ? profile_begin(cells) : -1;
  switch (variant) {
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil_direct_kernel<HPGMG_HIP_FV4_VC_HELMHOLTZ, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    case HPGMG_HIP_FV4_VC_POISSON:   hipLaunchKernelGGL((stencil_direct_kernel<HPGMG_HIP_FV4_VC_POISSON, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil_direct_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil_direct_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    case HPGMG_HIP_7PT_CC:           hipLaunchKernelGGL((stencil_direct_kernel<HPGMG_HIP_7PT_CC, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    default: return record_error(hipErrorInvalidValue, "stencil variant not implemented");
  }
  profile_end(prof, cells);
  HPGMG_LAUNCH_CHECK("stencil_direct_kernel");
  return 0;
}

template <int MODE>
static int launch(const hpgmg_hip_level *L, int variant, StencilArgs P, bool is_smoother) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  if (variant == HPGMG_HIP_27PT_CC) return launch27<MODE>(L, P, is_smoother);
  if (variant == HPGMG_HIP_FV4_VC_HELMHOLTZ || variant == HPGMG_HIP_FV4_VC_POISSON) return launch_direct<MODE>(L, variant, P, is_smoother);
  if (int e = refuse_unfused("fused residual form requested for a variant without tiled kernels")) return e;
  dim3 block; int grid;
  plan(L, P, block, grid);
  if (g_defer_mode) {
    if (!P.ghost_free || P.copy_other_colour || MODE == MODE_BLACKBOX) return record_error(hipErrorInvalidValue, "deferred stencil launch needs the ghost-free 7-point path");
    if (g_defer_mode == 2) {
      if (L->dim < 2) return record_error(hipErrorInvalidValue, "deferred stencil launch needs boxes of at least 2^3");
      dim3 sgrid((L->dim * L->dim + 255) / 256, L->num_boxes * 6);
      switch (variant) {
        case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_shell_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE>), sgrid, dim3(256), 0, g_stream, *L, P); break;
        case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_shell_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE>), sgrid, dim3(256), 0, g_stream, *L, P); break;
        case HPGMG_HIP_7PT_CC:           hipLaunchKernelGGL((stencil7_shell_kernel<HPGMG_HIP_7PT_CC, MODE>), sgrid, dim3(256), 0, g_stream, *L, P); break;
        default: return record_error(hipErrorInvalidValue, "stencil variant not implemented");
      }
      HPGMG_LAUNCH_CHECK("stencil7_shell_kernel");
      return 0;
    }
    P.defer = 1;
  }
  const long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  int prof = is_smoother ? profile_begin(cells) : -1;
  static const int no_wide = env_int("HPGMG_TUNE_NO_WIDE", 0);
  // wide kernel: side a multiple of 128, 16-byte aligned interior rows (even strides; the box bases are checked by the host)
  if (g_fold_now.which && L->dim % 128 == 0) { return record_error(hipErrorInvalidValue, "folded interpolation: not in the wide kernel (hpgmg_hip_smooth_cheby_fold_supported)"); }
  if (!no_wide && L->dim % 128 == 0 && L->jStride % 2 == 0 && L->kStride % 2 == 0 && L->volume % 2 == 0 && (L->flags & 1)) {
    static const int wj = env_int("HPGMG_TUNE_WIDE_WJ", 8), tune_kc = env_int("HPGMG_TUNE_KCHUNK", 0);
    block = dim3(64, wj, 1);
    P.tiles_i = L->dim / 128; P.tiles_j = L->dim / (2 * wj);
    const int kchunk = tune_kc > 0 ? tune_kc : 16;
    P.kchunk = kchunk; P.chunks_k = (L->dim + kchunk - 1) / kchunk;
    P.total_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i;
    grid = grid_for(P.total_blocks, &P.per_xcd);
#define WIDE_CASE(VAR) case VAR: \
      if (wj == 8) hipLaunchKernelGGL((stencil7_wide_kernel<VAR, MODE, 8>), dim3(grid), block, 0, g_stream, *L, P, FusedArgs{}); \
      else         hipLaunchKernelGGL((stencil7_wide_kernel<VAR, MODE, 4>), dim3(grid), block, 0, g_stream, *L, P, FusedArgs{}); \
      break;
    switch (variant) {
      WIDE_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ)
      WIDE_CASE(HPGMG_HIP_7PT_VC_POISSON)
      WIDE_CASE(HPGMG_HIP_7PT_CC)
      default: return record_error(hipErrorInvalidValue, "stencil variant not implemented");
    }
#undef WIDE_CASE
    profile_end(prof, cells);
    HPGMG_LAUNCH_CHECK("stencil7_wide_kernel");
    return 0;
  }
  // boxes of side 64 m (the 128^3 level of config 2): LDS-staged tile kernel (stencil7_tile.hpp)
  static const int no_tile7 = env_int("HPGMG_TUNE_7PT_NO_TILE", 0), tile7_kc = env_int("HPGMG_TUNE_7PT_TILE_KCHUNK", 8);
  if (!no_tile7 && !g_defer_mode && MODE != MODE_BLACKBOX && L->dim % 64 == 0) {
    constexpr int TJ = 8, TM = (MODE == MODE_BLACKBOX) ? MODE_APPLY : MODE;
    S7TileArgs A = {};
    A.xn_id = P.xn_id; A.xout_id = P.xout_id; A.rhs_id = P.rhs_id; A.a = P.a; A.b = P.b; A.h2inv = P.h2inv; A.c1 = P.c1; A.c2 = P.c2; A.sweep = P.sweep;
    A.ghost_free = P.ghost_free;
    A.tiles_i = L->dim / 64; A.tiles_j = L->dim / TJ;
    A.kchunk = (tile7_kc > 0 && L->dim % tile7_kc == 0) ? tile7_kc : 8; A.chunks_k = L->dim / A.kchunk;
    A.total_blocks = L->num_boxes * A.chunks_k * A.tiles_j * A.tiles_i;
    const int tgrid = grid_for(A.total_blocks, &A.per_xcd);
    if (MODE == MODE_CHEBY && g_fold_now.which) {
      const InterpFold F = g_fold_now;
      if (!P.ghost_free) return record_error(hipErrorInvalidValue, "folded interpolation needs the ghost-free path");
      switch (variant) {
        case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE_CHEBY, TJ, true>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A, F); break;
        case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE_CHEBY, TJ, true>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A, F); break;
        case HPGMG_HIP_7PT_CC:           hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_CC, MODE_CHEBY, TJ, true>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A, F); break;
        default: return record_error(hipErrorInvalidValue, "stencil variant not implemented");
      }
      profile_end(prof, cells);
      HPGMG_LAUNCH_CHECK("stencil7_tile_kernel (interpolation folded in)");
      return 0;
    }
    switch (variant) {
      case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, TM, TJ>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A); break;
      case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_VC_POISSON, TM, TJ>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A); break;
      case HPGMG_HIP_7PT_CC:           hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_CC, TM, TJ>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A); break;
      default: return record_error(hipErrorInvalidValue, "stencil variant not implemented");
    }
    profile_end(prof, cells);
    HPGMG_LAUNCH_CHECK("stencil7_tile_kernel");
    return 0;
  }
  if (MODE == MODE_CHEBY && g_fold_now.which) {
    const InterpFold F = g_fold_now;
    if (!P.ghost_free || P.defer) return record_error(hipErrorInvalidValue, "folded interpolation needs the ghost-free path with every box local");
    switch (variant) {
      case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE_CHEBY, true>), dim3(grid), block, 0, g_stream, *L, P, F); break;
      case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE_CHEBY, true>), dim3(grid), block, 0, g_stream, *L, P, F); break;
      case HPGMG_HIP_7PT_CC:           hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_CC, MODE_CHEBY, true>), dim3(grid), block, 0, g_stream, *L, P, F); break;
      default: return record_error(hipErrorInvalidValue, "stencil variant not implemented");
    }
    profile_end(prof, cells);
    HPGMG_LAUNCH_CHECK("stencil7_kernel (interpolation folded in)");
    return 0;
  }
  switch (variant) {
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    case HPGMG_HIP_7PT_CC:           hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_CC, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    default: return record_error(hipErrorInvalidValue, "stencil variant not implemented");
  }
  profile_end(prof, cells);
  HPGMG_LAUNCH_CHECK("stencil7_kernel");
  return 0;
}

// ---- fused forms of residual() on the bandwidth-bound fine level (stencil7_wide_kernel, ghost-free, every face local) ----
// residual + restriction + zero_vector on the launch-bound levels (boxes of an even side <= 32, every box local): stencil7_kernel<.., RR>
static int small_fused_ok(const hpgmg_hip_level *L, int variant) {
  if (variant != HPGMG_HIP_7PT_VC_HELMHOLTZ && variant != HPGMG_HIP_7PT_VC_POISSON && variant != HPGMG_HIP_7PT_CC) return 0;
  static const int on = env_int("HPGMG_TUNE_SMALL_RR", 1);      // 0: the two launches (experiments)
  return on && L->num_boxes > 0 && g_ghost_free && L->box_nbr && !g_defer_mode && !g_tile_part && L->dim % 2 == 0 && (L->dim <= 32 || (L->dim % 64 == 0 && L->dim % 128 != 0));
}
static int launch_small_fused(const hpgmg_hip_level *L, int variant, StencilArgs P, FusedArgs FA, int extra_blocks) {
  dim3 block; int grid;
  plan(L, P, block, grid);
  if (L->dim % 64 == 0) {      // boxes of 64^3 (config 2's 128^3 level): the LDS-staged tile kernel carries the form
    constexpr int TJ = 8;
    S7TileArgs A = {};
    A.xn_id = P.xn_id; A.xout_id = P.xout_id; A.rhs_id = P.rhs_id; A.a = P.a; A.b = P.b; A.h2inv = P.h2inv; A.ghost_free = 1;
    A.tiles_i = L->dim / 64; A.tiles_j = L->dim / TJ; A.kchunk = 8; A.chunks_k = L->dim / A.kchunk;
    A.total_blocks = L->num_boxes * A.chunks_k * A.tiles_j * A.tiles_i;
    const int tgrid = grid_for(A.total_blocks, &A.per_xcd) + extra_blocks;
    switch (variant) {
      case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE_RESIDUAL, TJ, false, true>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A, InterpFold{}, FA); break;
      case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE_RESIDUAL, TJ, false, true>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A, InterpFold{}, FA); break;
      default:                         hipLaunchKernelGGL((stencil7_tile_kernel<HPGMG_HIP_7PT_CC, MODE_RESIDUAL, TJ, false, true>), dim3(tgrid), dim3(64, TJ), 0, g_stream, *L, A, InterpFold{}, FA); break;
    }
    HPGMG_LAUNCH_CHECK("stencil7_tile_kernel (residual + restriction + zero_vector)");
    return 0;
  }
  if (!P.ghost_free || block.x > 32) return record_error(hipErrorInvalidValue, "fused residual on a small level: ghost-free path, boxes of side <= 32");
  if (P.kchunk < 2 || (P.kchunk & 1)) {          // plane PAIRS stay inside a chunk
    P.kchunk = (P.kchunk < 2) ? 2 : P.kchunk + 1;
    P.chunks_k = (L->dim + P.kchunk - 1) / P.kchunk;
    P.total_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i;
    grid = grid_for(P.total_blocks, &P.per_xcd);
  }
  grid += extra_blocks;
  switch (variant) {
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE_RESIDUAL, false, true>), dim3(grid), block, 0, g_stream, *L, P, InterpFold{}, FA); break;
    case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE_RESIDUAL, false, true>), dim3(grid), block, 0, g_stream, *L, P, InterpFold{}, FA); break;
    default:                         hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_CC, MODE_RESIDUAL, false, true>), dim3(grid), block, 0, g_stream, *L, P, InterpFold{}, FA); break;
  }
  HPGMG_LAUNCH_CHECK("stencil7_kernel (residual + restriction + zero_vector)");
  return 0;
}
static int wide_fused_ok(const hpgmg_hip_level *L, int variant) {
  if (variant != HPGMG_HIP_7PT_VC_HELMHOLTZ && variant != HPGMG_HIP_7PT_VC_POISSON && variant != HPGMG_HIP_7PT_CC) return 0;
  return L->num_boxes > 0 && g_ghost_free && L->box_nbr && !g_defer_mode && L->dim % 128 == 0 && L->jStride % 2 == 0 && L->kStride % 2 == 0 && L->volume % 2 == 0 && (L->flags & 1);
}
template <int MODE>
static int launch_wide_fused(const hpgmg_hip_level *L, int variant, StencilArgs P, FusedArgs F, int extra_blocks) {
  constexpr int wj = 8;
  P.ghost_free = 1;
  P.tiles_i = L->dim / 128; P.tiles_j = L->dim / (2 * wj); P.kchunk = 16; P.chunks_k = (L->dim + 15) / 16;
  F.compute_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i;
  P.total_blocks = F.compute_blocks;
  int grid = grid_for(P.total_blocks, &P.per_xcd);
  if (g_tile_part) {       // one part of the launch (faces on other ranks): part 1 = the tiles at no remote face, under the exchange; part 2 = the others, and the zeroing
    int count = 0;
    P.order = tile_part_order(L, P.tiles_i, P.tiles_j, P.chunks_k, g_tile_part, false, &grid, &P.per_xcd, &count);
    if (g_tile_part == 1) extra_blocks = 0;
    if (grid == 0 && extra_blocks == 0) return 0;
    if (grid > 0 && !P.order) return record_error(hipErrorOutOfMemory, "fused residual: dispatch list of a partial launch");
  }
  grid += extra_blocks;                                                      // the extra (zeroing) workgroups follow the stencil grid
  switch (variant) {
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_wide_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE, wj>), dim3(grid), dim3(64, wj), 0, g_stream, *L, P, F); break;
    case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_wide_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE, wj>), dim3(grid), dim3(64, wj), 0, g_stream, *L, P, F); break;
    default:                         hipLaunchKernelGGL((stencil7_wide_kernel<HPGMG_HIP_7PT_CC, MODE, wj>), dim3(grid), dim3(64, wj), 0, g_stream, *L, P, F); break;
  }
  HPGMG_LAUNCH_CHECK("stencil7_wide_kernel (fused residual)");
  return 0;
}
}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

void hpgmg_hip_set_ghost_free(int on) { g_ghost_free = on; }
void hpgmg_hip_set_defer_mode(int mode) { g_defer_mode = mode; }
void hpgmg_hip_set_tile_ghost_free(int on) { g_tile_ghost_free = on; }
void hpgmg_hip_set_27pt_tile32(int on) { g_s27_tile32 = on ? 1 : 0; }
// would smooth / residual / apply_op of this variant run the LDS-tiled kernel on this level (out of place)?
int hpgmg_hip_tile_kernel_applies(const hpgmg_hip_level *L, int variant, int out_of_place) {
  static const int no_fv4 = env_int("HPGMG_TUNE_FV4_DIRECT", 0), no_27 = env_int("HPGMG_TUNE_27PT_DIRECT", 0);
  if (L->num_boxes <= 0 || !out_of_place) return 0;
  if (variant == HPGMG_HIP_27PT_CC) return !no_27 && L->dim % s27_tile_granule() == 0;
  if (variant == HPGMG_HIP_FV4_VC_HELMHOLTZ || variant == HPGMG_HIP_FV4_VC_POISSON) return !no_fv4 && L->ghosts >= 2 && L->dim % fv4_tile_granule() == 0;
  return 0;
}
int hpgmg_hip_get_ghost_free(void) { return g_ghost_free; }

void hpgmg_hip_profile_smoother(int enable) {
  if (enable) { profile_flush(); g_prof_ms_flushed = 0.0; g_prof_cells = 0; g_prof_launches = 0; }
  g_profile = enable != 0;
}
void hpgmg_hip_profile_smoother_min_cells(long long min_cells) { g_profile_min_cells = min_cells; }
void hpgmg_hip_profile_smoother_stride(int stride) { g_profile_stride = stride > 0 ? stride : 1; g_profile_seen = 0; }
int hpgmg_hip_profile_smoother_read(double *total_ms, long long *launches, long long *cells) {
  profile_flush();
  if (total_ms) *total_ms = g_prof_ms_flushed;
  if (launches) *launches = g_prof_launches;
  if (cells) *cells = g_prof_cells;
  return 0;
}

// the NEXT hpgmg_hip_smooth_cheby launch reads its operand with the piecewise-constant interpolation from (Lc, coarse_id) folded in (InterpFold)
int hpgmg_hip_smooth_cheby_fold_supported(const hpgmg_hip_level *L, int variant) {
  return (variant == HPGMG_HIP_7PT_VC_HELMHOLTZ || variant == HPGMG_HIP_7PT_VC_POISSON || variant == HPGMG_HIP_7PT_CC) && L->box_nbr && L->dim % 128 != 0 && L->dim % 2 == 0 && !g_defer_mode;
}
void hpgmg_hip_stencil_fold_interpolation(const hpgmg_hip_level *Lc, int coarse_id, const int *map, int which) {
  g_fold = InterpFold{}; if (Lc && which) { g_fold.which = which; g_fold.Lc = *Lc; g_fold.coarse_id = coarse_id; g_fold.map = map; }
}
int hpgmg_hip_smooth_cheby(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                           double a, double b, double h2inv, double c1, double c2) {
  StencilArgs P = {}; P.xn_id = xn_id; P.xout_id = xnp1_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv; P.c1 = c1; P.c2 = c2;
  g_fold_now = g_fold; g_fold = InterpFold{};            // a requested fold belongs to THIS call, also when the launch is skipped (graph replay)
  if (g_fold_now.which && variant != HPGMG_HIP_7PT_VC_HELMHOLTZ && variant != HPGMG_HIP_7PT_VC_POISSON && variant != HPGMG_HIP_7PT_CC) { g_fold_now = InterpFold{}; return record_error(hipErrorInvalidValue, "folded interpolation: 7-point variants only"); }
  const int e = launch<MODE_CHEBY>(L, variant, P, true);
  g_fold_now = InterpFold{};
  return e;
}
// Two Chebyshev sweeps in one pass (cheby_pair.hpp).  Vector references are (scratch?, id) pairs: scratch ids 0/1
// address the two plugin-private vectors behind scr_base.  Returns hipErrorNotSupported-like status 1 (no launch,
// no error recorded) when the level does not fit the kernel's assumptions, so the caller can fall back.
int hpgmg_hip_smooth_gsrb(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                          double a, double b, double h2inv, int sweep) {
  StencilArgs P = {}; P.xn_id = xn_id; P.xout_id = xnp1_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv; P.sweep = sweep;
  P.copy_other_colour = (xn_id != xnp1_id);
  return launch<MODE_GSRB>(L, variant, P, true);
}
// Both coloured half sweeps (sweep, sweep + 1; sweep even) of an out-of-place 27-point GSRB sweep in one pass (stencil27_rb.hpp):
// x_id -> out_id, the intermediate vector never stored.  Needs boxes of side 64 m, all of them local, apply_BCs_p2 done on x_id.
static long long g_rb27_launches = 0;
long long hpgmg_hip_rb27_launch_count(void) { return g_rb27_launches; }   // launches of the one-pass red + black kernel so far (tests)
int hpgmg_hip_smooth_gsrb27_rb_supported(const hpgmg_hip_level *L) {
  static const int off = env_int("HPGMG_TUNE_27PT_NO_RB", 0);
  return !off && L->num_boxes > 0 && L->dim % 64 == 0 && L->box_nbr != nullptr && L->ghosts >= 1;
}
int hpgmg_hip_smooth_gsrb27_rb(const hpgmg_hip_level *L, int x_id, int out_id, int rhs_id, double a, double b, double h2inv, int sweep) {
  HPGMG_SKIP_IF_REPLAY();
  if (!hpgmg_hip_smooth_gsrb27_rb_supported(L) || x_id == out_id || (sweep & 1)) return record_error(hipErrorInvalidValue, "smooth_gsrb27_rb: level / arguments not supported");
  constexpr int TJ = 16;                                // rows of a tile; a lane owns two of them
  S27RbArgs A = {};
  A.xn_id = x_id; A.xout_id = out_id; A.rhs_id = rhs_id; A.a = a; A.b = b; A.h2inv = h2inv; A.sweep = sweep;
  A.tiles_i = L->dim / 64; A.tiles_j = L->dim / TJ;
  int kchunk = L->dim;
  // two workgroups fit a CU: 512 workgroups fill the chip, and every k chunk costs three extra planes of loads and a red stage more
  // (measured at 512^3: whole boxes 1.09 ms, chunks of 64 planes 1.19, of 32 planes 1.27)
  while (kchunk > 16 && (long long)L->num_boxes * A.tiles_i * A.tiles_j * (L->dim / kchunk) < 512) kchunk /= 2;
  static const int tune_kc = env_int("HPGMG_TUNE_27PT_RB_KCHUNK", 0);
  if (tune_kc > 0 && L->dim % tune_kc == 0) kchunk = tune_kc;
  A.kchunk = kchunk; A.chunks_k = (L->dim + kchunk - 1) / kchunk;
  A.total_blocks = L->num_boxes * A.chunks_k * A.tiles_j * A.tiles_i;
  int grid = grid_for(A.total_blocks, &A.per_xcd);
  long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  const long long whole_cells = cells;
  if (g_tile_part) {      // one part of the launch: part 1 = the tiles that read nothing of an image of another rank's box
    int count = 0;
    A.order = tile_part_order(L, A.tiles_i, A.tiles_j, A.chunks_k, g_tile_part, false, &grid, &A.per_xcd, &count);
    if (grid == 0) return 0;
    if (!A.order) return record_error(hipErrorOutOfMemory, "smooth_gsrb27_rb: dispatch list of a partial launch");
    cells = cells * count / A.total_blocks;
  }
  const int prof = profile_begin(whole_cells);
  hipLaunchKernelGGL((stencil27_rb_kernel<TJ>), dim3(grid), dim3(64, TJ / 2), 0, g_stream, *L, A);
  g_rb27_launches++;
  profile_end(prof, 2 * cells, g_tile_part == 1);     // one launch = two half sweeps over every cell
  HPGMG_LAUNCH_CHECK("stencil27_rb_kernel");
  return 0;
}
// The same on small levels: one workgroup per box of 2^3 ... 16^3 cells (stencil27_rb_box.hpp); forms the boundary ghost cells of x_id itself,
// so the caller runs neither exchange_boundary nor apply_BCs_p2.  Every box local.
// Largest box side it takes.  Boxes of 16^3 and 32^3 are handled as cubes of 8^3 (bit-identical, tested) but measure no faster (16^3) or
// slower (32^3: 14.3 vs 14.2 ms per `7 64` F-cycle) than the launches they replace -- every cube re-reads a two-cell rim, 3.4 values per cell.
static int g_rb_box_maxdim = -1;
void hpgmg_hip_set_27pt_rb_box_maxdim(int dim) { g_rb_box_maxdim = dim; }
int hpgmg_hip_smooth_gsrb27_rb_box_supported(const hpgmg_hip_level *L) {
  static const int off = env_int("HPGMG_TUNE_27PT_NO_RB_BOX", 0);
  if (g_rb_box_maxdim < 0) g_rb_box_maxdim = env_int("HPGMG_TUNE_27PT_RB_BOX_MAXDIM", 8);
  const int maxdim = g_rb_box_maxdim;
  return !off && L->num_boxes > 0 && L->box_nbr != nullptr && L->ghosts >= 1 && L->dim <= maxdim && (L->dim == 2 || L->dim == 4 || L->dim == 8 || L->dim == 16 || L->dim == 32);
}
int hpgmg_hip_smooth_gsrb27_rb_box(const hpgmg_hip_level *L, int x_id, int out_id, int rhs_id, double a, double b, double h2inv, int sweep) {
  HPGMG_SKIP_IF_REPLAY();
  if (!hpgmg_hip_smooth_gsrb27_rb_box_supported(L) || x_id == out_id || (sweep & 1)) return record_error(hipErrorInvalidValue, "smooth_gsrb27_rb_box: level / arguments not supported");
  S27RbBoxArgs A = {}; A.xn_id = x_id; A.xout_id = out_id; A.rhs_id = rhs_id; A.a = a; A.b = b; A.h2inv = h2inv; A.sweep = sweep;
  // cubes of 8^3 for boxes of 8^3 and more (a whole box of 16^3 in one workgroup measured 33 us, slower than the four launches it replaces)
#define RB_BOX_CASE(DD, NTT) { \
    A.cubes = L->dim / DD; \
    const size_t lds = (size_t)((DD + 4) * (DD + 4) * (DD + 4) + (DD + 2) * (DD + 2) * (DD + 2)) * sizeof(double); \
    hipLaunchKernelGGL((stencil27_rb_box_kernel<DD, NTT>), dim3(L->num_boxes * A.cubes * A.cubes * A.cubes), dim3(NTT), lds, g_stream, *L, A); }
  if (L->dim >= 8) RB_BOX_CASE(8, 512)
  else if (L->dim == 4) RB_BOX_CASE(4, 256)
  else RB_BOX_CASE(2, 64)
#undef RB_BOX_CASE
  g_rb27_launches++;
  HPGMG_LAUNCH_CHECK("stencil27_rb_box_kernel");
  return 0;
}
int hpgmg_hip_smooth_jacobi(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                            double a, double b, double h2inv, double weight) {
  StencilArgs P = {}; P.xn_id = xn_id; P.xout_id = xnp1_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv; P.c2 = weight;
  return launch<MODE_JACOBI>(L, variant, P, true);
}
int hpgmg_hip_blackbox_accumulate(const hpgmg_hip_level *L, int variant, int x_id, int Aii_id, int sumAbs_id, double a, double b, double h2inv) {
  StencilArgs P = {}; P.xn_id = x_id; P.xout_id = Aii_id; P.rhs_id = sumAbs_id; P.a = a; P.b = b; P.h2inv = h2inv;
  if (variant == HPGMG_HIP_27PT_CC) return launch27<MODE_BLACKBOX>(L, P, false);
  return launch_direct<MODE_BLACKBOX>(L, variant, P, false);
}
static bool tiled_variant(int variant) { return variant == HPGMG_HIP_27PT_CC || variant == HPGMG_HIP_FV4_VC_HELMHOLTZ || variant == HPGMG_HIP_FV4_VC_POISSON; }
int hpgmg_hip_residual_fused_supported(const hpgmg_hip_level *L, int variant) {
  if (tiled_variant(variant)) return hpgmg_hip_tile_kernel_applies(L, variant, 1) && L->dim % 2 == 0;   // the tiled kernels carry the fused forms (27-point, fv4)
  return wide_fused_ok(L, variant);
}
// residual + restriction + zero_vector only (not residual + norm): also the launch-bound levels of small boxes
int hpgmg_hip_residual_restrict_supported(const hpgmg_hip_level *L, int variant) {
  return hpgmg_hip_residual_fused_supported(L, variant) || (!tiled_variant(variant) && small_fused_ok(L, variant));
}
// residual (never stored) -> restriction into vector coarse_id of Lc, plus zero_vector(Lc, zero_id) when zero_id >= 0: the end of
// MGVCycle's down leg (mg.c:1150-1153) in one pass over the fine level.  map[4 b .. 4 b + 3] = coarse box and coarse (i, j, k) under fine box b's first cell.
int hpgmg_hip_residual_restrict_store(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id, double a, double b, double h2inv,
                                      const hpgmg_hip_level *Lc, int coarse_id, const int *map, int zero_id);
int hpgmg_hip_residual_restrict(const hpgmg_hip_level *L, int variant, int x_id, int rhs_id, double a, double b, double h2inv,
                                const hpgmg_hip_level *Lc, int coarse_id, const int *map, int zero_id) {
  return hpgmg_hip_residual_restrict_store(L, variant, -1, x_id, rhs_id, a, b, h2inv, Lc, coarse_id, map, zero_id);
}
// res_id >= 0 (7-point only): the residual is stored as residual() would store it as well
int hpgmg_hip_residual_restrict_store(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id, double a, double b, double h2inv,
                                      const hpgmg_hip_level *Lc, int coarse_id, const int *map, int zero_id) {
  HPGMG_SKIP_IF_REPLAY();
  if (tiled_variant(variant)) {
    if (!hpgmg_hip_residual_fused_supported(L, variant) || Lc->num_boxes <= 0 || res_id >= 0) return record_error(hipErrorInvalidValue, "residual_restrict: level not supported");
    g_tile_fused = TileFused{}; g_tile_fused.kind = 2; g_tile_fused.Lc = *Lc; g_tile_fused.coarse_id = coarse_id; g_tile_fused.map = map;
    StencilArgs T = {}; T.xn_id = x_id; T.xout_id = rhs_id; T.rhs_id = rhs_id; T.a = a; T.b = b; T.h2inv = h2inv;    // xout only has to differ from xn: nothing is stored
    if (int e = launch<MODE_RESIDUAL>(L, variant, T, false)) return e;
    return zero_id >= 0 ? hpgmg_hip_fill(Lc, zero_id, 0.0) : 0;
  }
  if ((!wide_fused_ok(L, variant) && !small_fused_ok(L, variant)) || Lc->num_boxes <= 0) return record_error(hipErrorInvalidValue, "residual_restrict: level not supported");
  if (res_id == x_id || res_id == rhs_id) return record_error(hipErrorInvalidValue, "residual_restrict: the residual may not overwrite its inputs");
  StencilArgs P = {}; P.xn_id = x_id; P.xout_id = (res_id >= 0) ? res_id : x_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv;
  FusedArgs F = {}; F.Lc = *Lc; F.coarse_id = coarse_id; F.zero_id = zero_id; F.map = map; F.store_res = (res_id >= 0);
  F.zero_chunks_per_box = (Lc->volume + 4095) / 4096;
  if (!wide_fused_ok(L, variant)) return launch_small_fused(L, variant, P, F, zero_id >= 0 ? F.zero_chunks_per_box * Lc->num_boxes : 0);
  return launch_wide_fused<MODE_RESIDUAL_RESTRICT>(L, variant, P, F, zero_id >= 0 ? F.zero_chunks_per_box * Lc->num_boxes : 0);
}
// residual stored to res_id (not stored when res_id < 0) AND its max-abs: residual() + norm() of the convergence check (mg.c:1321-1323) in one pass
int hpgmg_hip_residual_norm(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id, double a, double b, double h2inv, double *norm_out) {
  if (int e = hpgmg_hip_graph_flush()) return e;
  *norm_out = 0.0;
  if (tiled_variant(variant)) {
    if (!hpgmg_hip_residual_fused_supported(L, variant) || res_id >= 0) return record_error(hipErrorInvalidValue, "residual_norm: level not supported (27-point / fv4: norm only, res_id < 0)");
    const long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
    double *part = reduction_scratch((int)(cells / 512 + 1024));      // one per workgroup: tiles of 512 cells, k chunks down to one plane (HPGMG_TUNE_*_KCHUNK)
    if (!part) return record_error(hipErrorOutOfMemory, "residual_norm: scratch");
    g_tile_fused = TileFused{}; g_tile_fused.kind = 1; g_tile_fused.partials = part;
    StencilArgs T = {}; T.xn_id = x_id; T.xout_id = rhs_id; T.rhs_id = rhs_id; T.a = a; T.b = b; T.h2inv = h2inv;
    if (int e = launch<MODE_RESIDUAL>(L, variant, T, false)) return e;
    return finish_max_reduction(g_tile_last_blocks, 0.0, norm_out);
  }
  if (!wide_fused_ok(L, variant)) return record_error(hipErrorInvalidValue, "residual_norm: level not supported");
  StencilArgs P = {}; P.xn_id = x_id; P.xout_id = res_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv;
  const int blocks = L->num_boxes * ((L->dim + 15) / 16) * (L->dim / 16) * (L->dim / 128);
  FusedArgs F = {}; F.partials = reduction_scratch(blocks); F.no_store = (res_id < 0);
  if (res_id < 0) P.xout_id = x_id;
  if (!F.partials) return record_error(hipErrorOutOfMemory, "residual_norm: scratch");
  if (int e = launch_wide_fused<MODE_RESIDUAL_NORM>(L, variant, P, F, 0)) return e;
  if (g_tile_part == 1) return 0;                 // the first part of a two-part launch: the tiles of the second part have not written their maxima yet
  return finish_max_reduction(blocks, 0.0, norm_out);
}

int hpgmg_hip_residual(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id, double a, double b, double h2inv) {
  StencilArgs P = {}; P.xn_id = x_id; P.xout_id = res_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv;
  if (rhs_id < 0) return launch<MODE_APPLY>(L, variant, P, false);
  return launch<MODE_RESIDUAL>(L, variant, P, false);
}

}  // extern "C"

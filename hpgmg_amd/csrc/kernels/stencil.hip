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
//   * cheby_pair_kernel (cheby_pair.hpp, included below): two sweeps per pass on the fine level;
//   * stencil7_shell_kernel: the cells next to faces owned by another rank, after the overlapped exchange;
//   * stencil27_kernel (27-point) and stencil_direct_kernel (4th-order fv4, black-box probes);
//   * logical tiles are ordered box, k, j, i and dealt to XCDs in contiguous ranges (common.hpp) so halo
//     planes shared by adjacent tiles hit the same L2.
#include <stdlib.h>
#include <cstring>
#include "common.hpp"
#include "stencil_math.hpp"
#include "cheby_pair.hpp"
#include "fv4_tile.hpp"
#include "stencil27_rb.hpp"
#include "stencil27_rb_box.hpp"
#include "stencil7_tile.hpp"
#include "block_ops.hpp"

namespace hpgmg {

enum { MODE_CHEBY = 0, MODE_GSRB = 1, MODE_JACOBI = 2, MODE_RESIDUAL = 3, MODE_APPLY = 4, MODE_BLACKBOX = 5,
       MODE_RESIDUAL_RESTRICT = 6,   // wide kernel only: the residual is not stored but restricted (restriction.c:54-57) into the coarse level
       MODE_RESIDUAL_NORM = 7 };     // wide kernel only: the residual is stored and its max-abs (misc.c:287-329) reduced per workgroup

// second kernel argument of the fused residual forms (unused by the plain modes)
struct FusedArgs {
  hpgmg_hip_level Lc;           // coarse level of MODE_RESIDUAL_RESTRICT
  int coarse_id, zero_id;       // restricted residual goes to vector coarse_id; zero_id >= 0: extra workgroups clear that coarse vector (zero_vector, misc.c:6-44)
  const int *map;               // per fine box: coarse box, and the (i, j, k) of the coarse cell under the fine box's first cell
  int zero_chunks_per_box, compute_blocks;
  double *partials;             // MODE_RESIDUAL_NORM: one max per workgroup
  int no_store;                 // MODE_RESIDUAL_NORM: the residual itself is not wanted, only its norm
  int store_res;                // MODE_RESIDUAL_RESTRICT: ALSO store the residual (to xout_id): the exact state of residual() + restriction()
};

struct StencilArgs {
  int xn_id, xout_id, rhs_id;   // xout = x_np1 (smoothers) or res/Ax
  double a, b, h2inv;
  double c1, c2;                // Chebyshev; Jacobi uses c2 = weight
  int sweep;                    // GSRB colour of this half sweep
  int copy_other_colour;        // GSRB out of place
  int tiles_i, tiles_j, chunks_k, kchunk, per_xcd, total_blocks;
  int ghost_free;               // read face neighbours from the adjacent box / apply the Dirichlet BC in registers
  int defer;                    // 1: leave the cells next to a face owned by another rank untouched (their ghost values are still
                                //    in flight); stencil7_shell_kernel computes them once the exchange has landed
};

// x may alias the output only for in-place GSRB; everywhere else it is restrict-qualified
// so the loads of plane k+1 can be issued ahead of the store of plane k.
template <bool kMayAlias> struct src_ptr { typedef const double *__restrict__ type; };
template <> struct src_ptr<true> { typedef const double *type; };

template <int V, int MODE>
__global__ __launch_bounds__(256) void stencil7_kernel(const hpgmg_hip_level L, const StencilArgs P) {
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

  int ijk = i + j * jS + k0 * kS;
  double xc = x[ijk];
  double xm = (gf && k0 == 0) ? outside(4, i + j * jS + last * kS, ijk - kS, xc) : x[ijk - kS];
  double bk0 = kVC ? beta_k[ijk] : 0.0;

  for (int k = k0; k < k1; k++, ijk += kS) {
    const double xp = (gf && k == last) ? outside(5, i + j * jS, ijk + kS, xc) : x[ijk + kS];
    const double bk1 = kVC ? beta_k[ijk + kS] : 0.0;
    bool update = true;
    if (MODE == MODE_GSRB) update = (((i ^ j ^ k ^ colour000) & 1) == 0);
    if (P.defer && (defer_ij || (k == 0 && defer_klo) || (k == last && defer_khi))) update = false;
    if (update) {
      const double xim = (gf && i == 0)    ? outside(0, last + j * jS + k * kS, ijk - 1, xc)  : x[ijk - 1];
      const double xip = (gf && i == last) ? outside(1, j * jS + k * kS, ijk + 1, xc)          : x[ijk + 1];
      const double xjm = (gf && j == 0)    ? outside(2, i + last * jS + k * kS, ijk - jS, xc) : x[ijk - jS];
      const double xjp = (gf && j == last) ? outside(3, i + k * kS, ijk + jS, xc)              : x[ijk + jS];
      double bi0 = 0.0, bi1 = 0.0, bj0 = 0.0, bj1 = 0.0, al = 0.0;
      if (kVC) { bi0 = beta_i[ijk]; bi1 = beta_i[ijk + 1]; bj0 = beta_j[ijk]; bj1 = beta_j[ijk + jS]; }
      if (kHelm) al = alpha[ijk];
      const double Ax = apply_op_7pt<V>(xc, xim, xip, xjm, xjp, xm, xp, bi0, bi1, bj0, bj1, bk0, bk1, al, P.a, P.b, P.h2inv);
      if (MODE == MODE_CHEBY) {
        const double xnm1 = out[ijk];
        out[ijk] = xc + P.c1 * (xc - xnm1) + P.c2 * dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_GSRB) {
        out[ijk] = xc + dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_JACOBI) {
        out[ijk] = xc + P.c2 * dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_RESIDUAL) {
        out[ijk] = rhs[ijk] - Ax;
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
  const int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
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
}  // namespace hpgmg
#include "stencil27_tile.hpp"   // C27_* constants + the LDS-staged kernel for boxes of side 64 m
namespace hpgmg {
struct plane9 { double v[3][3]; };   // [dj+1][di+1]
template <typename P>
__device__ __forceinline__ plane9 load_plane(P p, int jS) {
  plane9 q;
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
#pragma unroll
    for (int ii = 0; ii < 3; ii++) q.v[jj][ii] = p[(ii - 1) + (jj - 1) * jS];
  }
  return q;
}
__device__ __forceinline__ double apply_op_27pt(const plane9 &m, const plane9 &c, const plane9 &p, double a, double b, double h2inv) {
  double s8 = m.v[0][0] + m.v[0][2]; s8 = s8 + m.v[2][0]; s8 = s8 + m.v[2][2];
  s8 = s8 + p.v[0][0]; s8 = s8 + p.v[0][2]; s8 = s8 + p.v[2][0]; s8 = s8 + p.v[2][2];
  double s12 = m.v[0][1] + m.v[1][0]; s12 = s12 + m.v[1][2]; s12 = s12 + m.v[2][1];
  s12 = s12 + c.v[0][0]; s12 = s12 + c.v[0][2]; s12 = s12 + c.v[2][0]; s12 = s12 + c.v[2][2];
  s12 = s12 + p.v[0][1]; s12 = s12 + p.v[1][0]; s12 = s12 + p.v[1][2]; s12 = s12 + p.v[2][1];
  double s6 = m.v[1][1] + c.v[0][1]; s6 = s6 + c.v[1][0]; s6 = s6 + c.v[1][2]; s6 = s6 + c.v[2][1]; s6 = s6 + p.v[1][1];
  double t = C27_3 * s8 + C27_2 * s12;
  t = t + C27_1 * s6;
  t = t + C27_0 * c.v[1][1];
  return a * c.v[1][1] - (b * h2inv) * t;
}

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


// ---------------------------------------------------------------------------------------------
// Direct-load kernel: one lane per (i,j) column walking +k, every operand read through L1/L2.
// Used for the 4th-order finite-volume operator (reference operators.fv4.c:55-114: radius 2,
// 25 x values and 30 coefficient values per cell, so the neighbourhood does not fit a register
// window worth keeping) and for the black-box probe of the 7-point operator (fv2 rebuild).
// Expression order is the macro's: T*(six face terms) + (0.25*T)*(twelve mixed terms), each
// group summed left to right; a mixed term is (beta+ - beta-) * (((x1 - x2) - x3) + x4).
#define FV4_TWELFTH ( 0.0833333333333333333)
// The fv4 expression below with its 55 operands read in batches (a scheduling fence after each) instead of where the
// expression names them (three batches: 19 + 24 + 12): out of LDS the compiler otherwise issues them one to three at a time, each group a round trip of its own -- ~30 of
// them per cell, which was the whole stencil phase of the single-workgroup kernels.  Same expression tree, term by term (MIX(B, o, t, d) =
// (B[o + t] - B[o - t]) * (((x[d + t] - x[t]) - x[d - t]) + x[-t])).
template <int V, typename XP, typename CP>
__device__ __forceinline__ double apply_op_fv4_batched(XP x, CP alpha, CP bi, CP bj, CP bk, int ijk, int jS, int kS, double a, double b, double h2inv) {
  const double xc = x[ijk], xm1 = x[ijk - 1], xp1 = x[ijk + 1], xm2 = x[ijk - 2], xp2 = x[ijk + 2];
  const double xmj = x[ijk - jS], xpj = x[ijk + jS], xm2j = x[ijk - 2 * jS], xp2j = x[ijk + 2 * jS];
  const double xmk = x[ijk - kS], xpk = x[ijk + kS], xm2k = x[ijk - 2 * kS], xp2k = x[ijk + 2 * kS];
  const double bi0 = bi[ijk], bi1 = bi[ijk + 1], bj0 = bj[ijk], bj1 = bj[ijk + jS], bk0 = bk[ijk], bk1 = bk[ijk + kS];
  const double al = (V == HPGMG_HIP_FV4_VC_HELMHOLTZ) ? alpha[ijk] : 0.0;
  __builtin_amdgcn_sched_barrier(0);
  double s1 = bi0 * (15.0 * (xm1 - xc) - (xm2 - xp1));
  s1 = s1 + bi1 * (15.0 * (xp1 - xc) - (xp2 - xm1));
  s1 = s1 + bj0 * (15.0 * (xmj - xc) - (xm2j - xpj));
  s1 = s1 + bj1 * (15.0 * (xpj - xc) - (xp2j - xmj));
  s1 = s1 + bk0 * (15.0 * (xmk - xc) - (xm2k - xpk));
  s1 = s1 + bk1 * (15.0 * (xpk - xc) - (xp2k - xmk));
  __builtin_amdgcn_sched_barrier(0);
  double xij[2][2], xik[2][2], xjk[2][2];                                   // x[(+-1) + (+-jS)], x[(+-1) + (+-kS)], x[(+-jS) + (+-kS)]: index 0 = minus, 1 = plus
  double bi_j[2], bi_k[2], bi1_j[2], bi1_k[2], bj_i[2], bj_k[2], bj1_i[2], bj1_k[2], bk_i[2], bk_j[2], bk1_i[2], bk1_j[2];   // B[o +- t]
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int sp = p ? 1 : -1;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int sq = q ? 1 : -1;
      xij[p][q] = x[ijk + sp + sq * jS]; xik[p][q] = x[ijk + sp + sq * kS]; xjk[p][q] = x[ijk + sp * jS + sq * kS];
    }
    bi_j[p] = bi[ijk + sp * jS]; bi_k[p] = bi[ijk + sp * kS];
    bj_i[p] = bj[ijk + sp]; bj_k[p] = bj[ijk + sp * kS];
    bk_i[p] = bk[ijk + sp]; bk_j[p] = bk[ijk + sp * jS];
  }
  __builtin_amdgcn_sched_barrier(0);
  double s2 = (bi_j[1] - bi_j[0]) * (xij[0][1] - xpj - xij[0][0] + xmj);
  s2 = s2 + (bi_k[1] - bi_k[0]) * (xik[0][1] - xpk - xik[0][0] + xmk);
  s2 = s2 + (bj_i[1] - bj_i[0]) * (xij[1][0] - xp1 - xij[0][0] + xm1);
  s2 = s2 + (bj_k[1] - bj_k[0]) * (xjk[0][1] - xpk - xjk[0][0] + xmk);
  s2 = s2 + (bk_i[1] - bk_i[0]) * (xik[1][0] - xp1 - xik[0][0] + xm1);
  s2 = s2 + (bk_j[1] - bk_j[0]) * (xjk[1][0] - xpj - xjk[0][0] + xmj);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 2; p++) {                                               // third batch: the coefficients of the far faces (1024 lanes: 128 registers each)
    const int sp = p ? 1 : -1;
    bi1_j[p] = bi[ijk + 1 + sp * jS]; bi1_k[p] = bi[ijk + 1 + sp * kS];
    bj1_i[p] = bj[ijk + jS + sp]; bj1_k[p] = bj[ijk + jS + sp * kS];
    bk1_i[p] = bk[ijk + kS + sp]; bk1_j[p] = bk[ijk + kS + sp * jS];
  }
  __builtin_amdgcn_sched_barrier(0);
  s2 = s2 + (bi1_j[1] - bi1_j[0]) * (xij[1][1] - xpj - xij[1][0] + xmj);
  s2 = s2 + (bi1_k[1] - bi1_k[0]) * (xik[1][1] - xpk - xik[1][0] + xmk);
  s2 = s2 + (bj1_i[1] - bj1_i[0]) * (xij[1][1] - xp1 - xij[0][1] + xm1);
  s2 = s2 + (bj1_k[1] - bj1_k[0]) * (xjk[1][1] - xpk - xjk[1][0] + xmk);
  s2 = s2 + (bk1_i[1] - bk1_i[0]) * (xik[1][1] - xp1 - xik[0][1] + xm1);
  s2 = s2 + (bk1_j[1] - bk1_j[0]) * (xjk[1][1] - xpj - xjk[0][1] + xmj);
  const double sum = FV4_TWELFTH * s1 + (0.25 * FV4_TWELFTH) * s2;
  if (V == HPGMG_HIP_FV4_VC_HELMHOLTZ) return (a * al) * xc - (b * h2inv) * sum;
  return ((-b) * h2inv) * sum;
}
// (pointer types are template parameters: the single-workgroup kernels pass LDS pointers for the vectors they hold an image of, and BATCH)
template <int V, typename XP, typename CP, bool BATCH = false>
__device__ __forceinline__ double apply_op_direct(XP x, CP alpha, CP bi, CP bj, CP bk,
                                                  int ijk, int jS, int kS, double a, double b, double h2inv) {
  if constexpr (BATCH && (V == HPGMG_HIP_FV4_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_POISSON)) return apply_op_fv4_batched<V>(x, alpha, bi, bj, bk, ijk, jS, kS, a, b, h2inv);
  const double xc = x[ijk];
  if (V == HPGMG_HIP_FV4_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_POISSON) {
    double s1 = bi[ijk] * (15.0 * (x[ijk - 1] - xc) - (x[ijk - 2] - x[ijk + 1]));
    s1 = s1 + bi[ijk + 1] * (15.0 * (x[ijk + 1] - xc) - (x[ijk + 2] - x[ijk - 1]));
    s1 = s1 + bj[ijk] * (15.0 * (x[ijk - jS] - xc) - (x[ijk - 2 * jS] - x[ijk + jS]));
    s1 = s1 + bj[ijk + jS] * (15.0 * (x[ijk + jS] - xc) - (x[ijk + 2 * jS] - x[ijk - jS]));
    s1 = s1 + bk[ijk] * (15.0 * (x[ijk - kS] - xc) - (x[ijk - 2 * kS] - x[ijk + kS]));
    s1 = s1 + bk[ijk + kS] * (15.0 * (x[ijk + kS] - xc) - (x[ijk + 2 * kS] - x[ijk - kS]));
#define MIX(B, o, t, d) ((B[ijk + (o) + (t)] - B[ijk + (o) - (t)]) * (x[ijk + (d) + (t)] - x[ijk + (t)] - x[ijk + (d) - (t)] + x[ijk - (t)]))
    double s2 = MIX(bi, 0, jS, -1);
    s2 = s2 + MIX(bi, 0, kS, -1);
    s2 = s2 + MIX(bj, 0, 1, -jS);
    s2 = s2 + MIX(bj, 0, kS, -jS);
    s2 = s2 + MIX(bk, 0, 1, -kS);
    s2 = s2 + MIX(bk, 0, jS, -kS);
    s2 = s2 + MIX(bi, 1, jS, 1);
    s2 = s2 + MIX(bi, 1, kS, 1);
    s2 = s2 + MIX(bj, jS, 1, jS);
    s2 = s2 + MIX(bj, jS, kS, jS);
    s2 = s2 + MIX(bk, kS, 1, kS);
    s2 = s2 + MIX(bk, kS, jS, kS);
#undef MIX
    const double sum = FV4_TWELFTH * s1 + (0.25 * FV4_TWELFTH) * s2;
    if (V == HPGMG_HIP_FV4_VC_HELMHOLTZ) return (a * alpha[ijk]) * xc - (b * h2inv) * sum;
    return ((-b) * h2inv) * sum;
  } else {
    constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
    return apply_op_7pt<V>(xc, x[ijk - 1], x[ijk + 1], x[ijk - jS], x[ijk + jS], x[ijk - kS], x[ijk + kS],
                           kVC ? bi[ijk] : 0.0, kVC ? bi[ijk + 1] : 0.0, kVC ? bj[ijk] : 0.0, kVC ? bj[ijk + jS] : 0.0,
                           kVC ? bk[ijk] : 0.0, kVC ? bk[ijk + kS] : 0.0, (V == HPGMG_HIP_7PT_VC_HELMHOLTZ) ? alpha[ijk] : 0.0, a, b, h2inv);
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

// ---------------------------------------------------------------------------------------------
// Small levels (<= 4096 cells) of the 27-point / fv2 / fv4 plugins: a whole smooth() -- per sweep exchange_boundary (local copies),
// apply_BCs and the stencil, the three launches operators.27pt.c / .fv2.c / .fv4.c sequence per sweep -- or a whole residual() /
// apply_op() as ONE single-workgroup launch, with barriers where the per-operator path has kernel boundaries.  On these levels a
// launch costs more than the work (fv4: 18 launches of ~6 us per smooth()); the arithmetic is the same entry routines and the same
// per-cell expressions as the streaming kernels, so results stay bit-identical.  (The 7-point plugin has its own, LDS-resident form
// of this idea: tail.hip.)
struct SmallArgs {
  int mode, sweeps, x_id, rhs_id, res_id, out_of_place, bc_kind, zero_first;   // bc_kind: 0 none (periodic), 1 p1, 2 p2, 3 v2, 4 v4
  double a, b, h2inv, c1[8], c2[8];
  const blockCopy_type *copy_list; int n_copy;
  const blockCopy_type *bc_list; int n_bc;
  int lds_resident;                 // a level of ONE box whose vectors fit the LDS: work on an image of the box there (see the kernel)
  unsigned long long *timeline;     // experiment builds (-DHPGMG_EXP_TIMELINE): lane 0 records the clock at the phase boundaries
};
// LDS-resident form (round 3): out of global memory every boundary entry and every stencil read of this one workgroup is a round trip to
// the L2 that nothing hides (measured: slower than the dozen launches it replaces, even on a level of one box).  For a level of ONE box
// the vectors the operator touches -- x, VECTOR_TEMP, rhs, Dinv, alpha, beta_i/j/k, the result -- are copied into LDS with the box's own
// padded layout, the level descriptor is pointed at that image (a one-entry box table in LDS, vector ids renumbered to slots), the very same
// entry routines and per-cell expressions run on it, and the vectors written go back to memory at the end: bit-identical by construction.
constexpr int kSmallSlots = 9;
typedef double __attribute__((address_space(3))) *lds_dptr;       // a pointer into LDS, typed as such: ds_read / ds_write, not FLAT
typedef const int __attribute__((address_space(3))) *lds_iptr;
// a pointer into global memory, typed as such: through a generic pointer a load is a FLAT instruction, which may address LDS -- the compiler
// then keeps every such load and every LDS store of a copy loop in program order, one round trip to memory per element
typedef double __attribute__((address_space(1))) *gbl_dptr;
typedef const double __attribute__((address_space(1))) *gbl_cdptr;
// the eight words of a boundary entry that the entry routines read (subtype, dim, read.box / i / j / k), kept in LDS by the single-workgroup
// kernels: read from the level's list in memory, the descriptor was a round trip per entry and half sweep
// (n <= 32.)  The entries are stored SORTED by kind -- corners, edges, faces; their order is immaterial, every entry reads the interior and writes
// ghost cells of its own -- and words[256..258] hold the three counts: the packed dispatch of apply_BCs_v4 below hands out lanes by kind.
constexpr int kBcWords = 32 * 8 + 4;
__device__ __forceinline__ void lds_bc_words_fill(int *words, const blockCopy_type *list, int n, int tid, int nth, bool sorted) {
  if (!sorted) {                                                  // every lane fetches a word (the sorted form is a wave's serial work: only where it pays)
    for (int t = tid; t < 8 * n; t += nth) {
      const blockCopy_type &g = list[t >> 3];
      const int f = t & 7;
      words[t] = f == 0 ? g.subtype : f == 1 ? g.dim.i : f == 2 ? g.dim.j : f == 3 ? g.dim.k : f == 4 ? g.read.box : f == 5 ? g.read.i : f == 6 ? g.read.j : g.read.k;
    }
    return;
  }
  if (tid >= 64) return;                                          // the first wave: a lane per entry, the positions by ballot
  const bool have = tid < n;
  int w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nn = 0;
  if (have) {
    const blockCopy_type &g = list[tid];
    w[0] = g.subtype; w[1] = g.dim.i; w[2] = g.dim.j; w[3] = g.dim.k; w[4] = g.read.box; w[5] = g.read.i; w[6] = g.read.j; w[7] = g.read.k;
    nn = (w[0] % 3 != 1) + ((w[0] % 9) / 3 != 1) + (w[0] / 9 != 1);
  }
  const unsigned long long m3 = __ballot(have && nn == 3), m2 = __ballot(have && nn == 2), m1 = __ballot(have && nn < 2);
  const unsigned long long below = (1ull << tid) - 1ull;
  const int n3 = __popcll(m3), n2 = __popcll(m2);
  const int pos = (nn == 3) ? __popcll(m3 & below) : (nn == 2) ? n3 + __popcll(m2 & below) : n3 + n2 + __popcll(m1 & below);
  if (have) {
#pragma unroll
    for (int f = 0; f < 8; f++) words[8 * pos + f] = w[f];
  }
  if (tid == 0) { words[256] = n3; words[257] = n2; words[258] = __popcll(m1); }
}
__device__ __forceinline__ blockCopy_type lds_bc_entry(const int *words, int e) {
  const lds_iptr w = (lds_iptr)words + 8 * e;
  blockCopy_type en;
  en.subtype = w[0]; en.dim.i = w[1]; en.dim.j = w[2]; en.dim.k = w[3]; en.read.box = w[4]; en.read.i = w[5]; en.read.j = w[6]; en.read.k = w[7];
  return en;
}
// The sweeps of one launch.  RES: the vectors live in the LDS image (`image`; vector "ids" are slots of it, the only box is box 0) and every
// access to them is an LDS instruction -- through generic pointers each was a FLAT access, and a corner entry of apply_BCs_v4 (64 dependent
// reads by one lane) or the 55 reads of a stencil took microseconds: 6.1 + 3.8 us per half sweep of an 8^3 level, 66 us per smooth().
// coef(s, c1, c2): the Chebyshev / Jacobi coefficients of sweep s (a functor: the caller knows where they live -- kernel arguments, memory)
template <int V, bool RES, typename CoefFn>
__device__ __forceinline__ void small_level_run(const hpgmg_hip_level &L, const SmallArgs &A, double *image, const blockCopy_type *bc_entries, const int *bc_words, const int *ids, unsigned long long *tl, int &tl_n, CoefFn coef) {
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride, per_box = dim * dim * dim, total = per_box * L.num_boxes;
  const size_t vol = (size_t)L.volume, first = (size_t)L.ghosts * (size_t)(1 + jS + kS);
  // origin of vector `id` of box `box`: a slot of the image (LDS) or the level's own storage
  auto vo = [&](int box, int id) {
    if constexpr (RES) { (void)box; return (lds_dptr)image + ((size_t)id * vol + first); }
    else return vec_origin(L, box, id);
  };
  // boundary entry e: from the level's list, or (RES, bc_words != 0) from the eight words of it that the entry routines read, kept in LDS
  auto entry = [&](int e) {
    if constexpr (RES) {
      if (bc_words) return lds_bc_entry(bc_words, e);
    }
    return bc_entries[e];
  };
  // the colour of cell (0,0,0) of the one box of an image (read once: every cell of every sweep asked the level for it)
  int low_parity = 0;
  if constexpr (RES) low_parity = L.box_low[0] ^ L.box_low[1] ^ L.box_low[2];
  const int x_id = ids[0], temp_id = ids[1], rhs_id = ids[2], dinv_id = ids[3], al_id = ids[4], bi_id = ids[5], bj_id = ids[6], bk_id = ids[7], res_id = ids[8];
#ifdef HPGMG_EXP_TIMELINE
#define SL_MARK() do { if (tl && threadIdx.x == 0 && tl_n < 250) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SL_MARK() do { (void)tl; (void)tl_n; } while (0)
#endif
  // (an image is one box of at most 512 cells, a cell per lane: its coordinates are worked out once, not per sweep -- four integer divisions)
  const int own_k = tid / (dim * dim), own_j = (tid / dim) % dim, own_i = tid % dim;
  for (int s = 0; s < A.sweeps; s++) {
    double cs1 = 0.0, cs2 = 0.0;
    coef(s, cs1, cs2);
    int src = x_id, dst = res_id;
    if (A.mode == MODE_CHEBY || A.mode == MODE_JACOBI || (A.mode == MODE_GSRB && A.out_of_place)) { src = (s & 1) ? temp_id : x_id; dst = (s & 1) ? x_id : temp_id; }
    else if (A.mode == MODE_GSRB) { src = x_id; dst = x_id; }
    // exchange_boundary(src): box -> box copies (blockCopy.c:6-105); an image is one box: nothing to copy
    if (!RES) { for (int e = wave; e < A.n_copy; e += nwaves) copy_entry<false>(L, src, A.copy_list[e], 0.0, lane, 64); __syncthreads(); }
    SL_MARK();
    // apply_BCs(src)
    if (A.bc_kind && A.zero_first) { for (int e = wave; e < A.n_bc; e += nwaves) { const blockCopy_type en = entry(e); bc_zero_entry_at(vo(en.read.box, src), L, en, lane, 64); } __syncthreads(); }
    bool packed = false;
    if constexpr (RES) packed = (bc_words != nullptr && A.bc_kind == 4);
    if (packed) {
      // apply_BCs_v4 with the lanes handed out by kind: 16 per corner, 32 per edge, 64 per face -- the 26 entries of a box are 896 lanes of
      // work, one round of a 1024-lane workgroup (two of a 512-lane one) instead of a wave per entry (two / four rounds)
      const lds_iptr cnt = (lds_iptr)bc_words + 256;
      const int n3 = cnt[0], n2 = cnt[1], n1 = cnt[2], lim3 = n3 * 16, lim2 = lim3 + n2 * 32, demand = lim2 + n1 * 64;
      for (int g0 = 0; g0 < demand; g0 += (int)blockDim.x) {
        const int g = g0 + tid;
        if (g < lim3)        bc_v4_entry_packed(vo(0, src), L, lds_bc_entry(bc_words, g >> 4), g & 15, lane);
        else if (g < lim2)   bc_v4_entry_packed(vo(0, src), L, lds_bc_entry(bc_words, n3 + ((g - lim3) >> 5)), (g - lim3) & 31, lane);
        else if (g < demand) bc_v4_entry_packed(vo(0, src), L, lds_bc_entry(bc_words, n3 + n2 + ((g - lim2) >> 6)), (g - lim2) & 63, lane);
      }
    }
    for (int e = wave; e < A.n_bc && !packed; e += nwaves) {
      const blockCopy_type en = entry(e);
      if (A.bc_kind == 1) bc_p1_entry_at(vo(en.read.box, src), L, en, lane, 64);
      else if (A.bc_kind == 2) bc_p2_entry_at(vo(en.read.box, src), L, en, lane, 64);
      else if (A.bc_kind == 3) bc_v2_entry_at(vo(en.read.box, src), L, en, lane, 64);
      else if (A.bc_kind == 4) bc_v4_entry_at(vo(en.read.box, src), L, en, lane, 64);
    }
    __syncthreads();
    SL_MARK();
    // the stencil over every cell (same expressions as stencil_direct_kernel / stencil27_kernel)
    for (int t = tid; t < total; t += (int)blockDim.x) {
      int box = 0, i = own_i, j = own_j, k = own_k;
      if (!RES || total > (int)blockDim.x) { box = t / per_box; const int r = t - box * per_box; k = r / (dim * dim); j = (r / dim) % dim; i = r % dim; }
      const int ijk = i + j * jS + k * kS;
      auto x = vo(box, src);
      auto out = vo(box, dst);
      const double xc = x[ijk];
      bool update = true;
      if (A.mode == MODE_GSRB) {
        const int lp = RES ? low_parity : (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2]);
        update = (((i ^ j ^ k ^ lp ^ s) & 1) == 0);
      }
      if (!update) { if (A.out_of_place) out[ijk] = xc; continue; }
      double Ax;
      if (k27) {
        const plane9 m = load_plane(x + (ijk - kS), jS), c = load_plane(x + ijk, jS), p = load_plane(x + (ijk + kS), jS);
        Ax = apply_op_27pt(m, c, p, A.a, A.b, A.h2inv);
      } else {
        auto none = vo(box, src); none = nullptr;
        // (not the batched form here: with 1024 lanes a lane has 128 registers, the batches then spill, and what the stencil phase gains the
        // boundary phase loses to the reloads -- measured 1.44 + 1.9 us against 1.8 + 1.7 us per half sweep)
        Ax = apply_op_direct<V, decltype(x), decltype(none), false>(x, kHelm ? vo(box, al_id) : none, kVC ? vo(box, bi_id) : none, kVC ? vo(box, bj_id) : none, kVC ? vo(box, bk_id) : none,
                                                                   ijk, jS, kS, A.a, A.b, A.h2inv);
      }
      if (A.mode == MODE_APPLY) { out[ijk] = Ax; continue; }
      const double rhs = vo(box, rhs_id)[ijk];
      if (A.mode == MODE_RESIDUAL) { out[ijk] = rhs - Ax; continue; }
      const double dinv = vo(box, dinv_id)[ijk];
      if (A.mode == MODE_CHEBY)      { const double xnm1 = out[ijk]; out[ijk] = xc + cs1 * (xc - xnm1) + cs2 * dinv * (rhs - Ax); }
      else if (A.mode == MODE_GSRB)  { out[ijk] = xc + dinv * (rhs - Ax); }
      else                           { out[ijk] = xc + cs2 * dinv * (rhs - Ax); }
    }
    __syncthreads();
    SL_MARK();
  }
#undef SL_MARK
}
template <int V>
__global__ __launch_bounds__(1024) void small_level_kernel(const hpgmg_hip_level L, const SmallArgs A) {
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  extern __shared__ double small_lds[];
  const int tid = (int)threadIdx.x;
  int tl_n = 0;
  unsigned long long *tl = nullptr;
#ifdef HPGMG_EXP_TIMELINE
  tl = A.timeline;
  if (tl && tid == 0) tl[tl_n++] = __builtin_amdgcn_s_memrealtime();
#endif
  if (!A.lds_resident) return;                                    // (the launcher only starts it on one box whose vectors fit the LDS)
  // ---- one box, its vectors in LDS: copy in, run, copy what was written back (ghost zones included: the boundary entries filled them)
  const bool smooth = (A.mode == MODE_CHEBY || A.mode == MODE_JACOBI || A.mode == MODE_GSRB);
  const bool uses_temp = smooth && !(A.mode == MODE_GSRB && !A.out_of_place);
  int slot_of[kSmallSlots];                                        // level vector id held in each slot (-1: unused)
  slot_of[0] = A.x_id; slot_of[1] = uses_temp ? VECTOR_TEMP : -1; slot_of[2] = (A.mode == MODE_APPLY) ? -1 : A.rhs_id; slot_of[3] = smooth ? VECTOR_DINV : -1;
  slot_of[4] = kHelm ? VECTOR_ALPHA : -1; slot_of[5] = kVC ? VECTOR_BETA_I : -1; slot_of[6] = kVC ? VECTOR_BETA_J : -1; slot_of[7] = kVC ? VECTOR_BETA_K : -1;
  slot_of[8] = (smooth || A.res_id == A.x_id) ? -1 : A.res_id;
  const size_t vol = (size_t)L.volume;
  lds_dptr img = (lds_dptr)small_lds;
  // the boundary entries of the box (26 at most) wait in LDS too: read from memory, the descriptor was a round trip per entry and half sweep
  __shared__ int s_bc[kBcWords];
  const bool bc_in_lds = A.n_bc <= 32;
  if (bc_in_lds) lds_bc_words_fill(s_bc, A.bc_list, A.n_bc, tid, (int)blockDim.x, A.bc_kind == 4);
#pragma unroll
  for (int q = 0; q < kSmallSlots; q++) {
    if (slot_of[q] < 0) continue;
    // (a result vector that is written in full needs no load, but its ghost zone must come back as it was: copy it all the same)
    const gbl_cdptr g = (gbl_cdptr)(L.box_base[0] + (size_t)slot_of[q] * vol);
#pragma unroll 8
    for (int t = tid; t < (int)vol; t += (int)blockDim.x) img[(size_t)q * vol + t] = g[t];      // unrolled: eight loads in flight per lane, not one
  }
  __syncthreads();
#ifdef HPGMG_EXP_TIMELINE
  if (tl && tid == 0) tl[tl_n++] = __builtin_amdgcn_s_memrealtime();
#endif
  const int ids[kSmallSlots] = { 0, 1, 2, 3, 4, 5, 6, 7, (smooth || A.res_id == A.x_id) ? 0 : 8 };
  small_level_run<V, true>(L, A, small_lds, A.bc_list, bc_in_lds ? (const int *)s_bc : nullptr, ids, tl, tl_n, [&](int s, double &c1, double &c2) { c1 = A.c1[s]; c2 = A.c2[s]; });
#pragma unroll
  for (int q = 0; q < kSmallSlots; q++) {
    const bool written = smooth ? (q == 0 || (q == 1 && slot_of[1] >= 0)) : (q == 0 || q == 8);      // x's ghost zone was filled too
    if (!written || slot_of[q] < 0) continue;
    const gbl_dptr g = (gbl_dptr)(L.box_base[0] + (size_t)slot_of[q] * vol);
#pragma unroll 8
    for (int t = tid; t < (int)vol; t += (int)blockDim.x) g[t] = img[(size_t)q * vol + t];
  }
#ifdef HPGMG_EXP_TIMELINE
  if (tl && tid == 0) { tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); tl[255] = (unsigned long long)tl_n; }
#endif
}


// ---------------------------------------------------------------------------------------------
// Bottom solve of the 27-point / fv2 / fv4 plugins: diagonally preconditioned BiCGStab (solvers/bicgstab.c:14-97) on a bottom level of ONE
// box as one single-workgroup launch.  Driven from the host it is ~25 launches and ~6 host round trips (the dot products and norms) per
// iteration on a level of 8 cells: 1.5 ms of a 33 ms fv4 F-cycle at 512^3, a quarter of one at 128^3.  One cell per lane, every vector a
// register; the vector the operator is applied to passes through an image of the padded box in LDS, on which the boundary entries of the
// level run (the same entry routines as the streaming kernels) before the stencil (the same per-cell expression).  The operation sequence,
// the expression of every BLAS-1 step (misc.c: c = sa*a + sb*b, c = s*a*b), the break-down tests and the order of the sums -- one partial
// per dim x 8 x 8 tile accumulated k, j, i, partials added in tile order (misc.c:261-269) -- are those of host/solvers.c, so the iterates,
// the iteration count and the coarse correction are bit-identical to the host-driven solve (the 7-point plugin's form of this: tail.hip).
struct BottomArgs {
  int e_id, R_id, krylov_base, bc_kind, zero_first, n_bc;
  double a, b, h2inv, want;
  const blockCopy_type *bc_list;
  int *krylov_iterations;
};
template <int V>
__device__ __forceinline__ void bottom_bicgstab_body(const hpgmg_hip_level &L, const BottomArgs &A) {
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  extern __shared__ double bb_lds[];                               // images of the padded box (5 x L.volume doubles: v, alpha, beta_i/j/k), then the reduction scratch
  __shared__ int s_bc[kBcWords];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride, total = dim * dim * dim;
  const bool active = tid < total;
  const int ci = tid % dim, cj = (tid / dim) % dim, ck = tid / (dim * dim), ijk = ci + cj * jS + ck * kS;
  // Every access to the images is an LDS instruction (pointers typed as such; through generic pointers each was a FLAT access), the
  // coefficients wait there too (read from memory they were a round trip to the L2 per apply), and so do the boundary entries.
  const int vol = L.volume, first = L.ghosts * (1 + jS + kS);
  const lds_dptr img = (lds_dptr)bb_lds;
  const lds_dptr xi = img + first;
  const lds_dptr alpha = img + (vol + first), bi = img + (2 * vol + first), bj = img + (3 * vol + first), bk = img + (4 * vol + first);
  const lds_dptr scr = img + 5 * vol;
  const bool bc_in_lds = A.n_bc <= 32;
  if (bc_in_lds) lds_bc_words_fill(s_bc, A.bc_list, A.n_bc, tid, (int)blockDim.x, false);
  {
    const gbl_cdptr g_al = (gbl_cdptr)(L.box_base[0] + (size_t)VECTOR_ALPHA * vol), g_bi = (gbl_cdptr)(L.box_base[0] + (size_t)VECTOR_BETA_I * vol);
    const gbl_cdptr g_bj = (gbl_cdptr)(L.box_base[0] + (size_t)VECTOR_BETA_J * vol), g_bk = (gbl_cdptr)(L.box_base[0] + (size_t)VECTOR_BETA_K * vol);
    for (int t = tid; t < vol; t += (int)blockDim.x) {
      img[t] = 0.0;
      if (kHelm) img[vol + t] = g_al[t];
      if (kVC) { img[2 * vol + t] = g_bi[t]; img[3 * vol + t] = g_bj[t]; img[4 * vol + t] = g_bk[t]; }
    }
  }
  __syncthreads();
  const int r0_id = A.krylov_base, r_id = r0_id + 1, p_id = r0_id + 2, q_id = r0_id + 3, s_id = r0_id + 4, t_id = r0_id + 5, Ap_id = r0_id + 6, As_id = r0_id + 7;
  double x = 0, r0 = 0, r = 0, p = 0, q = 0, sv = 0, tv = 0, Ap = 0, As = 0, tmp = 0, dinv = 0, rhs = 0;
  if (active) {
    x = vec_origin(L, 0, A.e_id)[ijk]; rhs = vec_origin(L, 0, A.R_id)[ijk]; dinv = vec_origin(L, 0, VECTOR_DINV)[ijk];
    r0 = vec_origin(L, 0, r0_id)[ijk]; r = vec_origin(L, 0, r_id)[ijk]; p = vec_origin(L, 0, p_id)[ijk]; q = vec_origin(L, 0, q_id)[ijk];
    sv = vec_origin(L, 0, s_id)[ijk]; tv = vec_origin(L, 0, t_id)[ijk]; Ap = vec_origin(L, 0, Ap_id)[ijk]; As = vec_origin(L, 0, As_id)[ijk];
    tmp = vec_origin(L, 0, VECTOR_TEMP)[ijk];
  }
  auto entry = [&](int e) { if (bc_in_lds) return lds_bc_entry(s_bc, e); return A.bc_list[e]; };
  // apply_op(v): exchange_boundary (nothing to exchange: one box) + apply_BCs + the stencil (operators.*.c: apply_op)
  auto apply = [&](double v) -> double {
    if (active) xi[ijk] = v;
    __syncthreads();
    if (A.bc_kind && A.zero_first) { for (int e = wave; e < A.n_bc; e += nwaves) bc_zero_entry_at(xi, L, entry(e), lane, 64); __syncthreads(); }
    for (int e = wave; e < A.n_bc; e += nwaves) {
      const blockCopy_type en = entry(e);
      if (A.bc_kind == 1) bc_p1_entry_at(xi, L, en, lane, 64);
      else if (A.bc_kind == 2) bc_p2_entry_at(xi, L, en, lane, 64);
      else if (A.bc_kind == 3) bc_v2_entry_at(xi, L, en, lane, 64);
      else if (A.bc_kind == 4) bc_v4_entry_at(xi, L, en, lane, 64);
    }
    __syncthreads();
    double Ax = 0.0;
    if (active) {
      if (k27) {
        const plane9 m = load_plane(xi + (ijk - kS), jS), c = load_plane(xi + ijk, jS), pp = load_plane(xi + (ijk + kS), jS);
        Ax = apply_op_27pt(m, c, pp, A.a, A.b, A.h2inv);
      } else {
        Ax = apply_op_direct<V, lds_dptr, lds_dptr, true>(xi, alpha, bi, bj, bk, ijk, jS, kS, A.a, A.b, A.h2inv);
      }
    }
    __syncthreads();
    return Ax;
  };
  // dot(a, b): misc.c:230-280 -- per tile of 8 x 8 rows a partial accumulated k, j, i; the partials added in tile order
  const int tiles_side = (dim + BLOCKCOPY_TILE_J - 1) / BLOCKCOPY_TILE_J, ntiles = tiles_side * ((dim + BLOCKCOPY_TILE_K - 1) / BLOCKCOPY_TILE_K);
  auto dot = [&](double va, double vb) -> double {
    if (active) scr[tid] = va * vb;
    __syncthreads();
    if (tid < ntiles) {
      const int k0 = (tid / tiles_side) * BLOCKCOPY_TILE_K, j0 = (tid % tiles_side) * BLOCKCOPY_TILE_J;
      const int k1 = min(k0 + BLOCKCOPY_TILE_K, dim), j1 = min(j0 + BLOCKCOPY_TILE_J, dim);
      double acc = 0.0;
      for (int k = k0; k < k1; k++) for (int j = j0; j < j1; j++) { const lds_dptr row = scr + dim * (j + dim * k); for (int i = 0; i < dim; i++) acc += row[i]; }
      scr[512 + tid] = acc;
    }
    __syncthreads();
    if (tid == 0) { double sum = 0.0; for (int t = 0; t < ntiles; t++) sum += scr[512 + t]; scr[1024] = sum; }
    __syncthreads();
    const double v = scr[1024];
    __syncthreads();
    return v;
  };
  auto norm = [&](double v) -> double {                            // max |v| (misc.c:303-349): exact under any order
    double m = 0.0;
    if (active) { const double f = fabs(v); m = (f > m) ? f : m; }
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(m, off, 64); m = (o > m) ? o : m; }
    if (lane == 0) scr[1032 + wave] = m;
    __syncthreads();
    m = scr[1032];
    for (int w = 1; w < nwaves; w++) m = (scr[1032 + w] > m) ? scr[1032 + w] : m;
    __syncthreads();
    return m;
  };
  const double want = A.want;
  int it = 0;
  // host/solvers.c bicgstab(), Dirichlet (no mean to remove)
  r0 = rhs - apply(x);
  r = 1.0 * r0;
  p = 1.0 * r0;
  {
    double rho = dot(r, r0);
    const double r0_norm = norm(r);
    if (!(rho == 0.0 || r0_norm == 0.0)) {
      while (it < 200) {
        it++;
        q = 1.0 * dinv * p;
        Ap = apply(q);
        const double Ap_r0 = dot(Ap, r0);
        if (Ap_r0 == 0.0) break;
        const double al = rho / Ap_r0;
        if (__builtin_isinf(al)) break;
        x = 1.0 * x + al * q;
        sv = 1.0 * r + (-al) * Ap;
        const double s_norm = norm(sv);
        if (s_norm == 0.0 || s_norm < want * r0_norm) break;
        tv = 1.0 * dinv * sv;
        As = apply(tv);
        const double As_As = dot(As, As);
        const double As_s = dot(As, sv);
        if (As_As == 0.0) break;
        const double omega = As_s / As_As;
        if (omega == 0.0 || __builtin_isinf(omega)) break;
        x = 1.0 * x + omega * tv;
        r = 1.0 * sv + (-omega) * As;
        const double r_norm = norm(r);
        if (r_norm == 0.0 || r_norm < want * r0_norm) break;
        const double rho_new = dot(r, r0);
        if (rho_new == 0.0) break;
        const double beta = (rho_new / rho) * (al / omega);
        if (__builtin_isinf(beta)) break;
        tmp = 1.0 * p + (-omega) * Ap;
        p = 1.0 * r + beta * tmp;
        rho = rho_new;
      }
    }
  }
  if (active) {
    vec_origin(L, 0, A.e_id)[ijk] = x;
    vec_origin(L, 0, r0_id)[ijk] = r0; vec_origin(L, 0, r_id)[ijk] = r; vec_origin(L, 0, p_id)[ijk] = p; vec_origin(L, 0, q_id)[ijk] = q;
    vec_origin(L, 0, s_id)[ijk] = sv; vec_origin(L, 0, t_id)[ijk] = tv; vec_origin(L, 0, Ap_id)[ijk] = Ap; vec_origin(L, 0, As_id)[ijk] = As;
    vec_origin(L, 0, VECTOR_TEMP)[ijk] = tmp;
  }
  if (tid == 0 && A.krylov_iterations) *A.krylov_iterations += it;
}
template <int V>
__global__ __launch_bounds__(512) void bottom_bicgstab_kernel(const hpgmg_hip_level L, const BottomArgs A) { bottom_bicgstab_body<V>(L, A); }

// ---------------------------------------------------------------------------------------------
// A queue of BLAS-1 / operator calls on a level of ONE small box (<= 512 cells) as one single-workgroup launch, ending -- if the caller
// wants a value -- in the dot product or norm that made the host ask.  This is what a host-driven Krylov solver does on the bottom level
// (the reference's solvers/bicgstab.c through operators.h, "Route B"): per iteration ~18 launches of an 8-cell kernel and 6 scalars fetched;
// the plugin postpones the void operators and issues them together with the value-returning one: 6 launches.  Every operation is the
// expression of its own kernel: misc.c add_vectors c = sa*a + sb*b, mul_vectors c = s*a*b, scale_vector c = s*a (blas1.hip
// elementwise_kernel); apply_op / residual = apply_BCs on the operand with the level's own boundary entries, then the stencil; dot = the
// products summed k, j, i (one dim x 8 x 8 tile: the order of misc.c:261-269 and tile_sum_kernel); norm = max |a|.
enum { SO_ADD = 1, SO_MUL, SO_SCALE, SO_APPLY, SO_RESIDUAL, SO_DOT, SO_NORM };
constexpr int kSmallOpsMax = 12;
struct SmallOp { int kind, c, a, b; double sa, sb; };
struct SmallOpsArgs {
  int n, bc_kind, zero_first, n_bc;
  const blockCopy_type *bc_list;
  double a, b, h2inv;
  ResultSlot *result; unsigned long long seq;
  SmallOp op[kSmallOpsMax];
};
// a launch may end in TWO value-returning operations (the second one a guess of what the host asks next): value k goes to the k-th double of
// the slot's payload (value, then the word behind the sequence number), the LAST operation of the list publishes the sequence number
__device__ __forceinline__ void small_ops_value(ResultSlot *slot, int k, double v, bool last, unsigned long long seq) {
  double *second = reinterpret_cast<double *>(slot) + 2;
  if (k == 0) slot->value = v; else *second = v;
  if (last) { __threadfence_system(); __hip_atomic_store(&slot->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
template <int V>
__global__ __launch_bounds__(512) void small_ops_kernel(const hpgmg_hip_level L, const SmallOpsArgs A) {
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  __shared__ double part[512 + 8];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride, total = dim * dim * dim;
  const bool active = tid < total;
  const int ci = tid % dim, cj = (tid / dim) % dim, ck = tid / (dim * dim), ijk = ci + cj * jS + ck * kS;
  int nvalues = 0;                                                 // value-returning operations so far: the first goes to result->value, the second behind it
  for (int q = 0; q < A.n; q++) {
    const int kind = A.op[q].kind, idc = A.op[q].c, ida = A.op[q].a, idb = A.op[q].b;
    const double sa = A.op[q].sa, sb = A.op[q].sb;
    if (kind == SO_ADD)        { if (active) vec_origin(L, 0, idc)[ijk] = sa * vec_origin(L, 0, ida)[ijk] + sb * vec_origin(L, 0, idb)[ijk]; }
    else if (kind == SO_MUL)   { if (active) vec_origin(L, 0, idc)[ijk] = sa * vec_origin(L, 0, ida)[ijk] * vec_origin(L, 0, idb)[ijk]; }
    else if (kind == SO_SCALE) { if (active) vec_origin(L, 0, idc)[ijk] = sa * vec_origin(L, 0, ida)[ijk]; }
    else if (kind == SO_APPLY || kind == SO_RESIDUAL) {
      // exchange_boundary: one box, nothing to copy; apply_BCs on the operand, then the stencil (operators.*.c apply_op / residual)
      if (A.bc_kind && A.zero_first) { for (int e = wave; e < A.n_bc; e += nwaves) bc_zero_entry(L, ida, A.bc_list[e], lane, 64); __syncthreads(); }
      for (int e = wave; e < A.n_bc; e += nwaves) {
        if (A.bc_kind == 1) bc_p1_entry(L, ida, A.bc_list[e], lane, 64);
        else if (A.bc_kind == 2) bc_p2_entry(L, ida, A.bc_list[e], lane, 64);
        else if (A.bc_kind == 3) bc_v2_entry(L, ida, A.bc_list[e], lane, 64);
        else if (A.bc_kind == 4) bc_v4_entry(L, ida, A.bc_list[e], lane, 64);
      }
      __syncthreads();
      if (active) {
        const double *x = vec_origin(L, 0, ida);
        double Ax;
        if (k27) {
          const plane9 m = load_plane(x + (ijk - kS), jS), c = load_plane(x + ijk, jS), pp = load_plane(x + (ijk + kS), jS);
          Ax = apply_op_27pt(m, c, pp, A.a, A.b, A.h2inv);
        } else {
          const double *none = nullptr;
          Ax = apply_op_direct<V>(x, kHelm ? (const double *)vec_origin(L, 0, VECTOR_ALPHA) : none, kVC ? (const double *)vec_origin(L, 0, VECTOR_BETA_I) : none,
                                  kVC ? (const double *)vec_origin(L, 0, VECTOR_BETA_J) : none, kVC ? (const double *)vec_origin(L, 0, VECTOR_BETA_K) : none, ijk, jS, kS, A.a, A.b, A.h2inv);
        }
        vec_origin(L, 0, idc)[ijk] = (kind == SO_RESIDUAL) ? vec_origin(L, 0, idb)[ijk] - Ax : Ax;
      }
    } else if (kind == SO_DOT) {
      if (active) part[tid] = vec_origin(L, 0, ida)[ijk] * vec_origin(L, 0, idb)[ijk];
      __syncthreads();
      if (tid == 0) { double acc = 0.0; for (int t = 0; t < total; t++) acc += part[t]; small_ops_value(A.result, nvalues, 0.0 + acc, q == A.n - 1, A.seq); }   // (one tile: its partial added to 0.0, as tile_sum_kernel does)
      nvalues++;
    } else if (kind == SO_NORM) {
      double m = 0.0;
      if (active) { const double f = fabs(vec_origin(L, 0, ida)[ijk]); m = (f > m) ? f : m; }
      for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(m, off, 64); m = (o > m) ? o : m; }
      if (lane == 0) part[512 + wave] = m;
      __syncthreads();
      if (tid == 0) { for (int w = 1; w < nwaves; w++) m = (part[512 + w] > m) ? part[512 + w] : m; small_ops_value(A.result, nvalues, m, q == A.n - 1, A.seq); }
      nvalues++;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The rest of a V-cycle below a level of ONE box, 27-point / fv2 / fv4 plugins (mg.c:1133-1166), as one single-workgroup launch.  Driven
// operator by operator a visit of such a level is 4 launches on the way down (smooth -- itself one launch, above --, the boundary fill and
// the stencil of residual(), restriction + zero_vector) and 3 on the way up, ~5 us each for work of a microsecond.  Here the levels of
// the chain take turns in LDS: an image of the level's box (nine vectors, padded layout) is loaded, small_level_run() smooths it and
// forms the residual, the restriction goes straight from the image to the coarse level's right-hand side in memory, the written vectors go
// back; the bottom solve is bottom_bicgstab_body(); on the way up the coarse correction is staged behind the image, its boundary
// conditions are applied there and the tensor rule adds it to the image's x before the smoother runs.  Every per-cell expression is the one
// of the per-operator kernels (restrict_entry / interp_tensor_kernel in blocks.hip), so the result is bit-identical to the launches it
// replaces (the coarse correction's ghost zone in MEMORY is left as it was: every reader fills it first).
template <int V>
__global__ __launch_bounds__(512) void small_vtail_kernel(const hpgmg_hip_small_tail_args *__restrict__ Tp, unsigned long long *tl) {
#ifdef HPGMG_EXP_TIMELINE
  int tl_n = 0;
#define VT_MARK() do { if (tl && threadIdx.x == 0 && tl_n < 250) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VT_MARK() do { (void)tl; } while (0)
#endif
  VT_MARK();
#ifdef HPGMG_EXP_TIMELINE
  if (tl && threadIdx.x == 0) tl[253] = __builtin_amdgcn_s_memtime();      // shader clock against the 100 MHz real-time marks
#endif
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  constexpr int ORDER = k27 ? 2 : 3;                               // interpolation_p2.c (27-point) / interpolation_v2.c (fv2, fv4)
  extern __shared__ double vt_lds[];
  __shared__ int s_bc[kBcWords];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6, nth = (int)blockDim.x;
  const int n = Tp->n, e_id = Tp->e_id, R_id = Tp->R_id;
  const lds_dptr img = (lds_dptr)vt_lds;
  // slots of an image: x, VECTOR_TEMP, rhs, Dinv, alpha, beta_i, beta_j, beta_k (small_level_run addresses them by slot number)
  const int slot_vec[8] = { e_id, VECTOR_TEMP, R_id, VECTOR_DINV, kHelm ? VECTOR_ALPHA : -1, kVC ? VECTOR_BETA_I : -1, kVC ? VECTOR_BETA_J : -1, kVC ? VECTOR_BETA_K : -1 };
  int unused_n = 0;
  const int legs = Tp->legs;                                       // bit 0: the way down, bit 1: the bottom solve, bit 2: the way up
  for (int ph = 0; ph < 2 * n - 1; ph++) {
    if (ph < n - 1 ? !(legs & 1) : (ph == n - 1 ? !(legs & 2) : !(legs & 4))) continue;
    if (ph == n - 1) {                                             // ---- the bottom solve (solvers.c IterativeSolver -> BiCGStab)
      const hpgmg_hip_small_tail_level &lb = Tp->lv[n - 1];
      const hpgmg_hip_level Lb = lb.L;
      BottomArgs B;
      B.e_id = e_id; B.R_id = R_id; B.krylov_base = Tp->krylov_base; B.bc_kind = lb.n_bc > 0 ? lb.bc_kind : 0; B.zero_first = lb.zero_first; B.n_bc = lb.n_bc;
      B.a = Tp->a; B.b = Tp->b; B.h2inv = lb.h2inv; B.want = Tp->want; B.bc_list = lb.bc_list; B.krylov_iterations = Tp->krylov_iterations;
      bottom_bicgstab_body<V>(Lb, B);
      __threadfence(); __syncthreads();
      VT_MARK();
      continue;
    }
    const bool down = ph < n - 1;
    const int l = down ? ph : (2 * (n - 1) - ph);
    const hpgmg_hip_small_tail_level &lv = Tp->lv[l];
    const hpgmg_hip_level L = lv.L;
    const int vol = L.volume, jS = L.jStride, kS = L.kStride, dim = L.dim, first = L.ghosts * (1 + jS + kS);
    // ---- the image of level l
    {
      const double *base = L.box_base[0];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (slot_vec[q] < 0) continue;
        const gbl_cdptr g = (gbl_cdptr)(base + (size_t)slot_vec[q] * vol);
#pragma unroll 4
        for (int t = tid; t < vol; t += nth) img[q * vol + t] = g[t];
      }
    }
    VT_MARK();
    if (!down) {
      // ---- interpolation_vcycle(level l, e, 1.0, level l + 1, e): the coarse correction behind the image, its boundary conditions, the rule
      const hpgmg_hip_small_tail_level &lc = Tp->lv[l + 1];
      const hpgmg_hip_level Lc = lc.L;
      const int cvol = Lc.volume, cj = Lc.jStride, ck = Lc.kStride, cfirst = Lc.ghosts * (1 + cj + ck);
      const lds_dptr stage = img + 8 * vol;
      const gbl_cdptr gx = (gbl_cdptr)(Lc.box_base[0] + (size_t)e_id * cvol);
      for (int t = tid; t < cvol; t += nth) stage[t] = gx[t];
      lds_bc_words_fill(s_bc, lc.ibc_list, lc.n_ibc, tid, nth, false);
      __syncthreads();
      const lds_dptr cx = stage + cfirst;
      if (lc.ibc_kind && lc.ibc_zero_first) { for (int e = wave; e < lc.n_ibc; e += nwaves) bc_zero_entry_at(cx, Lc, lds_bc_entry(s_bc, e), lane, 64); __syncthreads(); }
      for (int e = wave; e < lc.n_ibc; e += nwaves) {
        const blockCopy_type en = lds_bc_entry(s_bc, e);
        if (lc.ibc_kind == 1) bc_p1_entry_at(cx, Lc, en, lane, 64);
        else if (lc.ibc_kind == 2) bc_p2_entry_at(cx, Lc, en, lane, 64);
        else if (lc.ibc_kind == 3) bc_v2_entry_at(cx, Lc, en, lane, 64);
      }
      __syncthreads();
      const lds_dptr xf = img + first;
      for (int t = tid; t < dim * dim * dim; t += nth) {
        const int i = t % dim, j = (t / dim) % dim, k = t / (dim * dim);
        const lds_dptr c = cx + ((i >> 1) + (j >> 1) * cj + (k >> 1) * ck);
        double tk[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
          double tj[3];
#pragma unroll
          for (int jj = 0; jj < 3; jj++) {
            double line[3];
#pragma unroll
            for (int ii = 0; ii < 3; ii++) line[ii] = c[(ii - 1) + (jj - 1) * cj + (kk - 1) * ck];
            tj[jj] = interp_rule<ORDER>((i & 1) != 0, line);
          }
          tk[kk] = interp_rule<ORDER>((j & 1) != 0, tj);
        }
        const double add = interp_rule<ORDER>((k & 1) != 0, tk);
        const int ijk = i + j * jS + k * kS;
        xf[ijk] = 1.0 * xf[ijk] + add;
      }
    }
    lds_bc_words_fill(s_bc, lv.bc_list, lv.n_bc, tid, nth, lv.bc_kind == 4);
    __syncthreads();
    VT_MARK();
    // ---- smooth(level l), and on the way down residual(level l, VECTOR_TEMP, e, R)
    for (int pass = 0; pass < (down ? 2 : 1); pass++) {
      SmallArgs A;
      A.mode = pass ? MODE_RESIDUAL : Tp->mode; A.sweeps = pass ? 1 : Tp->sweeps; A.out_of_place = pass ? 0 : Tp->out_of_place;
      A.x_id = 0; A.rhs_id = 2; A.res_id = 1; A.bc_kind = lv.n_bc > 0 ? lv.bc_kind : 0; A.zero_first = lv.zero_first;
      A.a = Tp->a; A.b = Tp->b; A.h2inv = lv.h2inv; A.copy_list = nullptr; A.n_copy = 0; A.bc_list = lv.bc_list; A.n_bc = lv.n_bc; A.lds_resident = 1; A.timeline = nullptr;
      const int ids[kSmallSlots] = { 0, 1, 2, 3, 4, 5, 6, 7, pass ? 1 : 0 };
      small_level_run<V, true>(L, A, vt_lds, lv.bc_list, lv.n_bc <= 32 ? (const int *)s_bc : nullptr, ids, nullptr, unused_n,
                               [&](int s, double &c1, double &c2) { c1 = lv.c1[s]; c2 = lv.c2[s]; });
      VT_MARK();
    }
    if (down) {
      // ---- restriction(level l + 1, R, level l, VECTOR_TEMP, RESTRICT_CELL) and zero_vector(level l + 1, e)
      const hpgmg_hip_level Lc = Tp->lv[l + 1].L;
      const int cdim = Lc.dim, cj = Lc.jStride, ck = Lc.kStride, cvol = Lc.volume;
      const lds_dptr tf = img + (vol + first);
      const gbl_dptr rc = (gbl_dptr)(Lc.box_base[0] + (size_t)R_id * cvol + (size_t)Lc.ghosts * (1 + cj + ck));
      for (int t = tid; t < cdim * cdim * cdim; t += nth) {
        const int i = t % cdim, j = (t / cdim) % cdim, k = t / (cdim * cdim);
        const lds_dptr f = tf + (2 * i + 2 * j * jS + 2 * k * kS);
        double v = f[0] + f[1]; v = v + f[jS]; v = v + f[1 + jS]; v = v + f[kS]; v = v + f[1 + kS]; v = v + f[jS + kS]; v = v + f[1 + jS + kS];
        rc[i + j * cj + k * ck] = v * 0.125;
      }
      const gbl_dptr zc = (gbl_dptr)(Lc.box_base[0] + (size_t)e_id * cvol);
      for (int t = tid; t < cvol; t += nth) zc[t] = 0.0;
    }
    // ---- what was written goes back: x and VECTOR_TEMP, ghost zones included (the boundary entries filled them)
    {
      double *base = L.box_base[0];
      const gbl_dptr gx = (gbl_dptr)(base + (size_t)e_id * vol), gt = (gbl_dptr)(base + (size_t)VECTOR_TEMP * vol);
#pragma unroll 4
      for (int t = tid; t < vol; t += nth) { gx[t] = img[t]; gt[t] = img[vol + t]; }
    }
    __threadfence(); __syncthreads();
    VT_MARK();
  }
#ifdef HPGMG_EXP_TIMELINE
  if (tl && threadIdx.x == 0) { tl[254] = __builtin_amdgcn_s_memtime(); tl[255] = (unsigned long long)tl_n; }
#endif
#undef VT_MARK
}

// ---- smoother-kernel profiling: hipEvent pair around every smoother launch ----
static bool g_profile = false;
static long long g_profile_min_cells = 0;   // only launches covering at least this many cells are timed
static const int kMaxPairs = 8192;
static hipEvent_t g_ev[2 * kMaxPairs];
static int g_pairs_alloc = 0, g_pairs_used = 0;
static long long g_prof_cells = 0, g_prof_launches = 0;
static double g_prof_ms_flushed = 0.0;

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
int profile_begin(long long cells) {
  if (!g_profile || cells < g_profile_min_cells || hpgmg_hip_graph_is_open()) return -1;
  if (g_pairs_used == kMaxPairs) profile_flush();
  if (g_pairs_used == g_pairs_alloc) { hipEventCreate(&g_ev[2 * g_pairs_alloc]); hipEventCreate(&g_ev[2 * g_pairs_alloc + 1]); g_pairs_alloc++; }
  int p = g_pairs_used++;
  hipEventRecord(g_ev[2 * p], g_stream);
  return p;
}
void profile_end(int p, long long cells) {
  if (p < 0) return;
  hipEventRecord(g_ev[2 * p + 1], g_stream);
  g_prof_cells += cells; g_prof_launches++;
}

static int g_ghost_free = 0;
static int g_tile_ghost_free = 0;
static TileFused g_tile_fused = {};      // consumed by the next tiled residual launch (27-point / fv4): what becomes of the residual
static int g_tile_last_blocks = 0;       // workgroups of that launch (= partial maxima written)   // the LDS-tiled fv4 / 27-point kernels read x outside a box from the neighbouring box (all boxes local)
static int g_defer_mode = 0;   // 0: whole boxes; 1: skip the cells next to remote faces; 2: only those cells (stencil7_shell_kernel)

static int env_int(const char *name, int dflt) { const char *e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
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
    const int tgrid = grid_for(A.total_blocks, &A.per_xcd);
    const long long tcells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
    const int tprof = is_smoother ? profile_begin(tcells) : -1;
    if (narrow) hipLaunchKernelGGL((stencil27_tile_kernel<TM, 16, 32>), dim3(tgrid), dim3(32, 16), 0, g_stream, *L, A);
    else        hipLaunchKernelGGL((stencil27_tile_kernel<TM, 8, 64>), dim3(tgrid), dim3(64, 8), 0, g_stream, *L, A);
    profile_end(tprof, tcells);
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
  static const int kc_min = env_int("HPGMG_TUNE_FV4_KCHUNK_MIN", 2);
  while (kchunk > kc_min && (long long)L->num_boxes * P.tiles_i * P.tiles_j * (L->dim / kchunk) < want) kchunk /= 2;
  static const int tune_kc = env_int("HPGMG_TUNE_FV4_KCHUNK", 0);
  if (tune_kc > 0 && L->dim % tune_kc == 0) kchunk = tune_kc;
  P.kchunk = kchunk; P.chunks_k = (L->dim + kchunk - 1) / kchunk;
  P.total_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i;
  g_tile_last_blocks = P.total_blocks;
  const int grid = grid_for(P.total_blocks, &P.per_xcd);
  const size_t lds = (size_t)11 * (TI + 4) * (TJ + 4) * sizeof(double);
  const long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  const int prof = is_smoother ? profile_begin(cells) : -1;
#define FV4_TILE_CASE(VAR) { \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)fv4_tile_kernel<VAR, MODE, TJ, TI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; } \
    hipLaunchKernelGGL((fv4_tile_kernel<VAR, MODE, TJ, TI>), dim3(grid), dim3(TI, TJ), lds, g_stream, *L, P); }
  if (variant == HPGMG_HIP_FV4_VC_HELMHOLTZ) FV4_TILE_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ)
  else FV4_TILE_CASE(HPGMG_HIP_FV4_VC_POISSON)
#undef FV4_TILE_CASE
  profile_end(prof, cells);
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

// ---- halo of a sweep pair across rank boundaries: regions of x0 / xm1 / rhs (or a level vector) <-> one message buffer ----
struct HaloRefs { VecRef x0, xm1; int rhs_id; double *const *scr_base; };
__device__ __forceinline__ double *halo_vec(const hpgmg_hip_level &L, const HaloRefs &R, int vec, int box) {
  const size_t first = (size_t)L.ghosts * (size_t)(1 + L.jStride + L.kStride);
  if (vec >= 16) return L.box_base[box] + (size_t)(vec - 16) * (size_t)L.volume + first;
  if (vec == 2) return L.box_base[box] + (size_t)R.rhs_id * (size_t)L.volume + first;
  const VecRef r = (vec == 0) ? R.x0 : R.xm1;
  return (r.scratch ? R.scr_base[box] : L.box_base[box]) + (size_t)r.id * (size_t)L.volume + first;
}
template <bool kUnpack>
__global__ __launch_bounds__(256) void pair_halo_kernel(const hpgmg_hip_level L, const HaloRefs R, const hpgmg_hip_halo_entry *__restrict__ list,
                                                        double *buf, double *deep, double *deep_beta) {
  const hpgmg_hip_halo_entry e = list[blockIdx.x];
  const int n = e.ni * e.nj * e.nk, jS = L.jStride, kS = L.kStride;
  double *v = halo_vec(L, R, e.vec, e.box) + e.i + e.j * jS + e.k * kS;
  double *b = buf + e.off;
  double *plane = nullptr;
  if (kUnpack && e.deep >= 8) plane = deep_beta + ((size_t)e.box * 3 + (e.deep - 8)) * (size_t)L.dim * L.dim;
  else if (kUnpack && e.deep >= 0) plane = deep + ((size_t)e.box * 6 + e.deep) * (size_t)L.dim * L.dim;
  for (int t = blockIdx.y * 256 + threadIdx.x; t < n; t += gridDim.y * 256) {
    const int ti = t % e.ni, tj = (t / e.ni) % e.nj, tk = t / (e.ni * e.nj);
    if (!kUnpack) b[t] = v[ti + tj * jS + tk * kS];
    else if (plane) plane[t] = b[t];
    else v[ti + tj * jS + tk * kS] = b[t];
  }
}
// ---- fused forms of residual() on the bandwidth-bound fine level (stencil7_wide_kernel, ghost-free, every face local) ----
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
  const int grid = grid_for(P.total_blocks, &P.per_xcd) + extra_blocks;      // the extra (zeroing) workgroups follow the stencil grid
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
int hpgmg_hip_profile_smoother_read(double *total_ms, long long *launches, long long *cells) {
  profile_flush();
  if (total_ms) *total_ms = g_prof_ms_flushed;
  if (launches) *launches = g_prof_launches;
  if (cells) *cells = g_prof_cells;
  return 0;
}

int hpgmg_hip_smooth_cheby(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                           double a, double b, double h2inv, double c1, double c2) {
  StencilArgs P = {}; P.xn_id = xn_id; P.xout_id = xnp1_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv; P.c1 = c1; P.c2 = c2;
  return launch<MODE_CHEBY>(L, variant, P, true);
}
// Two Chebyshev sweeps in one pass (cheby_pair.hpp).  Vector references are (scratch?, id) pairs: scratch ids 0/1
// address the two plugin-private vectors behind scr_base.  Returns hipErrorNotSupported-like status 1 (no launch,
// no error recorded) when the level does not fit the kernel's assumptions, so the caller can fall back.
static int pair_supported_dims(const hpgmg_hip_level *L, int variant, int Di, int Dj, int Dk) {
  if (variant != HPGMG_HIP_7PT_VC_HELMHOLTZ && variant != HPGMG_HIP_7PT_VC_POISSON && variant != HPGMG_HIP_7PT_CC) return 0;
  if (L->num_boxes <= 0 || L->num_boxes > kPairMaxBoxes || L->periodic || !(L->flags & 1) || L->ghosts < 1 || Di % 128 != 0) return 0;
  // a wave owns a 128-cell row: whole multiples of 128 per box, or several boxes (consecutive in one slab) per row
  if (L->dim % 128 != 0 && !(128 % L->dim == 0 && L->dim >= 16 && (L->box_stride > 0 || L->num_boxes == 1))) return 0;
  if (L->jStride % 2 || L->kStride % 2 || L->volume % 2) return 0;
  if (Di % L->dim || Dj % L->dim || Dk % L->dim) return 0;
  if ((long long)(Di / L->dim) * (Dj / L->dim) * (Dk / L->dim) != L->num_boxes) return 0;
  return 1;
}
int hpgmg_hip_smooth_cheby_pair_supported(const hpgmg_hip_level *L, int variant) { return pair_supported_dims(L, variant, L->dim_i, L->dim_j, L->dim_k); }
// several ranks: this rank's boxes form a brick of nbi x nbj x nbk boxes (numbered lexicographically inside it)
int hpgmg_hip_smooth_cheby_pair_supported_brick(const hpgmg_hip_level *L, int variant, int nbi, int nbj, int nbk) {
  if (L->dim % 128 != 0) return 0;
  return pair_supported_dims(L, variant, nbi * L->dim, nbj * L->dim, nbk * L->dim);
}
int hpgmg_hip_coef32_refresh(const hpgmg_hip_level *L, float *const *c32_base, int num_vectors) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  hipLaunchKernelGGL(coef32_convert_kernel, dim3(512, L->num_boxes), dim3(256), 0, g_stream, *L, c32_base, num_vectors);
  HPGMG_LAUNCH_CHECK("coef32_convert_kernel");
  return 0;
}
// interpolation_vcycle folded into the NEXT sweep-pair launch (consumed by it): x0 := prescale * x0 + parent(coarse_id of Lc)
// remote faces of the NEXT sweep-pair launch (consumed by it): see hpgmg_hip_pair_set_halo
static bool g_pair_halo_set = false;
static int g_pair_rem[6], g_pair_brick[3];
static const double *g_pair_deep = nullptr, *g_pair_deep_beta = nullptr;
static long long g_pair_launches = 0, g_pair_remote_launches = 0;
#ifdef HPGMG_EXP_TIMELINE
static double *g_exp_timeline = nullptr;     // experiment build: where the pair kernel's first workgroup records its step timeline
#endif
static int g_pair_discard_x1 = 0;     // consumed by the next Chebyshev pair launch: its out1 vector is scratch, do not store x1
static const hpgmg_hip_level *g_pair_interp_level = nullptr;
static int g_pair_interp_id = 0;
static double g_pair_interp_prescale = 1.0;
// The pre-pass's coefficient values, packed once per operator rebuild (cheby_pair.hpp: PairArgs.edge_coef).  Keyed by the level's box table;
// hpgmg_hip_pair_packed_invalidate() when the coefficients change, _forget() when the level goes away.  HPGMG_TUNE_PAIR_PACKED=0: read in place.
struct EdgePack { const void *key; int variant, Di, Dj, Dk; double *buf; bool valid; };
static std::vector<EdgePack> g_edge_packs;
extern "C" void hpgmg_hip_pair_packed_invalidate(const hpgmg_hip_level *L) { for (EdgePack &e : g_edge_packs) if (!L || e.key == (const void *)L->box_base) e.valid = false; }
extern "C" void hpgmg_hip_pair_packed_forget(const hpgmg_hip_level *L) {
  for (size_t q = 0; q < g_edge_packs.size();) {
    if (!L || g_edge_packs[q].key == (const void *)L->box_base) { (void)hipStreamSynchronize(g_stream); (void)hipFree(g_edge_packs[q].buf); g_edge_packs.erase(g_edge_packs.begin() + (long)q); }
    else q++;
  }
}
static EdgePack *edge_pack_slot(const hpgmg_hip_level *L, int variant, const PairArgs &A) {
  static const int on = env_int("HPGMG_TUNE_PAIR_PACKED", 1);
  if (!on || A.tiles_i < 2 || variant == HPGMG_HIP_7PT_CC) return nullptr;
  for (EdgePack &e : g_edge_packs) if (e.key == (const void *)L->box_base && e.variant == variant && e.Di == A.Di && e.Dj == A.Dj && e.Dk == A.Dk) return &e;
  EdgePack e = { (const void *)L->box_base, variant, A.Di, A.Dj, A.Dk, nullptr, false };
  const size_t n = (size_t)2 * (A.tiles_i - 1) * A.Dk * A.Dj * 8;
  if (hipMalloc((void **)&e.buf, n * sizeof(double)) != hipSuccess) return nullptr;
  g_edge_packs.push_back(e);
  return &g_edge_packs.back();
}
// Two-part launches of the sweep pair across rank boundaries (hpgmg_hip_set_tile_part): part 1 = the workgroups whose slab / chunk / tile touches
// no face another rank owns -- they read nothing the halo exchange delivers (ghost zones, deep planes, ghost columns of the pre-pass) --, part 2 the
// others.  Each part runs the (cheap) pre-pass in full: before the exchange its cells next to remote faces are formed from stale ghost values and
// read by nobody, the second run overwrites them.
struct PairOrder { int ti, sj, ck, rem[6], part; int *d_order; int grid, per_xcd, count; };
static std::vector<PairOrder> g_pair_orders;
static const PairOrder *pair_part_order(const PairArgs &A, int part) {
  for (const PairOrder &o : g_pair_orders)
    if (o.ti == A.tiles_i && o.sj == A.slabs_j && o.ck == A.chunks_k && o.part == part && memcmp(o.rem, A.rem, sizeof o.rem) == 0) return &o;
  PairOrder o = {}; o.ti = A.tiles_i; o.sj = A.slabs_j; o.ck = A.chunks_k; o.part = part; memcpy(o.rem, A.rem, sizeof o.rem);
  std::vector<int> sel;
  for (int l = 0; l < A.total_blocks; l++) {
    int t = l;
    const int ti = t % A.tiles_i; t /= A.tiles_i;
    const int sj = t % A.slabs_j; t /= A.slabs_j;
    const int ck = t;
    const bool later = (ti == 0 && A.rem[0]) || (ti == A.tiles_i - 1 && A.rem[1]) || (sj == 0 && A.rem[2]) || (sj == A.slabs_j - 1 && A.rem[3]) || (ck == 0 && A.rem[4]) || (ck == A.chunks_k - 1 && A.rem[5]);
    if (later == (part == 2)) sel.push_back(l);
  }
  o.count = (int)sel.size();
  if (o.count > 0) {
    o.per_xcd = (o.count + kXcds - 1) / kXcds; o.grid = o.per_xcd * kXcds;
    sel.resize((size_t)o.grid, A.total_blocks);
    if (hipMalloc((void **)&o.d_order, sel.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(o.d_order, sel.data(), sel.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(o.d_order); return nullptr; }
  }
  g_pair_orders.push_back(o);
  return &g_pair_orders.back();
}
static int smooth_pair(const hpgmg_hip_level *L, int variant, int gsrb, int sweep_a, double *const *scr_base, const float *const *c32_base,
                       int x0_scr, int x0_id, int xm1_scr, int xm1_id, int out1_scr, int out1_id, int out2_scr, int out2_id,
                       int rhs_id, double a, double b, double h2inv, double c1a, double c2a, double c1b, double c2b) {
  // the requests set for THIS launch (hpgmg_hip_pair_set_halo / _discard_x1 / _set_interp) are taken here, before any exit: whatever
  // happens below, none of them can stay pending and change a later launch
  const bool remote = g_pair_halo_set;
  const int discard_x1 = g_pair_discard_x1;
  const hpgmg_hip_level *const interp_level = g_pair_interp_level;
  g_pair_halo_set = false; g_pair_discard_x1 = 0; g_pair_interp_level = nullptr;
  HPGMG_SKIP_IF_REPLAY();
  const int Di = remote ? g_pair_brick[0] * L->dim : L->dim_i, Dj = remote ? g_pair_brick[1] * L->dim : L->dim_j, Dk = remote ? g_pair_brick[2] * L->dim : L->dim_k;
  if (!pair_supported_dims(L, variant, Di, Dj, Dk)) return record_error(hipErrorInvalidValue, "smooth_cheby_pair: level not supported");
  if (remote && (L->dim % 128 != 0 || c32_base || interp_level)) return record_error(hipErrorInvalidValue, "smooth_cheby_pair: remote faces need whole-row boxes, fp64 coefficients, no folded interpolation");
  static const int tune_kc = env_int("HPGMG_TUNE_PAIR_KC", 0);
  constexpr int nw = 16;
  // k chunk: every workgroup costs KC+2 plane steps and (at 128 VGPRs, 16 waves) one workgroup occupies a CU, so the
  // launch takes ceil(workgroups / 256) rounds of KC+2 steps: pick the KC that minimises that product
  int kc = tune_kc;
  if (kc <= 0) {
    const int per_plane = (Di / 128) * ((Dj + (nw - 2) - 1) / (nw - 2)), slots = 256;
    long long best = -1;
    for (int c = 8; c <= 64 && c <= Dk; c++) {
      const long long wgs = (long long)per_plane * ((Dk + c - 1) / c), cost = ((wgs + slots - 1) / slots) * (c + 2);
      if (best < 0 || cost < best) { best = cost; kc = c; }
    }
  }
  PairArgs A = {};
  A.x0 = VecRef{x0_scr, x0_id}; A.xm1 = VecRef{xm1_scr, xm1_id}; A.out1 = VecRef{out1_scr, out1_id}; A.out2 = VecRef{out2_scr, out2_id};
  A.rhs_id = rhs_id; A.a = a; A.b = b; A.h2inv = h2inv; A.c1a = c1a; A.c2a = c2a; A.c1b = c1b; A.c2b = c2b;
  A.scr_base = scr_base; A.c32_base = c32_base; A.sweep_a = sweep_a;
  A.keep_x1 = discard_x1 ? 0 : 1;
  const bool interp = (interp_level != nullptr);
  if (interp) {
    const hpgmg_hip_level *C = interp_level;
    if (L->dim % 128 != 0 || C->num_boxes != L->num_boxes || 2 * C->dim != L->dim) return record_error(hipErrorInvalidValue, "smooth pair with interpolation: level pair not supported");
    A.Lc = *C; A.coarse_id = g_pair_interp_id; A.prescale = g_pair_interp_prescale;
  }
  A.nbi = Di / L->dim; A.nbj = Dj / L->dim;
  A.Di = Di; A.Dj = Dj; A.Dk = Dk;
  if (remote) { for (int d = 0; d < 6; d++) A.rem[d] = g_pair_rem[d]; A.deep = g_pair_deep; A.deep_beta = g_pair_deep_beta; }
#ifdef HPGMG_EXP_TIMELINE
  else A.deep = g_exp_timeline;
#endif
  A.tiles_i = A.Di / 128; A.slabs_j = (A.Dj + (nw - 2) - 1) / (nw - 2); A.KC = kc; A.chunks_k = (A.Dk + kc - 1) / kc;
  A.total_blocks = A.tiles_i * A.slabs_j * A.chunks_k;
  int grid = grid_for(A.total_blocks, &A.per_xcd);
  long long cells = (long long)A.Di * A.Dj * A.Dk;
  const int part = remote ? g_tile_part : 0;
  if (part) {
    const PairOrder *o = pair_part_order(A, part);
    if (!o) return record_error(hipErrorOutOfMemory, "smooth pair: dispatch list of a partial launch");
    if (o->count == 0) return 0;
    A.order = o->d_order; grid = o->grid; A.per_xcd = o->per_xcd;
    cells = cells * o->count / A.total_blocks;
  }
  const size_t lds = (size_t)nw * 6 * 64 * sizeof(p2);
  if (!remote && !c32_base) {        // the pre-pass reads its coefficient values packed; (re)pack them after an operator rebuild
    EdgePack *pk = edge_pack_slot(L, variant, A);
    if (pk) {
      A.edge_coef = pk->buf;
      if (!pk->valid) {
        A.edge_blocks = ((A.Dj + 63) / 64) * A.Dk * 2 * (A.tiles_i - 1);
        const dim3 pgrid(grid_for(A.edge_blocks, &A.edge_per_xcd));
        if (variant == HPGMG_HIP_7PT_VC_HELMHOLTZ) hipLaunchKernelGGL((cheby_pair_edge_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, false, PAIR_CHEBY, false, false, true>), pgrid, dim3(64), 0, g_stream, *L, A);
        else hipLaunchKernelGGL((cheby_pair_edge_kernel<HPGMG_HIP_7PT_VC_POISSON, false, PAIR_CHEBY, false, false, true>), pgrid, dim3(64), 0, g_stream, *L, A);
        HPGMG_LAUNCH_CHECK("cheby_pair_edge_kernel (packing)");
        pk->valid = true;
      }
    }
  }
  const int prof = profile_begin(cells);
#define PAIR_LAUNCH2(VAR, C32, SM, NARROW, INTERP) { \
      static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)cheby_pair_kernel<VAR, nw, C32, SM, NARROW, INTERP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; } \
      hipLaunchKernelGGL((cheby_pair_kernel<VAR, nw, C32, SM, NARROW, INTERP>), dim3(grid), dim3(64, nw), lds, g_stream, *L, A); }
#define PAIR_LAUNCH_REMOTE(VAR, SM) { \
      static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)cheby_pair_kernel<VAR, nw, false, SM, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; } \
      const int ecols = 2 * (A.tiles_i - 1) + (A.rem[0] ? 1 : 0) + (A.rem[1] ? 1 : 0); \
      A.edge_blocks = ((A.Dj + 63) / 64) * A.Dk * ecols; const int egrid_r = grid_for(A.edge_blocks, &A.edge_per_xcd); \
      if (ecols > 0) hipLaunchKernelGGL((cheby_pair_edge_kernel<VAR, false, SM, false, true>), dim3(egrid_r), dim3(64), 0, g_stream, *L, A); \
      hipLaunchKernelGGL((cheby_pair_kernel<VAR, nw, false, SM, false, false, true>), dim3(grid), dim3(64, nw), lds, g_stream, *L, A); }
#define PAIR_LAUNCH(VAR, C32, SM) { \
      A.edge_blocks = ((A.Dj + 63) / 64) * A.Dk * 2 * (A.tiles_i - 1); const dim3 egrid(A.edge_blocks > 0 ? grid_for(A.edge_blocks, &A.edge_per_xcd) : 1); \
      if (remote) PAIR_LAUNCH_REMOTE(VAR, SM) \
      else if (interp) { \
        if (A.tiles_i > 1) hipLaunchKernelGGL((cheby_pair_edge_kernel<VAR, C32, SM, true>), egrid, dim3(64), 0, g_stream, *L, A); \
        PAIR_LAUNCH2(VAR, C32, SM, false, true) \
      } else { \
        if (A.tiles_i > 1) hipLaunchKernelGGL((cheby_pair_edge_kernel<VAR, C32, SM, false>), egrid, dim3(64), 0, g_stream, *L, A); \
        if (L->dim % 128 == 0) PAIR_LAUNCH2(VAR, C32, SM, false, false) else PAIR_LAUNCH2(VAR, C32, SM, true, false) \
      } }
#define PAIR_CASE(VAR) case VAR: \
    if (gsrb) { if (c32_base) PAIR_LAUNCH(VAR, true, PAIR_GSRB) else PAIR_LAUNCH(VAR, false, PAIR_GSRB) } \
    else      { if (c32_base) PAIR_LAUNCH(VAR, true, PAIR_CHEBY) else PAIR_LAUNCH(VAR, false, PAIR_CHEBY) } break;
  switch (variant) {
    PAIR_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ)
    PAIR_CASE(HPGMG_HIP_7PT_VC_POISSON)
    PAIR_CASE(HPGMG_HIP_7PT_CC)
    default: return record_error(hipErrorInvalidValue, "smooth_cheby_pair: variant");
  }
#undef PAIR_CASE
#undef PAIR_LAUNCH
#undef PAIR_LAUNCH2
#undef PAIR_LAUNCH_REMOTE
  if (part != 1) { g_pair_launches++; if (remote) g_pair_remote_launches++; }      // the two parts of a launch count once (part 2 is never empty: it holds the workgroups at the remote faces)
  profile_end(prof, 2 * cells);                      // one launch = two sweeps over every cell
  HPGMG_LAUNCH_CHECK("cheby_pair_kernel");
  return 0;
}
int hpgmg_hip_smooth_cheby_pair(const hpgmg_hip_level *L, int variant, double *const *scr_base, const float *const *c32_base,
                                int x0_scr, int x0_id, int xm1_scr, int xm1_id, int out1_scr, int out1_id, int out2_scr, int out2_id,
                                int rhs_id, double a, double b, double h2inv, double c1a, double c2a, double c1b, double c2b) {
  return smooth_pair(L, variant, 0, 0, scr_base, c32_base, x0_scr, x0_id, xm1_scr, xm1_id, out1_scr, out1_id, out2_scr, out2_id, rhs_id, a, b, h2inv, c1a, c2a, c1b, c2b);
}
void hpgmg_hip_pair_fold_interpolation(const hpgmg_hip_level *Lc, int coarse_id, double prescale) {
  g_pair_interp_level = Lc; g_pair_interp_id = coarse_id; g_pair_interp_prescale = prescale;
}
void hpgmg_hip_pair_set_halo(const int brick_boxes[3], const int remote_face[6], const double *deep, const double *deep_beta) {
  for (int d = 0; d < 3; d++) g_pair_brick[d] = brick_boxes[d];
  for (int d = 0; d < 6; d++) g_pair_rem[d] = remote_face[d];
  g_pair_deep = deep; g_pair_deep_beta = deep_beta; g_pair_halo_set = true;
}
void hpgmg_hip_pair_discard_x1(void) { g_pair_discard_x1 = 1; }
#ifdef HPGMG_EXP_TIMELINE
void hpgmg_hip_exp_timeline(void *buf) { g_exp_timeline = (double *)buf; }
#endif
void hpgmg_hip_pair_launch_counts(long long out[2]) { out[0] = g_pair_launches; out[1] = g_pair_remote_launches; }

// ---- the halo of a sweep pair across rank boundaries: one pack launch, one grouped send/recv, one unpack launch ----
static int pair_halo_move(bool unpack, const hpgmg_hip_level *L, double *const *scr_base, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id,
                          const hpgmg_hip_halo_entry *entries, int n, double *buf, double *deep, double *deep_beta) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  HaloRefs R; R.x0 = VecRef{x0_scr, x0_id}; R.xm1 = VecRef{xm1_scr, xm1_id}; R.rhs_id = rhs_id; R.scr_base = scr_base;
  const int slabs = (L->dim * L->dim + 4095) / 4096;                 // a face of dim^2 values: 16 values per lane
  if (unpack) hipLaunchKernelGGL((pair_halo_kernel<true>), dim3(n, slabs), dim3(256), 0, g_stream, *L, R, entries, buf, deep, deep_beta);
  else        hipLaunchKernelGGL((pair_halo_kernel<false>), dim3(n, slabs), dim3(256), 0, g_stream, *L, R, entries, buf, deep, deep_beta);
  HPGMG_LAUNCH_CHECK("pair_halo_kernel");
  return 0;
}
int hpgmg_hip_pair_halo_pack(const hpgmg_hip_level *L, double *const *scr_base, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id,
                             const hpgmg_hip_halo_entry *entries, int n, double *sendbuf) {
  return pair_halo_move(false, L, scr_base, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id, entries, n, sendbuf, nullptr, nullptr);
}
int hpgmg_hip_pair_halo_unpack(const hpgmg_hip_level *L, double *const *scr_base, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id,
                               const hpgmg_hip_halo_entry *entries, int n, double *recvbuf, double *deep, double *deep_beta) {
  return pair_halo_move(true, L, scr_base, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id, entries, n, recvbuf, deep, deep_beta);
}

// two consecutive in-place GSRB half sweeps (sweep, sweep + 1): x2 -> out2; the scratch vector `edge_scr_id` receives the
// few x1 values the kernel exchanges across 128-cell tile edges
int hpgmg_hip_smooth_gsrb_pair(const hpgmg_hip_level *L, int variant, double *const *scr_base, const float *const *c32_base,
                               int x0_scr, int x0_id, int edge_scr_id, int out2_scr, int out2_id, int rhs_id,
                               double a, double b, double h2inv, int sweep) {
  return smooth_pair(L, variant, 1, sweep, scr_base, c32_base, x0_scr, x0_id, x0_scr, x0_id, 1, edge_scr_id, out2_scr, out2_id, rhs_id, a, b, h2inv, 0.0, 0.0, 0.0, 0.0);
}
int hpgmg_hip_small_level_max_cells(void) { return 4096; }
// mode: 0 Chebyshev, 1 GSRB, 2 Jacobi (x_id <-> VECTOR_TEMP ping-pong as smooth() does; GSRB in place unless out_of_place), 3 residual
// (res_id = rhs - A x), 4 apply_op (res_id = A x); c1 / c2: per-sweep Chebyshev coefficients (Jacobi: c2 = the weight)
int hpgmg_hip_small_level_op(const hpgmg_hip_level *L, int variant, int mode, int sweeps, int x_id, int rhs_id, int res_id, int out_of_place,
                             double a, double b, double h2inv, const double *c1, const double *c2,
                             const blockCopy_type *copy_list, int n_copy, const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  if (sweeps < 1 || sweeps > 8 || mode < MODE_CHEBY || mode > MODE_APPLY || (long long)L->num_boxes * L->dim * L->dim * L->dim > 4096)
    return record_error(hipErrorInvalidValue, "small_level_op: arguments");
  SmallArgs A = {};
  A.mode = mode; A.sweeps = sweeps; A.x_id = x_id; A.rhs_id = rhs_id; A.res_id = res_id; A.out_of_place = out_of_place;
  A.bc_kind = n_bc > 0 ? bc_kind : 0; A.zero_first = zero_first; A.a = a; A.b = b; A.h2inv = h2inv;
  for (int q = 0; q < sweeps; q++) { A.c1[q] = c1 ? c1[q] : 0.0; A.c2[q] = c2 ? c2[q] : 0.0; }
  A.copy_list = copy_list; A.n_copy = copy_list ? n_copy : 0; A.bc_list = bc_list; A.n_bc = bc_list ? n_bc : 0;
  const long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  const int threads = cells >= 1024 ? 1024 : (cells >= 256 ? 256 : 64);
  // one box whose vectors fit the LDS: work on an image of it there
  const size_t image = (size_t)kSmallSlots * (size_t)L->volume * sizeof(double);
  static const int no_lds = env_int("HPGMG_TUNE_SMALL_NO_LDS", 0);
  A.lds_resident = (!no_lds && L->num_boxes == 1 && A.n_copy == 0 && image <= 150 * 1024) ? 1 : 0;
#ifdef HPGMG_EXP_TIMELINE
  A.timeline = (unsigned long long *)g_exp_timeline;
#endif
  const size_t lds = A.lds_resident ? image : 0;
  const int threads_used = A.lds_resident ? 1024 : threads;        // the image is copied by every lane there is
  if (!A.lds_resident) return record_error(hipErrorInvalidValue, "small_level_op: a level of one box whose vectors fit the LDS");
#define SMALL_CASE(VAR) { \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)small_level_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); once = true; } \
    hipLaunchKernelGGL((small_level_kernel<VAR>), dim3(1), dim3(threads_used), lds, g_stream, *L, A); }
  switch (variant) {
    case HPGMG_HIP_27PT_CC:          SMALL_CASE(HPGMG_HIP_27PT_CC) break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: SMALL_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ) break;
    case HPGMG_HIP_FV4_VC_POISSON:   SMALL_CASE(HPGMG_HIP_FV4_VC_POISSON) break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: SMALL_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ) break;
    case HPGMG_HIP_7PT_VC_POISSON:   SMALL_CASE(HPGMG_HIP_7PT_VC_POISSON) break;
    case HPGMG_HIP_7PT_CC:           SMALL_CASE(HPGMG_HIP_7PT_CC) break;
    default: return record_error(hipErrorInvalidValue, "small_level_op: variant");
  }
#undef SMALL_CASE
  HPGMG_LAUNCH_CHECK("small_level_kernel");
  return 0;
}

// BiCGStab bottom solve of the 27-point / fv2 / fv4 plugins on a level of one box (bottom_bicgstab_kernel): x_id holds the initial guess and
// receives the solution; the eight work vectors start at krylov_base; bc_list / bc_kind / zero_first as for hpgmg_hip_small_level_op;
// krylov_iterations: device-visible host counter the kernel adds its iteration count to, or NULL.  Dirichlet only.
int hpgmg_hip_bottom_bicgstab_max_cells(void) { return 512; }
int hpgmg_hip_bottom_bicgstab(const hpgmg_hip_level *L, int variant, int x_id, int rhs_id, int krylov_base, double a, double b, double h2inv, double want,
                              const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first, int *krylov_iterations) {
  HPGMG_SKIP_IF_REPLAY();
  const long long cells = (long long)L->dim * L->dim * L->dim;
  if (L->num_boxes != 1 || cells > 512 || L->periodic) return record_error(hipErrorInvalidValue, "bottom_bicgstab: a level of one box of at most 512 cells, Dirichlet");
  BottomArgs A = {};
  A.e_id = x_id; A.R_id = rhs_id; A.krylov_base = krylov_base; A.a = a; A.b = b; A.h2inv = h2inv; A.want = want;
  A.bc_list = bc_list; A.n_bc = bc_list ? n_bc : 0; A.bc_kind = A.n_bc > 0 ? bc_kind : 0; A.zero_first = zero_first; A.krylov_iterations = krylov_iterations;
  // eight waves whatever the level: the boundary entries (26 of them) are a wave's work each
  static const int tune_threads = env_int("HPGMG_TUNE_BOTTOM_THREADS", 512);
  const int threads = (cells > 256 || tune_threads >= 512) ? 512 : (cells > 64 || tune_threads >= 256 ? 256 : 64);
  const size_t lds = ((size_t)5 * L->volume + 1100) * sizeof(double);
  if (lds > 150 * 1024) return record_error(hipErrorInvalidValue, "bottom_bicgstab: box too large for the LDS image");
#define BOTTOM_CASE(VAR) { \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)bottom_bicgstab_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); once = true; } \
    hipLaunchKernelGGL((bottom_bicgstab_kernel<VAR>), dim3(1), dim3(threads), lds, g_stream, *L, A); }
  switch (variant) {
    case HPGMG_HIP_27PT_CC:          BOTTOM_CASE(HPGMG_HIP_27PT_CC) break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: BOTTOM_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ) break;
    case HPGMG_HIP_FV4_VC_POISSON:   BOTTOM_CASE(HPGMG_HIP_FV4_VC_POISSON) break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: BOTTOM_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ) break;
    case HPGMG_HIP_7PT_VC_POISSON:   BOTTOM_CASE(HPGMG_HIP_7PT_VC_POISSON) break;
    case HPGMG_HIP_7PT_CC:           BOTTOM_CASE(HPGMG_HIP_7PT_CC) break;
    default: return record_error(hipErrorInvalidValue, "bottom_bicgstab: variant");
  }
#undef BOTTOM_CASE
  HPGMG_LAUNCH_CHECK("bottom_bicgstab_kernel");
  return 0;
}
// A queue of BLAS-1 / operator calls on a level of one box of <= 512 cells as one launch (small_ops_kernel).  kinds: 1 add (c = sa*a + sb*b),
// 2 mul (c = sa*a*b), 3 scale (c = sa*a), 4 apply_op (c = A a), 5 residual (c = b - A a), 6 dot (a, b), 7 norm (a); a value-returning
// operation may only come last, its value goes to *value_out (the call then waits for it).
static long long g_small_ops_launches = 0;
long long hpgmg_hip_small_ops_launch_count(void) { return g_small_ops_launches; }
int hpgmg_hip_small_ops_max(void) { return kSmallOpsMax; }
int hpgmg_hip_small_ops(const hpgmg_hip_level *L, int variant, int n, const int *kinds, const int *c, const int *a, const int *b, const double *sa, const double *sb,
                        const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first, double op_a, double op_b, double h2inv, double *value_out, double *value2_out) {
  if (int e = hpgmg_hip_graph_flush()) return e;
  if (n < 1 || n > kSmallOpsMax || L->num_boxes != 1 || L->dim > 8 || L->periodic) return record_error(hipErrorInvalidValue, "small_ops: one Dirichlet box of side <= 8, 1..12 operations");
  SmallOpsArgs A = {};
  A.n = n; A.bc_list = bc_list; A.n_bc = bc_list ? n_bc : 0; A.bc_kind = A.n_bc > 0 ? bc_kind : 0; A.zero_first = zero_first; A.a = op_a; A.b = op_b; A.h2inv = h2inv;
  int nvalues = 0;
  for (int q = 0; q < n; q++) {
    const bool is_value = (kinds[q] == SO_DOT || kinds[q] == SO_NORM);
    if (kinds[q] < SO_ADD || kinds[q] > SO_NORM || (is_value && q < n - 2) || (is_value && q == n - 2 && !(kinds[n - 1] == SO_DOT || kinds[n - 1] == SO_NORM)))
      return record_error(hipErrorInvalidValue, "small_ops: operation list (value-returning operations only as the last one or two entries)");
    A.op[q].kind = kinds[q]; A.op[q].c = c[q]; A.op[q].a = a[q]; A.op[q].b = b[q]; A.op[q].sa = sa[q]; A.op[q].sb = sb[q];
    nvalues += is_value ? 1 : 0;
  }
  const bool wants = nvalues > 0;
  if ((nvalues >= 1) != (value_out != nullptr) || (nvalues == 2) != (value2_out != nullptr)) return record_error(hipErrorInvalidValue, "small_ops: one output pointer per value-returning operation");
  if (wants) { A.result = reduction_slot_next(&A.seq); if (!A.result) return record_error(hipErrorOutOfMemory, "small_ops: result slot"); }
  const int cells = L->dim * L->dim * L->dim;
  const int threads = cells > 64 ? 512 : (A.n_bc > 4 ? 512 : 64);           // one cell per lane (no striding); the boundary entries are a wave's work each
#define SMALL_OPS_CASE(VAR) hipLaunchKernelGGL((small_ops_kernel<VAR>), dim3(1), dim3(threads), 0, g_stream, *L, A);
  switch (variant) {
    case HPGMG_HIP_27PT_CC:          SMALL_OPS_CASE(HPGMG_HIP_27PT_CC) break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: SMALL_OPS_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ) break;
    case HPGMG_HIP_FV4_VC_POISSON:   SMALL_OPS_CASE(HPGMG_HIP_FV4_VC_POISSON) break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: SMALL_OPS_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ) break;
    case HPGMG_HIP_7PT_VC_POISSON:   SMALL_OPS_CASE(HPGMG_HIP_7PT_VC_POISSON) break;
    case HPGMG_HIP_7PT_CC:           SMALL_OPS_CASE(HPGMG_HIP_7PT_CC) break;
    default: return record_error(hipErrorInvalidValue, "small_ops: variant");
  }
#undef SMALL_OPS_CASE
  g_small_ops_launches++;
  HPGMG_LAUNCH_CHECK("small_ops_kernel");
  if (wants) { if (int e = reduction_fetch(value_out)) return e; if (value2_out) *value2_out = reduction_second_value(); }
  return 0;
}
// V-cycle tail below a level of one box (small_vtail_kernel).  The argument block lives in device memory: it is the same for every visit of
// a chain in a solve, so a few of them are kept and uploaded only when their contents change.
long long hpgmg_hip_small_vtail_lds_limit(void) { return 150 * 1024 / (long long)sizeof(double); }
long long hpgmg_hip_small_vtail_lds_doubles(const hpgmg_hip_small_tail_args *T) {
  long long need = 0;
  for (int l = 0; l + 1 < T->n; l++) { const long long v = 8LL * T->lv[l].L.volume + T->lv[l + 1].L.volume; if (v > need) need = v; }
  const long long bottom = 5LL * T->lv[T->n - 1].L.volume + 1100;
  return bottom > need ? bottom : need;
}
static long long g_small_vtail_launches = 0;
long long hpgmg_hip_small_vtail_launch_count(void) { return g_small_vtail_launches; }
int hpgmg_hip_small_vtail(const hpgmg_hip_small_tail_args *T, int variant) {
  HPGMG_SKIP_IF_REPLAY();
  if (!T || T->n < 2 || T->n > HPGMG_HIP_SMALL_TAIL_MAX_LEVELS || T->legs < 1 || T->legs > 7 || T->sweeps < 1 || T->sweeps > 8 || T->mode < MODE_CHEBY || T->mode > MODE_JACOBI)
    return record_error(hipErrorInvalidValue, "small_vtail: arguments");
  for (int l = 0; l < T->n; l++) {
    const hpgmg_hip_level &L = T->lv[l].L;
    if (L.num_boxes != 1 || L.periodic || (l > 0 && 2 * L.dim != T->lv[l - 1].L.dim)) return record_error(hipErrorInvalidValue, "small_vtail: a chain of levels of one box, halving, Dirichlet");
  }
  if ((T->legs & 2) && (long long)T->lv[T->n - 1].L.dim * T->lv[T->n - 1].L.dim * T->lv[T->n - 1].L.dim > 512) return record_error(hipErrorInvalidValue, "small_vtail: bottom level too large");
  const long long need = hpgmg_hip_small_vtail_lds_doubles(T);
  if (need > hpgmg_hip_small_vtail_lds_limit()) return record_error(hipErrorInvalidValue, "small_vtail: the chain does not fit the LDS");
  constexpr int kSlots = 8;
  static hpgmg_hip_small_tail_args host_copy[kSlots];
  static hpgmg_hip_small_tail_args *dev_copy[kSlots];
  static int used = 0, next = 0;
  int slot = -1;
  for (int q = 0; q < used; q++) if (memcmp(&host_copy[q], T, sizeof *T) == 0) { slot = q; break; }
  if (slot < 0) {
    slot = (used < kSlots) ? used++ : (next++ % kSlots);
    if (!dev_copy[slot]) HPGMG_CHECK(hipMalloc((void **)&dev_copy[slot], sizeof *T));
    host_copy[slot] = *T;
    HPGMG_CHECK(hipMemcpyAsync(dev_copy[slot], &host_copy[slot], sizeof *T, hipMemcpyHostToDevice, g_stream));   // stream order: after every launch that reads the slot
  }
  const size_t lds = (size_t)need * sizeof(double);
  unsigned long long *tl = nullptr;
#ifdef HPGMG_EXP_TIMELINE
  tl = (unsigned long long *)g_exp_timeline;
#endif
#define VTAIL_CASE(VAR) { \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)small_vtail_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); once = true; } \
    hipLaunchKernelGGL((small_vtail_kernel<VAR>), dim3(1), dim3(512), lds, g_stream, (const hpgmg_hip_small_tail_args *)dev_copy[slot], tl); }
  switch (variant) {
    case HPGMG_HIP_27PT_CC:          VTAIL_CASE(HPGMG_HIP_27PT_CC) break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: VTAIL_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ) break;
    case HPGMG_HIP_FV4_VC_POISSON:   VTAIL_CASE(HPGMG_HIP_FV4_VC_POISSON) break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: VTAIL_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ) break;
    case HPGMG_HIP_7PT_VC_POISSON:   VTAIL_CASE(HPGMG_HIP_7PT_VC_POISSON) break;
    default: return record_error(hipErrorInvalidValue, "small_vtail: variant");
  }
#undef VTAIL_CASE
  g_small_vtail_launches++;
  HPGMG_LAUNCH_CHECK("small_vtail_kernel");
  return 0;
}
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
  if (g_tile_part) {      // one part of the launch: part 1 = the tiles that read nothing of an image of another rank's box
    int count = 0;
    A.order = tile_part_order(L, A.tiles_i, A.tiles_j, A.chunks_k, g_tile_part, false, &grid, &A.per_xcd, &count);
    if (grid == 0) return 0;
    if (!A.order) return record_error(hipErrorOutOfMemory, "smooth_gsrb27_rb: dispatch list of a partial launch");
    cells = cells * count / A.total_blocks;
  }
  const int prof = profile_begin(cells);
  hipLaunchKernelGGL((stencil27_rb_kernel<TJ>), dim3(grid), dim3(64, TJ / 2), 0, g_stream, *L, A);
  g_rb27_launches++;
  profile_end(prof, 2 * cells);                       // one launch = two half sweeps over every cell
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
  if (!wide_fused_ok(L, variant) || Lc->num_boxes <= 0) return record_error(hipErrorInvalidValue, "residual_restrict: level not supported");
  if (res_id == x_id || res_id == rhs_id) return record_error(hipErrorInvalidValue, "residual_restrict: the residual may not overwrite its inputs");
  StencilArgs P = {}; P.xn_id = x_id; P.xout_id = (res_id >= 0) ? res_id : x_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv;
  FusedArgs F = {}; F.Lc = *Lc; F.coarse_id = coarse_id; F.zero_id = zero_id; F.map = map; F.store_res = (res_id >= 0);
  F.zero_chunks_per_box = (Lc->volume + 4095) / 4096;
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
  return finish_max_reduction(blocks, 0.0, norm_out);
}

int hpgmg_hip_residual(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id, double a, double b, double h2inv) {
  StencilArgs P = {}; P.xn_id = x_id; P.xout_id = res_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv;
  if (rhs_id < 0) return launch<MODE_APPLY>(L, variant, P, false);
  return launch<MODE_RESIDUAL>(L, variant, P, false);
}

}  // extern "C"

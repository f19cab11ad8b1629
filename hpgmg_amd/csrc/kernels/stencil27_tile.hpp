// stencil27_tile.hpp -- the 27-point constant-coefficient operator (reference operators.27pt.c:48-51,60-91) as an LDS-staged,
// k-marching kernel for boxes whose side is a multiple of 64 (TI = 64) or of 32 (TI = 32: a wave is two rows of a 32 x TJ tile).
//
// stencil27_kernel (stencil.hip) keeps the three 3 x 3 planes around a cell in registers and re-reads 9 values per step through
// the vector L1: 72 B of L1 traffic per cell and step, which bounds it (8.8 TB/s of L1 traffic at 1.10 ms per coloured half sweep of
// 512^3).  Here a 64 x TJ workgroup owns a 64 (i) x TJ (j) tile of a box and marches in +k with planes k-1, k, k+1 of x in LDS
// (ring of three (TJ+2) x 66 tiles): a lane stores its own value (loaded one step earlier into a register) and at most one halo
// cell, so x enters LDS once per workgroup, and the update reads its 26 neighbours from LDS (a wave reads 64 consecutive doubles:
// conflict free).  The weighted sums are formed exactly as apply_op_27pt forms them -- ((C3*corners + C2*edges) + C1*faces) + C0*centre,
// each group summed left to right in the listed order -- so results are bit-identical to the register kernel and to the reference.
// P.ghost_free (all boxes local): x outside the box is read from the neighbouring box itself (common.hpp gf_column), the caller then
// runs only apply_BCs_p2 before the launch, no exchange_boundary.
#pragma once
#include "common.hpp"

#ifndef C27_0
#define C27_0 (-4.2666666666666666666)
#define C27_1 ( 0.4666666666666666666)
#define C27_2 ( 0.1000000000000000000)
#define C27_3 ( 0.0333333333333333333)
#endif

namespace hpgmg {
// LDS reads of the 27-point stencils: volatile, so that the backend issues ds_read_b64 and does not merge pairs of them into ds_read2_b64 (half the bytes per LDS cycle on gfx950)
typedef const volatile double __attribute__((address_space(3))) *lds27r;

struct S27TileArgs {
  int xn_id, xout_id, rhs_id, mode;     // mode: MODE_* of stencil.hip (Chebyshev, GSRB, Jacobi, residual, apply_op)
  double a, b, h2inv, c1, c2;
  int sweep, ghost_free;
  TileFused fused;                      // MODE 3 only: what becomes of the residual (common.hpp)
  int tiles_i, tiles_j, chunks_k, kchunk, per_xcd, total_blocks;
  const int *order;                     // dispatch slot -> tile (nullptr: identity): two-part launches across rank boundaries (common.hpp tile_part_order)
};

template <int MODE, int TJ, int TI = 64>
__global__ __launch_bounds__(TI * TJ) void stencil27_tile_kernel(const hpgmg_hip_level L, const S27TileArgs P) {
  constexpr int W = TI + 2, H = TJ + 2, NT = TI * TJ, PLANE = W * H;
  constexpr int NH = 2 * W + 2 * TJ;                            // halo cells of one plane tile
  static_assert(NH <= NT, "one halo cell per lane at most");
  constexpr bool kSmooth = (MODE == 0 || MODE == 1 || MODE == 2);
  __shared__ double sX[3 * PLANE];
  __shared__ double sR[(MODE == 3) ? 2 * TJ * TI : 1];           // fused residual forms: a plane of residuals / the workgroup's partial maxima

  int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (P.order) logical = P.order[logical];
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;
  const int li = (int)threadIdx.x, lj = (int)threadIdx.y, tid = lj * TI + li;
  const int i0 = ti * TI, j0 = tj * TJ, i = i0 + li, j = j0 + lj;
  const int k0 = ck * P.kchunk, k1 = (k0 + P.kchunk < L.dim) ? k0 + P.kchunk : L.dim;
  const int jS = L.jStride, kS = L.kStride;

  gcptr x = gvec_origin(L, box, P.xn_id);
  gptr out = gvec_origin(L, box, P.xout_id);      // 27-pt GSRB is out of place, Chebyshev / Jacobi ping-pong: never aliases x
  gcptr rhs = (MODE == 4) ? nullptr : gvec_origin(L, box, P.rhs_id);
  gcptr dinv = kSmooth ? gvec_origin(L, box, VECTOR_DINV) : nullptr;
  int colour000 = 0;
  if (MODE == 1) colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;

  const int own_g = i + j * jS, own_s = (lj + 1) * W + (li + 1);
  int halo_g = 0, halo_s = 0;
  GfColumn hcol = {box, 0};
  const bool has_halo = tid < NH;
  if (has_halo) {
    int hi, hj;
    if (tid < W)          { hj = -1; hi = -1 + tid; }
    else if (tid < 2 * W) { hj = TJ; hi = -1 + (tid - W); }
    else                  { const int h = tid - 2 * W; hj = h >> 1; hi = (h & 1) ? TI : -1; }
    halo_g = (i0 + hi) + (j0 + hj) * jS;
    halo_s = (hj + 1) * W + (hi + 1);
    if (P.ghost_free) hcol = gf_column(L, box, i0 + hi, j0 + hj);
  }
  gcptr xh = (has_halo && P.ghost_free) ? gvec_origin(L, hcol.box, P.xn_id) + hcol.off : x + halo_g;
  const bool gf = P.ghost_free != 0;
  const int dim = L.dim;
  // planes below the box (p < 0) are only met in the prologue of the first chunk: looked up there.  Planes above it (p >= dim) are met
  // in the last steps of the last chunk: one alternative base pointer per column, selected by p, keeps the marching loop free of branches
  gcptr xo_hi = x + own_g, xh_hi = xh;
  if (gf && k1 == dim) {
    const int n = L.box_nbr[6 * box + 5];
    if (n >= 0) xo_hi = gvec_origin(L, n, P.xn_id) + own_g - (long long)dim * kS;
    if (has_halo) { const int m = L.box_nbr[6 * hcol.box + 5]; if (m >= 0) xh_hi = gvec_origin(L, m, P.xn_id) + hcol.off - (long long)dim * kS; }
  }
  auto x_own = [&](int p) -> double {
    if (gf && p < 0) return gf_load_outside(L, P.xn_id, GfColumn{box, own_g}, p);
    return ((p >= dim) ? xo_hi : x + own_g)[p * kS];
  };
  auto x_halo = [&](int p) -> double {
    if (gf && p < 0) return gf_load_outside(L, P.xn_id, hcol, p);
    return ((p >= dim) ? xh_hi : xh)[p * kS];
  };
  auto x_own_fwd = [&](int p) -> double { return ((p >= dim) ? xo_hi : x + own_g)[p * kS]; };     // p >= 0: the marching loop
  auto x_halo_fwd = [&](int p) -> double { return ((p >= dim) ? xh_hi : xh)[p * kS]; };
  auto slot3 = [](int p) { return ((p % 3) + 3) % 3; };

  // prologue: planes k0-1 and k0 into LDS; the plane k0+1 values and the per-cell streams of plane k0 in flight
  for (int p = k0 - 1; p <= k0; p++) {
    const int s = slot3(p) * PLANE;
    sX[s + own_s] = x_own(p);
    if (has_halo) sX[s + halo_s] = x_halo(p);
  }
  double n_x = x_own(k0 + 1), h_x = has_halo ? x_halo(k0 + 1) : 0.0;
  double c_rhs = (MODE == 4) ? 0.0 : rhs[own_g + k0 * kS], c_dinv = kSmooth ? dinv[own_g + k0 * kS] : 0.0;
  double c_old = (MODE == 0) ? out[own_g + k0 * kS] : 0.0;
  const TileFused &F = P.fused;
  TileFusedState<TI, TJ> fs;
  gptr coarse = nullptr;
  if (MODE == 3 && F.kind == 2) {
    const int *mp = F.map + 4 * box;
    coarse = gvec_origin(F.Lc, mp[0], F.coarse_id) + (mp[1] + (i >> 1)) + (mp[2] + (j >> 1)) * F.Lc.jStride + (mp[3] + (k0 >> 1)) * F.Lc.kStride;
  }

  for (int k = k0; k < k1; k++) {
    const int pg = k * kS;
    __syncthreads();                                            // every wave is done reading the slot that plane k+1 overwrites
    if (MODE == 3 && F.kind == 2 && k > k0) fs.gather(F, sR, li, lj, k - 1, k0, coarse);
    { const int s = slot3(k + 1) * PLANE; sX[s + own_s] = n_x; if (has_halo) sX[s + halo_s] = h_x; }
    double nn_rhs = 0, nn_dinv = 0, nn_old = 0;
    if (k + 1 < k1) {                                           // loads of the next step
      const int cg = own_g + (k + 1) * kS;
      n_x = x_own_fwd(k + 2);
      if (has_halo) h_x = x_halo_fwd(k + 2);
      if (MODE != 4) nn_rhs = rhs[cg];
      if (kSmooth) nn_dinv = dinv[cg];
      if (MODE == 0) nn_old = out[cg];
    }
    __syncthreads();

    const double *c = sX + slot3(k) * PLANE + own_s, *m = sX + slot3(k - 1) * PLANE + own_s, *p = sX + slot3(k + 1) * PLANE + own_s;      // (plain reads here: with the ds_read_b64 form of the red + black kernel this kernel measured 0.3 % slower)
    const double xc = c[0];
    bool update = true;
    if (MODE == 1) update = (((i ^ j ^ k ^ colour000) & 1) == 0);
    if (update) {
      // operators.27pt.c:60-91 in apply_op_27pt's order: 8 corners, 12 edges, 6 faces, centre
      double s8 = m[-W - 1] + m[-W + 1]; s8 = s8 + m[W - 1]; s8 = s8 + m[W + 1];
      s8 = s8 + p[-W - 1]; s8 = s8 + p[-W + 1]; s8 = s8 + p[W - 1]; s8 = s8 + p[W + 1];
      double s12 = m[-W] + m[-1]; s12 = s12 + m[1]; s12 = s12 + m[W];
      s12 = s12 + c[-W - 1]; s12 = s12 + c[-W + 1]; s12 = s12 + c[W - 1]; s12 = s12 + c[W + 1];
      s12 = s12 + p[-W]; s12 = s12 + p[-1]; s12 = s12 + p[1]; s12 = s12 + p[W];
      double s6 = m[0] + c[-W]; s6 = s6 + c[-1]; s6 = s6 + c[1]; s6 = s6 + c[W]; s6 = s6 + p[0];
      double tt = C27_3 * s8 + C27_2 * s12;
      tt = tt + C27_1 * s6;
      tt = tt + C27_0 * xc;
      const double Ax = P.a * xc - (P.b * P.h2inv) * tt;
      double o;
      if (MODE == 0)      o = xc + P.c1 * (xc - c_old) + P.c2 * c_dinv * (c_rhs - Ax);
      else if (MODE == 1) o = xc + c_dinv * (c_rhs - Ax);
      else if (MODE == 2) o = xc + P.c2 * c_dinv * (c_rhs - Ax);
      else if (MODE == 3) o = c_rhs - Ax;
      else                o = Ax;
      if (MODE == 3 && F.kind == 1) { const double f = fabs(o); fs.lane_max = (f > fs.lane_max) ? f : fs.lane_max; }
      else if (MODE == 3 && F.kind == 2) sR[((k & 1) * TJ + lj) * TI + li] = o;
      else out[own_g + pg] = o;
    } else {
      out[own_g + pg] = xc;                                     // out-of-place GSRB copies the other colour (gsrb.c:94-98)
    }
    c_rhs = nn_rhs; c_dinv = nn_dinv; c_old = nn_old;
  }
  if (MODE == 3 && F.kind == 2) { __syncthreads(); fs.gather(F, sR, li, lj, k1 - 1, k0, coarse); }
  if (MODE == 3 && F.kind == 1) {                                 // a maximum is exact under any order (misc.c:307-317)
    double mx = fs.lane_max;
    for (int off = 32; off > 0; off >>= 1) { const double o2 = __shfl_down(mx, off, 64); mx = (o2 > mx) ? o2 : mx; }
    __syncthreads();
    if ((tid & 63) == 0) sR[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) { double m2 = sR[0]; for (int q = 1; q < NT / 64; q++) m2 = (sR[q] > m2) ? sR[q] : m2; F.partials[logical] = m2; }
  }
}

}  // namespace hpgmg

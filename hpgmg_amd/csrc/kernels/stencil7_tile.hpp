// stencil7_tile.hpp -- one sweep of the 7-point operator for boxes whose side is a multiple of 64 but not of 128 (the 128^3 level of
// config 2: 8 boxes of 64^3), LDS-staged.
//
// stencil7_kernel re-reads the +-i / +-j neighbours of x through the vector L1: 14 eight-byte accesses per cell and step, 8.7 TB/s of L1
// traffic at 27 us per Chebyshev sweep of the 128^3 level -- the same L1 ceiling the first fv4 / 27-point kernels sat at.  Here a 64 x TJ
// workgroup owns a 64 (i) x TJ (j) tile and marches in +k with plane k of x in LDS (double buffered, one barrier per step): a lane stores
// its own value and at most one halo cell, reads its four in-plane neighbours from LDS, keeps x[k-1], x[k+1] and beta_k[k] of its column in
// registers and gets beta_i's high face from the next lane -- 10 accesses per cell instead of 14, all issued one step ahead.
// Same expression (apply_op_7pt + the smoother updates) and the same ghost-free face rules as stencil7_kernel: a face neighbour in
// another local box is read from that box, a Dirichlet face is ghost = -centre, a face of another rank comes from the ghost zone.
#pragma once
#include "common.hpp"
#include "stencil_math.hpp"
#include "stencil_direct.hpp"      // InterpFold

namespace hpgmg {

struct S7TileArgs {
  int xn_id, xout_id, rhs_id;
  double a, b, h2inv, c1, c2;
  int sweep, ghost_free;
  int tiles_i, tiles_j, chunks_k, kchunk, per_xcd, total_blocks;
};

// IP: the piecewise-constant interpolation folded into the first two sweeps of a smooth() (InterpFold, stencil_direct.hpp); every box local.
// RR (MODE 3): residual + restriction + zero_vector of MGVCycle's down leg (mg.c:1150-1153) in one launch -- a plane of residuals passes through an LDS
// tile, the lane of an even (i, j) sums its 2 x 2 patch a step later in restriction.c:54-57's order and carries it over the plane pair (TileFusedState,
// common.hpp); the coarse zero_vector rides along as extra workgroups; the residual itself is stored only when FA.store_res says so.
template <int V, int MODE, int TJ, bool IP = false, bool RR = false>
__global__ __launch_bounds__(64 * TJ) void stencil7_tile_kernel(const hpgmg_hip_level L, const S7TileArgs P, const InterpFold F = InterpFold{}, const FusedArgs FA = FusedArgs{}) {
  constexpr int TI = 64, W = TI + 2, H = TJ + 2, NT = 64 * TJ, PLANE = W * H;
  constexpr int NH = 2 * TI + 2 * TJ;                           // halo cells of a plane tile: a row above and below, a column left and right (no corners: star stencil)
  static_assert(NH <= NT, "one halo cell per lane at most");
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  constexpr bool kSmooth = (MODE == 0 || MODE == 1 || MODE == 2);  // Chebyshev, GSRB, Jacobi | 3 residual, 4 apply_op
  __shared__ double sX[2 * PLANE];
  __shared__ double sR[RR ? 2 * TJ * TI : 1];

  if (RR && (int)blockIdx.x >= kXcds * P.per_xcd) {      // zero_vector(coarse, zero_id): whole padded boxes, ghosts included
    const int z = (int)blockIdx.x - kXcds * P.per_xcd, zbox = z / FA.zero_chunks_per_box, chunk = z - zbox * FA.zero_chunks_per_box;
    if (zbox >= FA.Lc.num_boxes) return;
    double *v = FA.Lc.box_base[zbox] + (size_t)FA.zero_id * (size_t)FA.Lc.volume;
    const int lo = chunk * 4096, hi = (lo + 4096 < FA.Lc.volume) ? lo + 4096 : FA.Lc.volume;
    for (int q = lo + (int)(threadIdx.y * 64 + threadIdx.x); q < hi; q += NT) v[q] = 0.0;
    return;
  }
  const int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;
  const int li = (int)threadIdx.x, lj = (int)threadIdx.y, tid = lj * 64 + li;
  const int i0 = ti * TI, j0 = tj * TJ, i = i0 + li, j = j0 + lj;
  const int k0 = ck * P.kchunk, k1 = (k0 + P.kchunk < L.dim) ? k0 + P.kchunk : L.dim;
  const int jS = L.jStride, kS = L.kStride, last = L.dim - 1;
  const bool gf = P.ghost_free != 0;

  gcptr x = gvec_origin(L, box, P.xn_id);                  // may alias out (in-place GSRB): a swept cell's neighbours all have the other colour
  gptr out = gvec_origin(L, box, P.xout_id);
  gcptr rhs = (MODE == 4) ? nullptr : gvec_origin(L, box, P.rhs_id);
  gcptr dinv = kSmooth ? gvec_origin(L, box, VECTOR_DINV) : nullptr;
  gcptr alpha = kHelm ? gvec_origin(L, box, VECTOR_ALPHA) : nullptr;
  gcptr beta_i = kVC ? gvec_origin(L, box, VECTOR_BETA_I) : nullptr;
  gcptr beta_j = kVC ? gvec_origin(L, box, VECTOR_BETA_J) : nullptr;
  gcptr beta_k = kVC ? gvec_origin(L, box, VECTOR_BETA_K) : nullptr;
  int colour000 = 0;
  if (MODE == 1) colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;

  const int own_g = i + j * jS, own_s = (lj + 1) * W + (li + 1);
  // this lane's halo cell (if any): where it sits in the tile, and how its value is obtained on plane k:
  //   kind 0: plain load (inside the box, or the ghost zone);  1: from the neighbouring local box;  2: Dirichlet, minus the adjacent interior cell
  const bool has_halo = tid < NH;
  int halo_s = 0, halo_g = 0, halo_kind = 0, halo_box = box, halo_i = 0, halo_j = 0;
  const bool fold_x = IP && F.which == 1, fold_old = IP && F.which == 2;
  gcptr halo_x = x;
  if (has_halo) {
    int hi, hj;
    if (tid < TI)          { hj = -1; hi = tid; }
    else if (tid < 2 * TI) { hj = TJ; hi = tid - TI; }
    else                   { const int h = tid - 2 * TI; hj = h >> 1; hi = (h & 1) ? TI : -1; }
    halo_s = (hj + 1) * W + (hi + 1);
    int ci = i0 + hi, cj = j0 + hj;
    int dir = -1;
    if (ci < 0) dir = 0; else if (ci > last) dir = 1; else if (cj < 0) dir = 2; else if (cj > last) dir = 3;
    if (dir >= 0 && gf) {
      const int nb = L.box_nbr[6 * box + dir];
      if (nb >= 0) { halo_kind = 1; halo_box = nb; halo_x = gvec_origin(L, nb, P.xn_id); if (dir == 0) ci = last; else if (dir == 1) ci = 0; else if (dir == 2) cj = last; else cj = 0; }
      else if (nb == -1) { halo_kind = 2; if (dir == 0) ci = 0; else if (dir == 1) ci = last; else if (dir == 2) cj = 0; else cj = last; }
    }
    halo_g = ci + cj * jS; halo_i = ci; halo_j = cj;
  }
  auto halo_at = [&](int k) -> double {
    double v = halo_x[halo_g + k * kS];
    if (fold_x) v = v + fold_coarse(F, halo_box, halo_i, halo_j, k);
    return halo_kind == 2 ? -v : v;
  };
  auto own_at = [&](int kk) -> double { const double v = x[own_g + kk * kS]; return fold_x ? v + fold_coarse(F, box, i, j, kk) : v; };
  // value of x just below / above the box on this lane's column (k faces).  The two neighbour numbers are read HERE, once: looked up inside
  // the march, the load was hoisted above its (rarely taken) branch and its wait -- vmcnt(0), in front of the prefetches just issued for the
  // next plane -- made every step a full round trip to memory.
  const int nb_k[2] = { L.box_nbr ? L.box_nbr[6 * box + 4] : -3, L.box_nbr ? L.box_nbr[6 * box + 5] : -3 };
  auto outside_k = [&](int dir, double centre, int kk) -> double {
    const int nb = nb_k[dir - 4];
    if (nb >= 0) { const double v = gvec_origin(L, nb, P.xn_id)[own_g + (dir == 4 ? last : 0) * kS]; return fold_x ? v + fold_coarse(F, nb, i, j, dir == 4 ? last : 0) : v; }
    if (nb == -1) return -centre;
    return x[own_g + kk * kS];
  };

  // registers: x[k-1], x[k], x[k+1] of the own column, the halo value of plane k, beta_k[k]; the streams of plane k in flight
  double xc = own_at(k0);
  double xm = (gf && k0 == 0) ? outside_k(4, xc, -1) : own_at(k0 - 1);
  double xp = (gf && k0 == last) ? outside_k(5, xc, k0 + 1) : own_at(k0 + 1);
  double h_c = has_halo ? halo_at(k0) : 0.0;
  double bk0 = kVC ? beta_k[own_g + k0 * kS] : 0.0;
  double c_bi = 0, c_bir = 0, c_bj0 = 0, c_bj1 = 0, c_bk1 = 0, c_al = 0, c_dinv = 0, c_rhs = 0, c_old = 0;
  auto load_streams = [&](int k, double &bi, double &bir, double &bj0, double &bj1, double &bk1, double &al, double &dv, double &rh, double &old) {
    const int g = own_g + k * kS;
    if (kVC) { bi = beta_i[g]; bir = (li == 63) ? beta_i[g + 1] : 0.0; bj0 = beta_j[g]; bj1 = beta_j[g + jS]; bk1 = beta_k[g + kS]; }
    if (kHelm) al = alpha[g];
    if (kSmooth) dv = dinv[g];
    if (MODE != 4) rh = rhs[g];
    if (MODE == 0) { old = out[g]; if (fold_old) old = old + fold_coarse(F, box, i, j, k); }
  };
  load_streams(k0, c_bi, c_bir, c_bj0, c_bj1, c_bk1, c_al, c_dinv, c_rhs, c_old);
  TileFused TF = {}; TileFusedState<TI, TJ> fs; gptr coarse = nullptr;
  if (RR) {
    const int *mp = FA.map + 4 * box;
    TF.kind = 2; TF.Lc = FA.Lc;
    coarse = gvec_origin(FA.Lc, mp[0], FA.coarse_id) + (mp[1] + (i >> 1)) + (mp[2] + (j >> 1)) * FA.Lc.jStride + (mp[3] + (k0 >> 1)) * FA.Lc.kStride;
  }

  for (int k = k0; k < k1; k++) {
    double *s = sX + (k & 1) * PLANE;
    s[own_s] = xc;
    if (has_halo) s[halo_s] = h_c;
    // next step's loads (their latency overlaps this step's barrier and arithmetic)
    double n_xp = 0, n_h = 0, n_bi = 0, n_bir = 0, n_bj0 = 0, n_bj1 = 0, n_bk1 = 0, n_al = 0, n_dinv = 0, n_rhs = 0, n_old = 0;
    if (k + 1 < k1) {
      n_xp = (gf && k + 1 == last) ? 0.0 : own_at(k + 2);
      if (has_halo) n_h = halo_at(k + 1);
      load_streams(k + 1, n_bi, n_bir, n_bj0, n_bj1, n_bk1, n_al, n_dinv, n_rhs, n_old);
    }
    __syncthreads();
    if (RR && k > k0) fs.gather(TF, sR, li, lj, k - 1, k0, coarse);      // the residuals of plane k-1 are all in sR (stored before this barrier)
    bool update = true;
    if (MODE == 1) update = (((i ^ j ^ k ^ colour000) & 1) == 0);
    // beta_i's high face = the next lane's low face (lane 63 loaded it)
    double bi1 = __shfl_down(c_bi, 1, 64);
    if (li == 63) bi1 = c_bir;
    if (update) {
      const double *c = s + own_s;
      const double Ax = apply_op_7pt<V>(xc, c[-1], c[1], c[-W], c[W], xm, xp, c_bi, bi1, c_bj0, c_bj1, bk0, c_bk1, c_al, P.a, P.b, P.h2inv);
      double o;
      if (MODE == 0)      o = xc + P.c1 * (xc - c_old) + P.c2 * c_dinv * (c_rhs - Ax);
      else if (MODE == 1) o = xc + c_dinv * (c_rhs - Ax);
      else if (MODE == 2) o = xc + P.c2 * c_dinv * (c_rhs - Ax);
      else if (MODE == 3) o = c_rhs - Ax;
      else                o = Ax;
      if (!RR || FA.store_res) out[own_g + k * kS] = o;
      if (RR) sR[((k & 1) * TJ + lj) * TI + li] = o;
    }
    // rotate; at the top of the box the plane above is the face rule applied to the new centre
    xm = xc; xc = xp;
    xp = (gf && k + 1 == last) ? outside_k(5, xc, k + 2) : n_xp;
    h_c = n_h; bk0 = c_bk1;
    c_bi = n_bi; c_bir = n_bir; c_bj0 = n_bj0; c_bj1 = n_bj1; c_bk1 = n_bk1; c_al = n_al; c_dinv = n_dinv; c_rhs = n_rhs; c_old = n_old;
  }
  if (RR) { __syncthreads(); fs.gather(TF, sR, li, lj, k1 - 1, k0, coarse); }
}

}  // namespace hpgmg

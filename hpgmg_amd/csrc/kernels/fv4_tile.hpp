// fv4_tile.hpp -- the 4th-order finite-volume operator (reference operators.fv4.c:55-134) as an LDS-tiled, k-marching kernel.
//
// The 25-point variable-coefficient stencil reads, per updated cell, 25 values of x and 30 face coefficients.  Read through
// the vector L1 (stencil_direct_kernel, the first version) that is ~58 eight-byte loads per update and the kernel is bound by
// the L1 / texture-address path at ~9 TB/s of cache traffic (3.3 ms per half sweep at 512^3 = 28 % of the HBM roofline).
// Here a workgroup owns a TI (i; 64, or 32 for boxes of 32^3) x TJ (j) tile of one box and marches in +k:
//   * planes k-1, k, k+1 of x, beta_i, beta_j and faces k, k+1 of beta_k live in LDS with a 2-cell halo (ring buffers;
//     11 tiles of (TJ+4) x 68 doubles = 72 KB for TJ = 8, two workgroups per CU);
//   * every value enters LDS ONCE per workgroup: a lane brings its own column (the x value from a register it loaded three
//     steps earlier -- x[k-2..k+3] of its own column slide through registers, which also serve the +-2 k neighbours) and at
//     most one halo cell per array; all global loads are issued one step before they are stored (their latency overlaps the
//     arithmetic of the current plane);
//   * the update reads its 50 neighbours from LDS (conflict free: a wave reads 64 consecutive doubles).
// Arithmetic: the expression tree of the reference macro, term by term (T * (6 face terms) + (0.25 T) * (12 mixed terms),
// every group summed left to right, a mixed term = (beta+ - beta-) * (((x1 - x2) - x3) + x4)), so results are bit-identical
// to the direct kernel and to the reference (tests/test_gpu_operators.py, tests/test_gpu_fcycle_parity.py).
// Ghost cells (depth 2) and the extrapolated coefficient ghosts are read exactly where the reference reads them; the caller has
// run exchange_boundary + apply_BCs_v4 (NO_CORNERS) before, as smooth()/residual() of operators.fv4.c do -- or, with P.ghost_free
// (all boxes local), only apply_BCs_v4: x outside the box is then read from the neighbouring box itself (common.hpp gf_column).
#pragma once
#include "common.hpp"
#include "fv4_math.hpp"

namespace hpgmg {

#ifndef FV4_TWELFTH
#define FV4_TWELFTH ( 0.0833333333333333333)
#endif

struct Fv4TileArgs {
  int xn_id, xout_id, rhs_id;
  double a, b, h2inv, c1, c2;
  int sweep, copy_other_colour, ghost_free;
  TileFused fused;                      // FV4_RESIDUAL only: what becomes of the residual (common.hpp)
  int tiles_i, tiles_j, chunks_k, kchunk, per_xcd, total_blocks;
  // the pre-pass of the one-pass red + black kernel (fv4_rb.hpp) runs a half sweep on a few planes only: chunk ck starts at plane
  // k_origin + ck * k_step (0 / 0: at ck * kchunk); wall_only: chunk 0 / 1 only in boxes whose low / high k face is the domain boundary;
  // x_base / out_base: box-base tables of the vectors xn_id / xout_id when they are plugin-private scratch vectors (NULL: level vectors)
  int k_origin, k_step, wall_only;
  double *const *x_base, *const *out_base;
  const int *order;                     // dispatch slot -> tile (nullptr: identity): two-part launches across rank boundaries (common.hpp tile_part_order)
};
enum { FV4_CHEBY = 0, FV4_GSRB = 1, FV4_JACOBI = 2, FV4_RESIDUAL = 3, FV4_APPLY = 4 };

// the kernel's body; `block` = blockIdx.x (a device function so that the pre-pass of fv4_rb.hpp can run other work in the same launch)
// GATHER: every operand of the bracket read up front with ds_read_b64 (fv4_math.hpp); false: named where the expression uses them (fewer registers)
template <int V, int MODE, int TJ, int TI = 64, bool GATHER = (MODE != FV4_RESIDUAL)>
__device__ __forceinline__ void fv4_tile_body(const hpgmg_hip_level &L, const Fv4TileArgs &P, int block) {
  constexpr int W = TI + 4, H = TJ + 4, NT = TI * TJ, PLANE = W * H;
  constexpr int NH = 4 * W + 4 * TJ;                           // halo cells of one plane tile (two rows above and below, two columns left and right)
  static_assert(NH <= NT, "one halo cell per lane at most");
  constexpr bool kHelm = (V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  constexpr bool kSmooth = (MODE == FV4_CHEBY || MODE == FV4_GSRB || MODE == FV4_JACOBI);
  extern __shared__ double fv4_lds[];
  __shared__ double sR[(MODE == FV4_RESIDUAL) ? 2 * TJ * TI : 1];   // fused residual forms: a plane of residuals / the workgroup's partial maxima
  double *sX = fv4_lds, *sBI = fv4_lds + 3 * PLANE, *sBJ = fv4_lds + 6 * PLANE, *sBK = fv4_lds + 9 * PLANE;   // rings of 3, 3, 3, 2 plane tiles

  int logical = xcd_logical_block(block, P.per_xcd);
  if (P.order) logical = P.order[logical];
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;
  const int li = (int)threadIdx.x, lj = (int)threadIdx.y, tid = lj * TI + li;
  const int i0 = ti * TI, j0 = tj * TJ, i = i0 + li, j = j0 + lj;
  const int k0 = P.k_origin + ck * (P.k_step ? P.k_step : P.kchunk), k1 = (k0 + P.kchunk < L.dim) ? k0 + P.kchunk : L.dim;
  const int jS = L.jStride, kS = L.kStride;
  if (P.wall_only && L.box_nbr[6 * box + (ck ? 5 : 4)] != -1) return;
  hpgmg_hip_level Lx = L, Lo = L;                                  // where the iterate is read / written (a scratch table, or the level)
  if (P.x_base) Lx.box_base = P.x_base;
  if (P.out_base) Lo.box_base = P.out_base;

  // device-memory pointers (common.hpp): global_load / global_store, so a wait for LDS data does not also wait for the loads in flight
  gcptr x = gvec_origin(Lx, box, P.xn_id);
  gptr out = gvec_origin(Lo, box, P.xout_id);
  gcptr rhs = (MODE == FV4_APPLY) ? nullptr : gvec_origin(L, box, P.rhs_id);
  gcptr dinv = kSmooth ? gvec_origin(L, box, VECTOR_DINV) : nullptr;
  gcptr alpha = kHelm ? gvec_origin(L, box, VECTOR_ALPHA) : nullptr;
  gcptr gbi = gvec_origin(L, box, VECTOR_BETA_I);
  gcptr gbj = gvec_origin(L, box, VECTOR_BETA_J);
  gcptr gbk = gvec_origin(L, box, VECTOR_BETA_K);
  int colour000 = 0;
  if (MODE == FV4_GSRB) colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;

  // this lane's own cell and (at most one) halo cell: offsets inside a plane of the box and inside a plane tile
  const int own_g = i + j * jS, own_s = (lj + 2) * W + (li + 2);
  int halo_g = 0, halo_s = 0;
  GfColumn hcol = {box, 0};
  const bool has_halo = tid < NH;
  // which coefficient arrays the stencil reads at this lane's halo cell (operators.fv4.c:87-108): none on the outer ring (only x reaches two
  // cells out); beta_i on rows -1 .. TJ of columns 0 .. TI, beta_j on rows 0 .. TJ of columns -1 .. TI, beta_k on the whole inner ring.
  // The rows never read are not fetched: 2 (beta_i, beta_k) or 3 (beta_j) of the 12 rows a tile of 8 would otherwise stream per plane.
  bool nb_i = false, nb_j = false, nb_k = false;
  if (has_halo) {
    int hi, hj;
    if (tid < 2 * W)      { hj = -2 + tid / W; hi = -2 + tid % W; }
    else if (tid < 4 * W) { const int h = tid - 2 * W; hj = TJ + h / W; hi = -2 + h % W; }
    else                  { const int h = tid - 4 * W, c = h % 4; hj = h / 4; hi = (c < 2) ? c - 2 : TI + (c - 2); }
    halo_g = (i0 + hi) + (j0 + hj) * jS;
    halo_s = (hj + 2) * W + (hi + 2);
    nb_k = hi >= -1 && hi <= TI && hj >= -1 && hj <= TJ; nb_i = nb_k && hi >= 0; nb_j = nb_k && hj >= 0;
    if (P.ghost_free) hcol = gf_column(L, box, i0 + hi, j0 + hj);
  }
  // x of the own column / the halo column on plane p (any p the stencil reaches): inside the box's k range a plain load
  gcptr xh = (has_halo && P.ghost_free) ? gvec_origin(Lx, hcol.box, P.xn_id) + hcol.off : x + halo_g;
  const bool gf = P.ghost_free != 0;
  const int dim = L.dim;
  // planes below the box (p < 0) are only met in the prologue of the first chunk: looked up there.  Planes above it (p >= dim) are met
  // in the last steps of the last chunk: one alternative base pointer per column, selected by p, keeps the marching loop free of branches
  gcptr xo_hi = x + own_g, xh_hi = xh;
  if (gf && k1 == dim) {
    const int n = L.box_nbr[6 * box + 5];
    if (n >= 0) xo_hi = gvec_origin(Lx, n, P.xn_id) + own_g - (long long)dim * kS;
    if (has_halo) { const int m = L.box_nbr[6 * hcol.box + 5]; if (m >= 0) xh_hi = gvec_origin(Lx, m, P.xn_id) + hcol.off - (long long)dim * kS; }
  }
  auto x_own = [&](int p) -> double {
    if (gf && p < 0) return gf_load_outside(Lx, P.xn_id, GfColumn{box, own_g}, p);
    return ((p >= dim) ? xo_hi : x + own_g)[p * kS];
  };
  auto x_halo = [&](int p) -> double {
    if (gf && p < 0) return gf_load_outside(Lx, P.xn_id, hcol, p);
    return ((p >= dim) ? xh_hi : xh)[p * kS];
  };
  auto x_own_fwd = [&](int p) -> double { return ((p >= dim) ? xo_hi : x + own_g)[p * kS]; };     // p >= 0: the marching loop
  auto x_halo_fwd = [&](int p) -> double { return ((p >= dim) ? xh_hi : xh)[p * kS]; };
  auto slot3 = [](int p) { return ((p % 3) + 3) % 3; };

  // ---- prologue: planes k0-1 and k0 of x / beta_i / beta_j and face k0 of beta_k into LDS; x[k0-2 .. k0+2] of the own column into registers
  double xm2 = x_own(k0 - 2), xm1 = x_own(k0 - 1), xc = x[own_g + k0 * kS];
  double xp1 = x_own(k0 + 1), xp2 = x_own(k0 + 2), xp3 = 0.0;
  for (int p = k0 - 1; p <= k0; p++) {
    const int s = slot3(p) * PLANE, pg = p * kS;
    sX[s + own_s] = (p == k0) ? xc : xm1;
    sBI[s + own_s] = gbi[own_g + pg];
    sBJ[s + own_s] = gbj[own_g + pg];
    if (has_halo) { sX[s + halo_s] = x_halo(p); sBI[s + halo_s] = nb_i ? gbi[halo_g + pg] : 0.0; sBJ[s + halo_s] = nb_j ? gbj[halo_g + pg] : 0.0; }
  }
  sBK[(k0 & 1) * PLANE + own_s] = gbk[own_g + k0 * kS];
  if (has_halo) sBK[(k0 & 1) * PLANE + halo_s] = nb_k ? gbk[halo_g + k0 * kS] : 0.0;
  // values in flight: plane k+1 (stored to LDS at the start of step k) and the per-cell streams of plane k
  double n_bi = gbi[own_g + (k0 + 1) * kS], n_bj = gbj[own_g + (k0 + 1) * kS], n_bk = gbk[own_g + (k0 + 1) * kS];
  double h_x = 0, h_bi = 0, h_bj = 0, h_bk = 0;
  if (has_halo) { const int pg = (k0 + 1) * kS; h_x = x_halo(k0 + 1); if (nb_i) h_bi = gbi[halo_g + pg]; if (nb_j) h_bj = gbj[halo_g + pg]; if (nb_k) h_bk = gbk[halo_g + pg]; }
  double c_rhs = (MODE == FV4_APPLY) ? 0.0 : rhs[own_g + k0 * kS], c_dinv = kSmooth ? dinv[own_g + k0 * kS] : 0.0;
  double c_al = kHelm ? alpha[own_g + k0 * kS] : 0.0, c_old = (MODE == FV4_CHEBY) ? out[own_g + k0 * kS] : 0.0;

  const TileFused &F = P.fused;
  TileFusedState<TI, TJ> fs;
  gptr coarse = nullptr;
  if (MODE == FV4_RESIDUAL && F.kind == 2) {
    const int *mp = F.map + 4 * box;
    coarse = gvec_origin(F.Lc, mp[0], F.coarse_id) + (mp[1] + (i >> 1)) + (mp[2] + (j >> 1)) * F.Lc.jStride + (mp[3] + (k0 >> 1)) * F.Lc.kStride;
  }

  for (int k = k0; k < k1; k++) {
    const int pg = k * kS;
    __syncthreads();                                            // every wave is done reading the slots that plane k+1 overwrites
    if (MODE == FV4_RESIDUAL && F.kind == 2 && k > k0) fs.gather(F, sR, li, lj, k - 1, k0, coarse);
    { // plane k+1 (loaded during the previous step) -> LDS
      const int s = slot3(k + 1) * PLANE, sk = ((k + 1) & 1) * PLANE;
      sX[s + own_s] = xp1; sBI[s + own_s] = n_bi; sBJ[s + own_s] = n_bj; sBK[sk + own_s] = n_bk;
      if (has_halo) { sX[s + halo_s] = h_x; sBI[s + halo_s] = h_bi; sBJ[s + halo_s] = h_bj; sBK[sk + halo_s] = h_bk; }
    }
    // issue the loads of the next step: plane k+2, own x three planes ahead, the per-cell streams of plane k+1
    double nn_rhs = 0, nn_dinv = 0, nn_al = 0, nn_old = 0;
    if (k + 1 < k1) {
      const int ng = (k + 2) * kS;
      n_bi = gbi[own_g + ng]; n_bj = gbj[own_g + ng]; n_bk = gbk[own_g + ng];
      if (has_halo) { h_x = x_halo_fwd(k + 2); if (nb_i) h_bi = gbi[halo_g + ng]; if (nb_j) h_bj = gbj[halo_g + ng]; if (nb_k) h_bk = gbk[halo_g + ng]; }
      xp3 = x_own_fwd(k + 3);
      const int cg = own_g + (k + 1) * kS;
      if (MODE != FV4_APPLY) nn_rhs = rhs[cg];
      if (kSmooth) nn_dinv = dinv[cg];
      if (kHelm) nn_al = alpha[cg];
      if (MODE == FV4_CHEBY) nn_old = out[cg];
    }
    __syncthreads();

    bool update = true;
    if (MODE == FV4_GSRB) update = (((i ^ j ^ k ^ colour000) & 1) == 0);
    if (update) {
      double sum;
      if constexpr (GATHER) {
        using namespace fv4rb;
        ldsr X0 = (ldsr)(sX + slot3(k) * PLANE + own_s), Xm = (ldsr)(sX + slot3(k - 1) * PLANE + own_s), Xp = (ldsr)(sX + slot3(k + 1) * PLANE + own_s);
        ldsr I0 = (ldsr)(sBI + slot3(k) * PLANE + own_s), Im = (ldsr)(sBI + slot3(k - 1) * PLANE + own_s), Ip = (ldsr)(sBI + slot3(k + 1) * PLANE + own_s);
        ldsr J0 = (ldsr)(sBJ + slot3(k) * PLANE + own_s), Jm = (ldsr)(sBJ + slot3(k - 1) * PLANE + own_s), Jp = (ldsr)(sBJ + slot3(k + 1) * PLANE + own_s);
        ldsr K0 = (ldsr)(sBK + (k & 1) * PLANE + own_s), K1 = (ldsr)(sBK + ((k + 1) & 1) * PLANE + own_s);
        // operators.fv4.c:87-108 (fv4_math.hpp): every operand read once, then the reference's expression tree
        X25 x; BG1 g1; BG2 g2; Br18 br;
        x.c = xc; x.im1 = X0[-1]; x.ip1 = X0[1]; x.im2 = X0[-2]; x.ip2 = X0[2];
        x.jm1 = X0[-W]; x.jp1 = X0[W]; x.jm2 = X0[-2 * W]; x.jp2 = X0[2 * W];
        x.km1 = xm1; x.kp1 = xp1; x.km2 = xm2; x.kp2 = xp2;
        x.mm = X0[-1 - W]; x.pm = X0[1 - W]; x.mp = X0[-1 + W]; x.pp = X0[1 + W];
        x.m_im = Xm[-1]; x.m_ip = Xm[1]; x.m_jm = Xm[-W]; x.m_jp = Xm[W];
        x.p_im = Xp[-1]; x.p_ip = Xp[1]; x.p_jm = Xp[-W]; x.p_jp = Xp[W];
        beta_g1<W>(g1, I0, Im, Ip, J0, Jm, Jp, K0, K1);
        fv4_brackets(br, x);
        beta_g2<W>(g2, I0, Im, Ip, J0, Jm, Jp, K0, K1);
        double s1, s2;
        fv4_combine_a(s1, s2, br, g1);
        sum = fv4_combine_b(s1, s2, br, g2);
      } else {
        // the residual forms carry the restriction / norm state as well, the pre-pass of the one-pass kernel the special-cell branch: with every operand
        // read up front they need 170 / 132 registers (fewer waves per CU: measured 17 % / 27 % slower), so here the operands are named where the
        // expression uses them, as in the first version (the same tree)
        const double *X0 = sX + slot3(k) * PLANE + own_s, *Xm = sX + slot3(k - 1) * PLANE + own_s, *Xp = sX + slot3(k + 1) * PLANE + own_s;
        const double *I0 = sBI + slot3(k) * PLANE + own_s, *Im = sBI + slot3(k - 1) * PLANE + own_s, *Ip = sBI + slot3(k + 1) * PLANE + own_s;
        const double *J0 = sBJ + slot3(k) * PLANE + own_s, *Jm = sBJ + slot3(k - 1) * PLANE + own_s, *Jp = sBJ + slot3(k + 1) * PLANE + own_s;
        const double *K0 = sBK + (k & 1) * PLANE + own_s, *K1 = sBK + ((k + 1) & 1) * PLANE + own_s;
        // operators.fv4.c:87-93: six face terms
        double s1 = I0[0] * (15.0 * (X0[-1] - xc) - (X0[-2] - X0[1]));
        s1 = s1 + I0[1] * (15.0 * (X0[1] - xc) - (X0[2] - X0[-1]));
        s1 = s1 + J0[0] * (15.0 * (X0[-W] - xc) - (X0[-2 * W] - X0[W]));
        s1 = s1 + J0[W] * (15.0 * (X0[W] - xc) - (X0[2 * W] - X0[-W]));
        s1 = s1 + K0[0] * (15.0 * (xm1 - xc) - (xm2 - xp1));
        s1 = s1 + K1[0] * (15.0 * (xp1 - xc) - (xp2 - xm1));
        // operators.fv4.c:95-108: twelve mixed terms, (beta+ - beta-) * (((x1 - x2) - x3) + x4)
        double s2 = (I0[W] - I0[-W]) * (X0[-1 + W] - X0[W] - X0[-1 - W] + X0[-W]);
        s2 = s2 + (Ip[0] - Im[0]) * (Xp[-1] - xp1 - Xm[-1] + xm1);
        s2 = s2 + (J0[1] - J0[-1]) * (X0[-W + 1] - X0[1] - X0[-W - 1] + X0[-1]);
        s2 = s2 + (Jp[0] - Jm[0]) * (Xp[-W] - xp1 - Xm[-W] + xm1);
        s2 = s2 + (K0[1] - K0[-1]) * (Xm[1] - X0[1] - Xm[-1] + X0[-1]);
        s2 = s2 + (K0[W] - K0[-W]) * (Xm[W] - X0[W] - Xm[-W] + X0[-W]);
        s2 = s2 + (I0[1 + W] - I0[1 - W]) * (X0[1 + W] - X0[W] - X0[1 - W] + X0[-W]);
        s2 = s2 + (Ip[1] - Im[1]) * (Xp[1] - xp1 - Xm[1] + xm1);
        s2 = s2 + (J0[W + 1] - J0[W - 1]) * (X0[W + 1] - X0[1] - X0[W - 1] + X0[-1]);
        s2 = s2 + (Jp[W] - Jm[W]) * (Xp[W] - xp1 - Xm[W] + xm1);
        s2 = s2 + (K1[1] - K1[-1]) * (Xp[1] - X0[1] - Xp[-1] + X0[-1]);
        s2 = s2 + (K1[W] - K1[-W]) * (Xp[W] - X0[W] - Xp[-W] + X0[-W]);
        sum = FV4_TWELFTH * s1 + (0.25 * FV4_TWELFTH) * s2;
      }
      const double Ax = kHelm ? (P.a * c_al) * xc - (P.b * P.h2inv) * sum : ((-P.b) * P.h2inv) * sum;
      double o;
      if (MODE == FV4_CHEBY)         o = xc + P.c1 * (xc - c_old) + P.c2 * c_dinv * (c_rhs - Ax);
      else if (MODE == FV4_GSRB)     o = xc + c_dinv * (c_rhs - Ax);
      else if (MODE == FV4_JACOBI)   o = xc + P.c2 * c_dinv * (c_rhs - Ax);
      else if (MODE == FV4_RESIDUAL) o = c_rhs - Ax;
      else                           o = Ax;
      if (MODE == FV4_RESIDUAL && F.kind == 1) { const double f = fabs(o); fs.lane_max = (f > fs.lane_max) ? f : fs.lane_max; }
      else if (MODE == FV4_RESIDUAL && F.kind == 2) sR[((k & 1) * TJ + lj) * TI + li] = o;
      else out[own_g + pg] = o;
    } else if (P.copy_other_colour) {
      out[own_g + pg] = xc;                                     // out-of-place GSRB copies the other colour (gsrb.c:94-98)
    }
    xm2 = xm1; xm1 = xc; xc = xp1; xp1 = xp2; xp2 = xp3;
    c_rhs = nn_rhs; c_dinv = nn_dinv; c_al = nn_al; c_old = nn_old;
  }
  if (MODE == FV4_RESIDUAL && F.kind == 2) { __syncthreads(); fs.gather(F, sR, li, lj, k1 - 1, k0, coarse); }
  if (MODE == FV4_RESIDUAL && F.kind == 1) {                      // a maximum is exact under any order (misc.c:307-317)
    double mx = fs.lane_max;
    for (int off = 32; off > 0; off >>= 1) { const double o2 = __shfl_down(mx, off, 64); mx = (o2 > mx) ? o2 : mx; }
    __syncthreads();
    if ((tid & 63) == 0) sR[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) { double m2 = sR[0]; for (int q = 1; q < NT / 64; q++) m2 = (sR[q] > m2) ? sR[q] : m2; F.partials[logical] = m2; }
  }
}
template <int V, int MODE, int TJ, int TI = 64>
__global__ __launch_bounds__(TI * TJ) void fv4_tile_kernel(const hpgmg_hip_level L, const Fv4TileArgs P) {
  fv4_tile_body<V, MODE, TJ, TI>(L, P, (int)blockIdx.x);
}

}  // namespace hpgmg

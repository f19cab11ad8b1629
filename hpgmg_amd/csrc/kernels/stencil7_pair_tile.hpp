// stencil7_pair_tile.hpp -- TWO Chebyshev sweeps of the 7-point operator in one launch for cache-resident levels (boxes of 64^3: the
// 128^3 level of config 2), in the tile form of stencil7_tile.hpp.
//
// cheby_pair.hpp does this for the bandwidth-bound levels with 16-wave workgroups that own whole 128-cell rows; on a 128^3 level those
// are too few and too long (latency bound: 80 us per pair against 2 x 26 us for single sweeps).  Here a 64 x TJ workgroup owns a
// 64 (i) x TJ (j) tile of a box and marches in +k with two stages per step, like stencil27_rb.hpp:
//   S1(q): x1 = x0 + c1a (x0 - xm1) + c2a Dinv (rhs - A x0) on plane q of the tile EXTENDED by one cell in i and j (the cells of
//          neighbouring tiles / boxes are recomputed, not exchanged), from planes q-1, q, q+1 of x0 in LDS; x1 goes to a second LDS ring;
//   S2(q-1): x2 = x1 + c1b (x1 - x0) + c2b Dinv (rhs - A x1) on plane q-1 of the tile proper from planes q-2, q-1, q of that ring.
// The Dirichlet boundary is the in-register rule ghost = -centre (apply_BCs_p1), applied to x0 in S1 and to x1 in S2 as the reference
// applies it between sweeps; cells outside the box but inside the domain are read from the box that owns them (every box local).
// Arithmetic per cell = apply_op_7pt + chebyshev.c:86-95, so x1 and x2 are bit-identical to two single sweeps.
#pragma once
#include "common.hpp"
#include "stencil_math.hpp"
#include "cheby_pair.hpp"      // VecRef

namespace hpgmg {

struct S7PairTileArgs {
  VecRef x0, xm1, out1, out2;
  int rhs_id, keep_x1;
  double a, b, h2inv, c1a, c2a, c1b, c2b;
  double *const *scr_base;            // per box: base of the two plugin-private vectors
  int tiles_i, tiles_j, chunks_k, kchunk, per_xcd, total_blocks;
};

template <int V> struct S7Coef { double bi0, bi1, bj0, bj1, bk0, bk1, al, dinv, rhs; };

// TI x TJ tiles: 64 x 8 for boxes of side 64 m; 32 x 8 / 16 x 16 for the levels of boxes of 32^3 / 16^3 (round 3: there a sweep is a launch of
// ~5 us that is mostly launch, and two sweeps per launch halve the count)
template <int V, int TJ, int TI = 64>
__global__ __launch_bounds__(TI * TJ) __attribute__((amdgpu_waves_per_eu(4, 8))) void stencil7_pair_tile_kernel(const hpgmg_hip_level L, const S7PairTileArgs P) {
  constexpr int NT = TI * TJ;
  constexpr int WO = TI + 4, HO = TJ + 4, PO = WO * HO;          // planes of x0: two-cell halo (the corners are never read: star stencil)
  constexpr int WP = TI + 2, HP = TJ + 2, PP = WP * HP;          // planes of x1: one-cell halo
  constexpr int NHO = 4 * WO + 4 * TJ;
  constexpr int NE = 2 * WP + 2 * TJ;
  static_assert(NHO <= NT && NE <= NT, "one extra cell per lane at most");
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  __shared__ double sO[3 * PO];
  __shared__ double sP[3 * PP];

  const int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;
  const int li = (int)threadIdx.x, lj = (int)threadIdx.y, tid = lj * TI + li;
  const int i0 = ti * TI, j0 = tj * TJ;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride;
  const long long vol = L.volume;
  const int k0 = ck * P.kchunk, k1 = (k0 + P.kchunk < dim) ? k0 + P.kchunk : dim;
  const bool read_xm1 = (P.c1a != 0.0);                           // the first Chebyshev sweep of a smooth() has c1 = 0: x_{n-1} is not read (cheby_pair.hpp)

  const int *nb = L.box_nbr + 6 * box;
  const bool wall_ilo = nb[0] == -1, wall_ihi = nb[1] == -1, wall_jlo = nb[2] == -1, wall_jhi = nb[3] == -1, wall_klo = nb[4] == -1, wall_khi = nb[5] == -1;
  auto out_lo = [&](int c, bool wall) { return c < 0 && wall; };
  auto out_hi = [&](int c, bool wall) { return c >= dim && wall; };
  const size_t first = (size_t)L.ghosts * (size_t)(1 + jS + kS);
  // base pointers of a column (all planes, all vectors) of box b: level vectors / scratch vectors
  auto lvl_col = [&](int b, int off) -> const double * { return L.box_base[b] + first + off; };
  auto scr_col = [&](int b, int off) -> const double * { return P.scr_base[b] + first + off; };
  auto vec_col = [&](VecRef r, int b, int off) -> const double * { return (r.scratch ? scr_col(b, off) : lvl_col(b, off)) + (long long)r.id * vol; };
  // plane p of a column given by its in-box pointer (already offset to the vector) and its GfColumn; p may lie outside the box
  auto col_at = [&](const double *inbox, VecRef r, GfColumn c, int p) -> double {
    if (p >= 0 && p < dim) return inbox[p * kS];
    int b = c.box;
    if (p < 0) { const int n = L.box_nbr[6 * b + 4]; if (n >= 0) { b = n; p += dim; } }
    else       { const int n = L.box_nbr[6 * b + 5]; if (n >= 0) { b = n; p -= dim; } }
    return vec_col(r, b, c.off)[p * kS];
  };

  // ---- this lane's cells: (1) own cell; (2) at most one halo cell of the x0 planes; (3) at most one ring cell of the x1 planes
  const int gi = i0 + li, gj = j0 + lj, own_g = gi + gj * jS;
  const int ownO = (lj + 2) * WO + (li + 2), ownP = (lj + 1) * WP + (li + 1);
  const GfColumn ocol = {box, own_g};
  const bool has_h = tid < NHO;
  int hO = 0; bool h_ok = false; GfColumn hcol = {box, 0};
  if (has_h) {
    int hi, hj;
    if (tid < 2 * WO)      { hj = -2 + tid / WO; hi = -2 + tid % WO; }
    else if (tid < 4 * WO) { const int h = tid - 2 * WO; hj = TJ + h / WO; hi = -2 + h % WO; }
    else                   { const int h = tid - 4 * WO, c = h % 4; hj = h / 4; hi = (c < 2) ? c - 2 : TI + (c - 2); }
    const int hgi = i0 + hi, hgj = j0 + hj;
    hO = (hj + 2) * WO + (hi + 2);
    // outside the domain nothing is read (the Dirichlet rule needs only the centre)
    h_ok = !out_lo(hgi, wall_ilo) && !out_hi(hgi, wall_ihi) && !out_lo(hgj, wall_jlo) && !out_hi(hgj, wall_jhi);
    if (h_ok) hcol = gf_column(L, box, hgi, hgj);
  }
  const bool has_e = tid < NE;
  int eP = 0, eO = 0, egi = 0, egj = 0; bool e_in = false; GfColumn ecol = {box, 0};
  if (has_e) {
    int ei, ej;
    if (tid < WP)          { ej = -1; ei = -1 + tid; }
    else if (tid < 2 * WP) { ej = TJ; ei = -1 + (tid - WP); }
    else                   { const int h = tid - 2 * WP; ej = h >> 1; ei = (h & 1) ? TI : -1; }
    egi = i0 + ei; egj = j0 + ej; eP = (ej + 1) * WP + (ei + 1); eO = (ej + 2) * WO + (ei + 2);
    const bool corner = (ei < 0 || ei >= TI) && (ej < 0 || ej >= TJ);      // S2 never reads the corners of the ring
    e_in = !corner && !out_lo(egi, wall_ilo) && !out_hi(egi, wall_ihi) && !out_lo(egj, wall_jlo) && !out_hi(egj, wall_jhi);
    if (e_in) ecol = gf_column(L, box, egi, egj);
  }

  auto slot3 = [](int p) { return ((p % 3) + 3) % 3; };
  auto plane_in = [&](int p) { return !(out_lo(p, wall_klo) || out_hi(p, wall_khi)); };
  // in-box column pointers
  const double *__restrict__ x0_o = vec_col(P.x0, box, own_g);
  const double *__restrict__ x0_h = h_ok ? vec_col(P.x0, hcol.box, hcol.off) : x0_o;
  const double *__restrict__ lvl_o = lvl_col(box, own_g);
  const double *__restrict__ lvl_e = e_in ? lvl_col(ecol.box, ecol.off) : lvl_o;
  const double *__restrict__ xm1_o = vec_col(P.xm1, box, own_g);
  const double *__restrict__ xm1_e = e_in ? vec_col(P.xm1, ecol.box, ecol.off) : xm1_o;
  auto load_x0_own = [&](int p) -> double { return plane_in(p) ? col_at(x0_o, P.x0, ocol, p) : 0.0; };
  auto load_x0_halo = [&](int p) -> double { return (h_ok && plane_in(p)) ? col_at(x0_h, P.x0, hcol, p) : 0.0; };
  // the coefficients of one cell of a column on plane p (inside the domain); bk0 may be handed over from the plane below
  auto lvl_at = [&](const double *inbox, GfColumn c, int id, int p, int extra) -> double {   // vector id at the column's cell on plane p, plus `extra` elements
    if (p >= 0 && p < dim) return inbox[(long long)id * vol + p * kS + extra];
    int b = c.box;
    if (p < 0) { const int n = L.box_nbr[6 * b + 4]; if (n >= 0) { b = n; p += dim; } }
    else       { const int n = L.box_nbr[6 * b + 5]; if (n >= 0) { b = n; p -= dim; } }
    return lvl_col(b, c.off)[(long long)id * vol + p * kS + extra];
  };
  auto load_coef = [&](const double *inbox, GfColumn c, int p, bool have_bk0, double bk0, S7Coef<V> &q) {
    q.rhs = lvl_at(inbox, c, P.rhs_id, p, 0);
    q.dinv = lvl_at(inbox, c, VECTOR_DINV, p, 0);
    q.al = kHelm ? lvl_at(inbox, c, VECTOR_ALPHA, p, 0) : 0.0;
    if (kVC) {
      q.bi0 = lvl_at(inbox, c, VECTOR_BETA_I, p, 0); q.bi1 = lvl_at(inbox, c, VECTOR_BETA_I, p, 1);
      q.bj0 = lvl_at(inbox, c, VECTOR_BETA_J, p, 0); q.bj1 = lvl_at(inbox, c, VECTOR_BETA_J, p, jS);
      q.bk0 = have_bk0 ? bk0 : lvl_at(inbox, c, VECTOR_BETA_K, p, 0);
      q.bk1 = lvl_at(inbox, c, VECTOR_BETA_K, p, kS);
    } else { q.bi0 = q.bi1 = q.bj0 = q.bj1 = q.bk0 = q.bk1 = 0.0; }
  };
  // one Chebyshev update of a cell at LDS position `pos` of three planes (row stride W), with the Dirichlet rule on the sides named in `wall`
  // (bits: 1 -i, 2 +i, 4 -j, 8 +j, 16 -k, 32 +k)
  auto update = [&](const double *m, const double *c, const double *pp, int pos, int W, int wall, double old, const S7Coef<V> &q, double c1, double c2) -> double {
    const double xc = c[pos];
    const double xim = (wall & 1) ? -xc : c[pos - 1], xip = (wall & 2) ? -xc : c[pos + 1];
    const double xjm = (wall & 4) ? -xc : c[pos - W], xjp = (wall & 8) ? -xc : c[pos + W];
    const double xkm = (wall & 16) ? -xc : m[pos], xkp = (wall & 32) ? -xc : pp[pos];
    const double Ax = apply_op_7pt<V>(xc, xim, xip, xjm, xjp, xkm, xkp, q.bi0, q.bi1, q.bj0, q.bj1, q.bk0, q.bk1, q.al, P.a, P.b, P.h2inv);
    return xc + c1 * (xc - old) + c2 * q.dinv * (q.rhs - Ax);
  };
  // walls next to a cell at box-relative (ci, cj): in-plane bits
  auto walls_ij = [&](int ci, int cj) {
    return (out_lo(ci - 1, wall_ilo) ? 1 : 0) | (out_hi(ci + 1, wall_ihi) ? 2 : 0) | (out_lo(cj - 1, wall_jlo) ? 4 : 0) | (out_hi(cj + 1, wall_jhi) ? 8 : 0);
  };
  auto walls_k = [&](int p) { return (out_lo(p - 1, wall_klo) ? 16 : 0) | (out_hi(p + 1, wall_khi) ? 32 : 0); };
  const int own_w = walls_ij(gi, gj), e_w = walls_ij(egi, egj);

  // ---- the march.  S1 on planes qlo .. qhi (inside the domain), S2 on k0 .. k1-1
  const int qlo = (k0 == 0 && wall_klo) ? 0 : k0 - 1, qhi = (k1 == dim && wall_khi) ? dim - 1 : k1;
  for (int p = qlo - 1; p <= qlo + 1; p++) {
    sO[slot3(p) * PO + ownO] = load_x0_own(p);
    if (has_h) sO[slot3(p) * PO + hO] = load_x0_halo(p);
  }
  S7Coef<V> qc = {}, qp = {}, qe = {};                             // own cell: plane q and plane q-1; ring cell: plane q
  double *__restrict__ o2 = const_cast<double *>(vec_col(P.out2, box, own_g));
  double *__restrict__ o1 = const_cast<double *>(vec_col(P.out1, box, own_g));
  bool have_bk = false, have_bk_e = false;

  for (int q = qlo; q <= qhi; q++) {
    __syncthreads();                                              // [A] planes q-1 .. q+1 of x0 are in LDS; S2(q-2) is done with the slot S1(q) overwrites
    double n_x = 0.0, n_h = 0.0;
    if (q < qhi) { n_x = load_x0_own(q + 2); if (has_h) n_h = load_x0_halo(q + 2); }
    const bool hand_over = (q != 0 && q != dim);                  // the first plane of a box reads its own lower face (as the reference does)
    load_coef(lvl_o, ocol, q, have_bk && hand_over, qp.bk1, qc); have_bk = kVC;
    double xm1v = 0.0, xm1e = 0.0;
    if (read_xm1) { xm1v = col_at(xm1_o, P.xm1, ocol, q); if (e_in) xm1e = col_at(xm1_e, P.xm1, ecol, q); }
    if (e_in) { load_coef(lvl_e, ecol, q, have_bk_e && hand_over, qe.bk1, qe); have_bk_e = kVC; }

    // ---- S1(q): x1 on plane q, own cell and ring cell
    {
      const double *c = sO + slot3(q) * PO, *m = sO + slot3(q - 1) * PO, *pp = sO + slot3(q + 1) * PO;
      const int wk = walls_k(q);
      sP[slot3(q) * PP + ownP] = update(m, c, pp, ownO, WO, own_w | wk, read_xm1 ? xm1v : c[ownO], qc, P.c1a, P.c2a);
      if (e_in) sP[slot3(q) * PP + eP] = update(m, c, pp, eO, WO, e_w | wk, read_xm1 ? xm1e : c[eO], qe, P.c1a, P.c2a);
    }
    __syncthreads();                                              // [B] x1 on plane q is complete

    // ---- S2(r): x2 on plane r of the tile proper; x0 (the older iterate) is still in LDS
    auto second = [&](int r, const S7Coef<V> &qq) {
      const double *c = sP + slot3(r) * PP, *m = sP + slot3(r - 1) * PP, *pp = sP + slot3(r + 1) * PP;
      const double old = sO[slot3(r) * PO + ownO];
      const double x2 = update(m, c, pp, ownP, WP, own_w | walls_k(r), old, qq, P.c1b, P.c2b);
      o2[r * kS] = x2;
      if (P.keep_x1) o1[r * kS] = c[ownP];
    };
    if (q - 1 >= k0 && q - 1 < k1) second(q - 1, qp);
    if (wall_khi && k1 == dim && q == dim - 1) second(q, qc);     // the top plane of the domain: x1 above it is the Dirichlet ghost
    qp = qc;
    if (q < qhi) {                                                // plane q+2 of x0 takes the slot of plane q-1: S1(q) read it before [B]; S2(q-1) read only this lane's own cell of it
      sO[slot3(q + 2) * PO + ownO] = n_x;
      if (has_h) sO[slot3(q + 2) * PO + hO] = n_h;
    }
  }
}

}  // namespace hpgmg

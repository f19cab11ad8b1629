// stencil27_rb.hpp -- the two coloured half sweeps of one out-of-place GSRB sweep of the 27-point operator (reference gsrb.c:24-132
// with operators.27pt.c:48-91) in ONE pass over the level.
//
// The reference runs   exchange + apply_BCs_p2(x);  t = red half sweep of x;  exchange + apply_BCs_p2(t);  x' = black half sweep of t
// -- two passes of 32 B per cell each (x, rhs, Dinv read; the other vector written in full: the colour not swept is copied).  Here a
// workgroup of 64 x TJ/2 lanes owns a 64 (i) x TJ (j) tile of a box -- a lane owns two cells, one above the other, hence always one of
// each colour: every lane has work in both stages, and (the row stride of the LDS planes being 4 mod 16 doubles) the lanes of a wave, which
// then alternate between two rows, still hit distinct banks -- and marches in +k with two stages per step:
//   R(q): the red half sweep on plane q, on the tile EXTENDED by one cell in i and j (cells of neighbouring tiles / boxes are recomputed
//         rather than exchanged), from planes q-1, q, q+1 of x held in LDS with a two-cell halo; the result -- red cells updated, black
//         cells copied, i.e. the reference's intermediate vector t -- goes to a second LDS ring, never to memory;
//   B(q-1): the black half sweep on plane q-1 of the tile proper, from planes q-2, q-1, q of that ring; x' is stored.
// Between the two, the ghost cells of t that lie OUTSIDE THE DOMAIN are formed in LDS by the quadratic extrapolation of apply_BCs_p2
// (boundary_fd.c:93-205: faces 2 terms, edges 4, corners 8, the reference's literals and term order) from t itself -- this is what the
// reference's second exchange + apply_BCs_p2 provides -- including the ghost PLANES below / above the domain (from planes 0, 1 / dim-1,
// dim-2 of t).  Cells outside the box but inside the domain are read from the box that owns them (common.hpp gf_column), so the caller
// runs only apply_BCs_p2 on x before the launch.  Every update is the expression tree of apply_op_27pt and gsrb.c:90-105 (the same code
// as stencil27_tile.hpp), so x' is bit-identical to the two separate half sweeps; the intermediate vector is simply never materialised.
// 16 B per cell and half sweep instead of 32.  x' must not alias x (neighbouring workgroups still read x on their halos).
#pragma once
#include "common.hpp"
#include "stencil27_tile.hpp"   // C27_* weights

namespace hpgmg {

struct S27RbArgs {
  int xn_id, xout_id, rhs_id;
  double a, b, h2inv;
  int sweep;                            // number of the first (even) half sweep: its colour is (i ^ j ^ k ^ sweep) & 1 == 0
  int tiles_i, tiles_j, chunks_k, kchunk, per_xcd, total_blocks;
  const int *order;                     // dispatch slot -> tile (nullptr: identity); partial launches (common.hpp tile_part_order)
};

// A x at the centre of three LDS planes (row stride W): operators.27pt.c:60-91 in apply_op_27pt's order -- 8 corners, 12 edges, 6 faces, centre
// (LDS reads through lds27r, stencil27_tile.hpp: one ds_read_b64 each)
template <int W>
__device__ __forceinline__ double apply27_lds(lds27r m, lds27r c, lds27r p, double a, double bh2inv) {
  double s8 = m[-W - 1] + m[-W + 1]; s8 = s8 + m[W - 1]; s8 = s8 + m[W + 1];
  s8 = s8 + p[-W - 1]; s8 = s8 + p[-W + 1]; s8 = s8 + p[W - 1]; s8 = s8 + p[W + 1];
  double s12 = m[-W] + m[-1]; s12 = s12 + m[1]; s12 = s12 + m[W];
  s12 = s12 + c[-W - 1]; s12 = s12 + c[-W + 1]; s12 = s12 + c[W - 1]; s12 = s12 + c[W + 1];
  s12 = s12 + p[-W]; s12 = s12 + p[-1]; s12 = s12 + p[1]; s12 = s12 + p[W];
  double s6 = m[0] + c[-W]; s6 = s6 + c[-1]; s6 = s6 + c[1]; s6 = s6 + c[W]; s6 = s6 + p[0];
  double tt = C27_3 * s8 + C27_2 * s12;
  tt = tt + C27_1 * s6;
  tt = tt + C27_0 * c[0];
  return a * c[0] - bh2inv * tt;
}

template <int TJ>
__global__ __launch_bounds__(32 * TJ) __attribute__((amdgpu_waves_per_eu(4, 8))) void stencil27_rb_kernel(const hpgmg_hip_level L, const S27RbArgs P) {
  constexpr int TI = 64, NT = TI * TJ / 2;                        // lanes: one per PAIR of vertically adjacent cells
  constexpr int WO = TI + 4, HO = TJ + 4, PO = WO * HO;          // planes of x: two-cell halo
  constexpr int WP = TI + 2, HP = TJ + 2, PP = WP * HP;          // planes of t: one-cell halo
  constexpr int NHO = 4 * WO + 4 * TJ;                            // halo cells of an x plane (one per lane at most)
  constexpr int NE = 2 * WP + 2 * TJ;                             // ring cells of a t plane (one per lane at most)
  static_assert(NHO <= NT && NE <= NT && TJ % 2 == 0, "one extra cell per lane at most");
  __shared__ double sO[3 * PO];
  __shared__ double sP[3 * PP];

  int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (P.order) logical = P.order[logical];
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;
  const int li = (int)threadIdx.x, lj = (int)threadIdx.y, tid = lj * TI + li;     // lj: pair of rows 2 lj, 2 lj + 1
  const int i0 = ti * TI, j0 = tj * TJ;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride;
  const int k0 = ck * P.kchunk, k1 = (k0 + P.kchunk < dim) ? k0 + P.kchunk : dim;
  const double bh2inv = P.b * P.h2inv;

  // which sides of the box are the domain boundary (Dirichlet: code -1); inside the domain a neighbouring box is read directly
  const int *nb = L.box_nbr + 6 * box;
  const bool wall_ilo = nb[0] == -1, wall_ihi = nb[1] == -1, wall_jlo = nb[2] == -1, wall_jhi = nb[3] == -1, wall_klo = nb[4] == -1, wall_khi = nb[5] == -1;
  // -1 / +1: box-relative coordinate c lies outside the domain on the low / high side (by how much: dist), 0: inside the domain
  auto side = [&](int c, bool wall_lo, bool wall_hi) { return (c < 0 && wall_lo) ? -1 : ((c >= dim && wall_hi) ? 1 : 0); };
  auto dist = [&](int c) { return c < 0 ? -c : c - dim + 1; };
  const int par0 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;   // parity offset: cell (i,j,k) of the box is red when ((i^j^k^par0)&1) == 0
  auto is_red = [&](int ci, int cj, int ck2) { return (((ci ^ cj ^ ck2 ^ par0) & 1) == 0); };

  // device-memory pointers (common.hpp): global_load / global_store, so a wait for LDS data does not also wait for the loads in flight
  gcptr x = gvec_origin(L, box, P.xn_id);
  gptr out = gvec_origin(L, box, P.xout_id);
  gcptr rhs = gvec_origin(L, box, P.rhs_id);
  gcptr dinv = gvec_origin(L, box, VECTOR_DINV);

  // ---- this lane's cells.  (1) its own two cells (gi, gj) and (gi, gj + 1); (2) at most one halo cell of the x planes; (3) at most one
  // ring cell of the t planes
  const int gi = i0 + li, gj = j0 + 2 * lj, own_g = gi + gj * jS;
  const int ownO = (2 * lj + 2) * WO + (li + 2), ownP = (2 * lj + 1) * WP + (li + 1);     // the upper cell is one row (WO / WP) further
  // (2) halo cell of the x planes
  const bool has_h = tid < NHO;
  int hO = 0, hgi = 0, hgj = 0; bool h_ok = false; GfColumn hcol = {box, 0};
  if (has_h) {
    int hi, hj;
    if (tid < 2 * WO)      { hj = -2 + tid / WO; hi = -2 + tid % WO; }
    else if (tid < 4 * WO) { const int h = tid - 2 * WO; hj = TJ + h / WO; hi = -2 + h % WO; }
    else                   { const int h = tid - 4 * WO, c = h % 4; hj = h / 4; hi = (c < 2) ? c - 2 : TI + (c - 2); }
    hgi = i0 + hi; hgj = j0 + hj; hO = (hj + 2) * WO + (hi + 2);
    // two cells outside the domain nothing is defined (and nothing is needed)
    h_ok = !((side(hgi, wall_ilo, wall_ihi) && dist(hgi) > 1) || (side(hgj, wall_jlo, wall_jhi) && dist(hgj) > 1));
    if (h_ok) hcol = gf_column(L, box, hgi, hgj);
  }
  // (3) ring cell of the t planes: rows -1 and TJ, then columns -1 and TI of rows 0 .. TJ-1
  const bool has_e = tid < NE;
  int eP = 0, eO = 0, egi = 0, egj = 0, e_si = 0, e_sj = 0; GfColumn ecol = {box, 0};
  if (has_e) {
    int ei, ej;
    if (tid < WP)          { ej = -1; ei = -1 + tid; }
    else if (tid < 2 * WP) { ej = TJ; ei = -1 + (tid - WP); }
    else                   { const int h = tid - 2 * WP; ej = h >> 1; ei = (h & 1) ? TI : -1; }
    egi = i0 + ei; egj = j0 + ej; eP = (ej + 1) * WP + (ei + 1); eO = (ej + 2) * WO + (ei + 2);
    e_si = side(egi, wall_ilo, wall_ihi); e_sj = side(egj, wall_jlo, wall_jhi);
    if (!e_si && !e_sj) ecol = gf_column(L, box, egi, egj);
  }
  const bool e_in = has_e && !e_si && !e_sj;                      // ring cell inside the domain (in i and j): R computes it; else it is a ghost of t

  auto slot3 = [](int p) { return ((p % 3) + 3) % 3; };
  // x on plane p (box-relative, -2 <= p <= dim+1) of the own columns / the halo column; 0 where nothing is defined
  auto k_ok = [&](int p) { const int s = side(p, wall_klo, wall_khi); return !(s && dist(p) > 1); };
  gcptr xo = x + own_g;
  gcptr xh = (has_h && h_ok) ? gvec_origin(L, hcol.box, P.xn_id) + hcol.off : xo;
  // own column pair on plane p of vector `id` whose in-box pointer (at the lower cell) is `inbox`
  auto load_own2 = [&](gcptr inbox, int id, int p, double &lo, double &hi) {
    if (p >= 0 && p < dim) { lo = inbox[p * kS]; hi = inbox[p * kS + jS]; }
    else { lo = gf_load_outside(L, id, GfColumn{box, own_g}, p); hi = gf_load_outside(L, id, GfColumn{box, own_g + jS}, p); }
  };
  auto load_x_own2 = [&](int p, double &lo, double &hi) { if (k_ok(p)) load_own2(xo, P.xn_id, p, lo, hi); else { lo = 0.0; hi = 0.0; } };
  auto load_x_halo = [&](int p) -> double {
    if (!(h_ok && k_ok(p))) return 0.0;
    if (p >= 0 && p < dim) return xh[p * kS];
    return gf_load_outside(L, P.xn_id, hcol, p);
  };
  // rhs / Dinv of the ring cell on plane p (inside the domain)
  gcptr rhs_e = e_in ? gvec_origin(L, ecol.box, P.rhs_id) + ecol.off : rhs;
  gcptr dinv_e = e_in ? gvec_origin(L, ecol.box, VECTOR_DINV) + ecol.off : dinv;
  auto load_e = [&](gcptr inbox, int id, int p) -> double {
    if (p >= 0 && p < dim) return inbox[p * kS];
    return gf_load_outside(L, id, ecol, p);
  };

  // apply_BCs_p2 for a ghost cell of t at LDS position `pos` of plane q: (oi, oj, ok) = -1 / 0 / +1 per axis, steps lead back in
  auto t_at = [&](int q, int pos) -> double { return sP[slot3(q) * PP + pos]; };
  auto bc_p2 = [&](int q, int pos, int oi, int oj, int ok) -> double {
    const int di = -oi, dj = -oj * WP, dk = -ok;                   // inward steps: i and j inside a plane, k between planes
    const int kind = (oi != 0) + (oj != 0) + (ok != 0);
    auto g = [&](int a, int b2, int c) { return t_at(q + c * dk, pos + a * di + b2 * dj); };
    double v;
    if (kind == 1) {
      const int a = oi ? 1 : 0, b2 = oj ? 1 : 0, c = ok ? 1 : 0;
      v = -2.0 * g(a, b2, c) + 0.333333333333333333 * g(2 * a, 2 * b2, 2 * c);
    } else if (kind == 2) {
      // the two leaving axes in i < j < k order: (r, s)
      if (!ok)      { v = 4.000000000000000000 * g(1, 1, 0) - 0.666666666666666667 * g(2, 1, 0); v = v - 0.666666666666666667 * g(1, 2, 0); v = v + 0.111111111111111111 * g(2, 2, 0); }
      else if (!oj) { v = 4.000000000000000000 * g(1, 0, 1) - 0.666666666666666667 * g(2, 0, 1); v = v - 0.666666666666666667 * g(1, 0, 2); v = v + 0.111111111111111111 * g(2, 0, 2); }
      else          { v = 4.000000000000000000 * g(0, 1, 1) - 0.666666666666666667 * g(0, 2, 1); v = v - 0.666666666666666667 * g(0, 1, 2); v = v + 0.111111111111111111 * g(0, 2, 2); }
    } else {
      v = -8.000000000000000000 * g(1, 1, 1) + 1.333333333333333333 * g(2, 1, 1);
      v = v + 1.333333333333333333 * g(1, 2, 1);
      v = v + 1.333333333333333333 * g(1, 1, 2);
      v = v - 0.222222222222222222 * g(2, 2, 1);
      v = v - 0.222222222222222222 * g(1, 2, 2);
      v = v - 0.222222222222222222 * g(2, 1, 2);
      v = v + 0.037037037037037037 * g(2, 2, 2);
    }
    return v;
  };

  // ---- the march.  R works on planes qlo .. qhi (those of t that B needs and that lie inside the domain), B on k0 .. k1-1.
  const int qlo = (k0 == 0 && wall_klo) ? 0 : k0 - 1, qhi = (k1 == dim && wall_khi) ? dim - 1 : k1;
  // prologue: planes qlo-1, qlo, qlo+1 of x into LDS, the per-cell streams of plane qlo in flight
  for (int p = qlo - 1; p <= qlo + 1; p++) {
    double lo, hi; load_x_own2(p, lo, hi);
    sO[slot3(p) * PO + ownO] = lo; sO[slot3(p) * PO + ownO + WO] = hi;
    if (has_h) sO[slot3(p) * PO + hO] = load_x_halo(p);
  }
  double b_rhs = 0.0, b_dinv = 0.0;                                // the black cell of the own pair on the plane behind (for B)
  // does the extended tile reach outside the domain in i or j (then t has ghost cells to be formed on every plane)?  Uniform per workgroup.
  const bool wall_ij = (wall_ilo && i0 == 0) || (wall_ihi && i0 + TI == dim) || (wall_jlo && j0 == 0) || (wall_jhi && j0 + TJ == dim);

  // Per step three barriers at most: [A] planes q-1 .. q+1 of x are in LDS and B(q-2) is done with the slot R(q) overwrites; [B] t on plane
  // q is complete inside the domain; [C] (only next to the domain boundary) its ghost cells are formed.
  for (int q = qlo; q <= qhi; q++) {
    __syncthreads();                                              // [A]
    // loads of this step, all consumed at its end or after the 27-point sums: plane q+2 of x (stored into LDS after B, into the slot of plane
    // q-1) and rhs / Dinv of plane q (the red cell needs them at the end of R(q), the black one a step later in B(q))
    double n_lo = 0.0, n_hi = 0.0, n_h = 0.0, c_rhs_lo, c_rhs_hi, c_dinv_lo, c_dinv_hi, e_rhs = 0.0, e_dinv = 0.0;
    if (q < qhi) { load_x_own2(q + 2, n_lo, n_hi); if (has_h) n_h = load_x_halo(q + 2); }
    load_own2(rhs + own_g, P.rhs_id, q, c_rhs_lo, c_rhs_hi); load_own2(dinv + own_g, VECTOR_DINV, q, c_dinv_lo, c_dinv_hi);
    if (e_in) { e_rhs = load_e(rhs_e, P.rhs_id, q); e_dinv = load_e(dinv_e, VECTOR_DINV, q); }

    // ---- R(q): t on plane q -- the red cell of the own pair updated (gsrb.c:90-105), the black one copied -- and the ring cell
    const int up = is_red(gi, gj, q) ? 0 : 1;                     // 0: the lower cell of the pair is the red one, 1: the upper
    {
      lds27r c = (lds27r)(sO + slot3(q) * PO), m = (lds27r)(sO + slot3(q - 1) * PO), pp = (lds27r)(sO + slot3(q + 1) * PO);
      const int oR = ownO + up * WO, oB = ownO + (1 - up) * WO;
      const double r_rhs = up ? c_rhs_hi : c_rhs_lo, r_dinv = up ? c_dinv_hi : c_dinv_lo;
      const double v = c[oR] + r_dinv * (r_rhs - apply27_lds<WO>(m + oR, c + oR, pp + oR, P.a, bh2inv));
      sP[slot3(q) * PP + ownP + up * WP] = v;
      sP[slot3(q) * PP + ownP + (1 - up) * WP] = c[oB];
      if (e_in) {
        double w = c[eO];
        if (is_red(egi, egj, q)) w = w + e_dinv * (e_rhs - apply27_lds<WO>(m + eO, c + eO, pp + eO, P.a, bh2inv));
        sP[slot3(q) * PP + eP] = w;
      }
    }
    __syncthreads();                                              // [B]
    const bool ghost_below = (wall_klo && k0 == 0 && q == 1);
    if (wall_ij || ghost_below) {
      // ---- ghost cells of t on plane q that lie outside the domain in i and / or j: apply_BCs_p2 from t itself
      if (has_e && !e_in) sP[slot3(q) * PP + eP] = bc_p2(q, eP, e_si, e_sj, 0);
      // ---- the ghost PLANE of t below the domain, once planes 0 and 1 exist (its slot is free: plane 2 of t comes later)
      if (ghost_below) {
        sP[slot3(-1) * PP + ownP] = bc_p2(-1, ownP, 0, 0, -1);
        sP[slot3(-1) * PP + ownP + WP] = bc_p2(-1, ownP + WP, 0, 0, -1);
        if (has_e) sP[slot3(-1) * PP + eP] = bc_p2(-1, eP, e_si, e_sj, -1);
      }
      __syncthreads();                                            // [C]
    }

    // ---- B(r): the black half sweep on the tile proper, from planes r-1, r, r+1 of t: the black cell of the pair updated, the red one stored as it is
    auto black = [&](int r, double k_rhs, double k_dinv) {
      lds27r c = (lds27r)(sP + slot3(r) * PP), m = (lds27r)(sP + slot3(r - 1) * PP), pp = (lds27r)(sP + slot3(r + 1) * PP);
      const int upb = is_red(gi, gj, r) ? 1 : 0;                  // 1: the upper cell of the pair is the black one
      const int oB = ownP + upb * WP, oR = ownP + (1 - upb) * WP;
      const double v = c[oB] + k_dinv * (k_rhs - apply27_lds<WP>(m + oB, c + oB, pp + oB, P.a, bh2inv));
      out[own_g + r * kS + upb * jS] = v;
      out[own_g + r * kS + (1 - upb) * jS] = c[oR];
    };
    if (q - 1 >= k0 && q - 1 < k1) black(q - 1, b_rhs, b_dinv);
    // the black cell of plane q is the one R(q) did not use
    const double kb_rhs = up ? c_rhs_lo : c_rhs_hi, kb_dinv = up ? c_dinv_lo : c_dinv_hi;
    // ---- the top of the domain: the ghost plane above it takes the slot of plane dim-3, which B(dim-2) has just read; then B(dim-1)
    if (wall_khi && k1 == dim && q == dim - 1) {
      __syncthreads();
      sP[slot3(dim) * PP + ownP] = bc_p2(dim, ownP, 0, 0, 1);
      sP[slot3(dim) * PP + ownP + WP] = bc_p2(dim, ownP + WP, 0, 0, 1);
      if (has_e) sP[slot3(dim) * PP + eP] = bc_p2(dim, eP, e_si, e_sj, 1);
      __syncthreads();
      black(q, kb_rhs, kb_dinv);
    }
    b_rhs = kb_rhs; b_dinv = kb_dinv;
    if (q < qhi) {                                                // plane q+2 of x takes the slot of plane q-1, which R(q) read before [B]
      sO[slot3(q + 2) * PO + ownO] = n_lo; sO[slot3(q + 2) * PO + ownO + WO] = n_hi;
      if (has_h) sO[slot3(q + 2) * PO + hO] = n_h;
    }
  }
}

}  // namespace hpgmg

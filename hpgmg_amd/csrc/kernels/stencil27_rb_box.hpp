// stencil27_rb_box.hpp -- the two coloured half sweeps of an out-of-place 27-point GSRB sweep in one launch on SMALL levels: one
// workgroup per cube of D^3 cells (D = 2, 4, 8) -- a whole box, or one of the (dim / D)^3 cubes of a box of 16^3 or 32^3 -- held in LDS.
//
// On the levels below 128^3 a coloured half sweep of the 27-point plugin is two launches of 4-5 us each (ghost fill, stencil), a sweep
// four, a smooth() eight, whatever the level size -- a fifth of the `7 64` F-cycle.  Here a workgroup
//   1. loads x on the box extended by two cells (cells of neighbouring boxes from the box that owns them);
//   2. forms the ghost cells of x outside the domain in LDS (apply_BCs_p2, boundary_fd.c:93-205, from x itself: the first ghost fill);
//   3. forms t -- red cells updated, black ones copied -- on the box extended by ONE cell (the neighbours' cells are recomputed);
//   4. forms the ghost cells of t outside the domain (the second ghost fill);
//   5. forms x' = black half sweep of t on the box proper and stores it.
// Same expression trees as stencil27_rb.hpp / the separate launches: bit-identical.  A smooth() of a cycle is then two launches, x -> TEMP -> x.
#pragma once
#include "common.hpp"
#include "stencil27_rb.hpp"     // apply27_lds, C27_*

namespace hpgmg {

struct S27RbBoxArgs {
  int xn_id, xout_id, rhs_id;
  double a, b, h2inv;
  int sweep;                            // the first (even) half sweep
  int cubes;                            // cubes per box side (box side = cubes * D)
};

// apply_BCs_p2 at one ghost cell of an LDS cube with strides (1, W, W*W): (oi, oj, ok) = -1 / 0 / +1 per axis, the value from the cells inside
__device__ __forceinline__ double box_bc_p2(const double *v, int pos, int W, int oi, int oj, int ok) {
  const int di = -oi, dj = -oj * W, dk = -ok * W * W;
  const int kind = (oi != 0) + (oj != 0) + (ok != 0);
  auto g = [&](int a, int b2, int c) { return v[pos + a * di + b2 * dj + c * dk]; };
  double r;
  if (kind == 1) {
    const int a = oi ? 1 : 0, b2 = oj ? 1 : 0, c = ok ? 1 : 0;
    r = -2.0 * g(a, b2, c) + 0.333333333333333333 * g(2 * a, 2 * b2, 2 * c);
  } else if (kind == 2) {
    if (!ok)      { r = 4.000000000000000000 * g(1, 1, 0) - 0.666666666666666667 * g(2, 1, 0); r = r - 0.666666666666666667 * g(1, 2, 0); r = r + 0.111111111111111111 * g(2, 2, 0); }
    else if (!oj) { r = 4.000000000000000000 * g(1, 0, 1) - 0.666666666666666667 * g(2, 0, 1); r = r - 0.666666666666666667 * g(1, 0, 2); r = r + 0.111111111111111111 * g(2, 0, 2); }
    else          { r = 4.000000000000000000 * g(0, 1, 1) - 0.666666666666666667 * g(0, 2, 1); r = r - 0.666666666666666667 * g(0, 1, 2); r = r + 0.111111111111111111 * g(0, 2, 2); }
  } else {
    r = -8.000000000000000000 * g(1, 1, 1) + 1.333333333333333333 * g(2, 1, 1);
    r = r + 1.333333333333333333 * g(1, 2, 1);
    r = r + 1.333333333333333333 * g(1, 1, 2);
    r = r - 0.222222222222222222 * g(2, 2, 1);
    r = r - 0.222222222222222222 * g(1, 2, 2);
    r = r - 0.222222222222222222 * g(2, 1, 2);
    r = r + 0.037037037037037037 * g(2, 2, 2);
  }
  return r;
}

// D = cube side; NT lanes
template <int D, int NT>
__global__ __launch_bounds__(NT) void stencil27_rb_box_kernel(const hpgmg_hip_level L, const S27RbBoxArgs P) {
  constexpr int WX = D + 4, NX = WX * WX * WX;                    // x on the box + 2
  constexpr int WT = D + 2, NTC = WT * WT * WT;                   // t on the box + 1
  extern __shared__ double rb_box_lds[];
  double *sX = rb_box_lds, *sT = rb_box_lds + NX;
  const int nc = P.cubes, cube = (int)blockIdx.x % (nc * nc * nc), box = (int)blockIdx.x / (nc * nc * nc), tid = (int)threadIdx.x;
  const int oi = (cube % nc) * D, oj = ((cube / nc) % nc) * D, ok = (cube / (nc * nc)) * D;      // the cube's origin inside its box
  const int dim = L.dim, jS = L.jStride, kS = L.kStride;
  const double bh2inv = P.b * P.h2inv;
  const int *nb = L.box_nbr + 6 * box;
  const bool wall[6] = {nb[0] == -1, nb[1] == -1, nb[2] == -1, nb[3] == -1, nb[4] == -1, nb[5] == -1};
  // -1 / +1: box-relative coordinate c lies outside the domain (by dist(c) cells), 0: inside
  auto side = [&](int c, int ax) { return (c < 0 && wall[2 * ax]) ? -1 : ((c >= dim && wall[2 * ax + 1]) ? 1 : 0); };
  auto dist = [&](int c) { return c < 0 ? -c : c - dim + 1; };
  const int par0 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;
  auto is_red = [&](int ci, int cj, int ck) { return (((ci ^ cj ^ ck ^ par0) & 1) == 0); };
  // vector `id` at the cell (ci, cj, ck), box-relative, somewhere inside the domain (read from the box that owns it)
  auto at = [&](int id, int ci, int cj, int ck) -> double {
    if ((unsigned)ci < (unsigned)dim && (unsigned)cj < (unsigned)dim && (unsigned)ck < (unsigned)dim) return vec_origin(L, box, id)[ci + cj * jS + ck * kS];   // the box itself: no look-ups
    const GfColumn col = gf_column(L, box, ci, cj);
    if (ck >= 0 && ck < dim) return vec_origin(L, col.box, id)[col.off + ck * kS];
    return gf_load_outside(L, id, col, ck);
  };

  // ---- 1. x on the box + 2, inside the domain
  for (int c = tid; c < NX; c += NT) {
    const int i = oi + c % WX - 2, j = oj + (c / WX) % WX - 2, k = ok + c / (WX * WX) - 2;      // box-relative
    const bool inside = !side(i, 0) && !side(j, 1) && !side(k, 2);
    sX[c] = inside ? at(P.xn_id, i, j, k) : 0.0;
  }
  __syncthreads();
  // ---- 2. its ghost cells one cell outside the domain
  for (int c = tid; c < NX; c += NT) {
    const int i = oi + c % WX - 2, j = oj + (c / WX) % WX - 2, k = ok + c / (WX * WX) - 2;
    const int si = side(i, 0), sj = side(j, 1), sk = side(k, 2);
    if ((si || sj || sk) && !(si && dist(i) > 1) && !(sj && dist(j) > 1) && !(sk && dist(k) > 1)) sX[c] = box_bc_p2(sX, c, WX, si, sj, sk);
  }
  __syncthreads();
  // ---- 3. t on the box + 1, inside the domain: the red half sweep (gsrb.c:90-105), black cells copied
  for (int c = tid; c < NTC; c += NT) {
    const int li = c % WT - 1, lj = (c / WT) % WT - 1, lk = c / (WT * WT) - 1, i = oi + li, j = oj + lj, k = ok + lk;
    if (side(i, 0) || side(j, 1) || side(k, 2)) continue;
    const int cx = (li + 2) + (lj + 2) * WX + (lk + 2) * WX * WX;
    double v = sX[cx];
    if (is_red(i, j, k)) {
      const double Ax = apply27_lds<WX>((lds27r)(sX + cx - WX * WX), (lds27r)(sX + cx), (lds27r)(sX + cx + WX * WX), P.a, bh2inv);
      v = v + at(VECTOR_DINV, i, j, k) * (at(P.rhs_id, i, j, k) - Ax);
    }
    sT[c] = v;
  }
  __syncthreads();
  // ---- 4. the ghost cells of t one cell outside the domain
  for (int c = tid; c < NTC; c += NT) {
    const int i = oi + c % WT - 1, j = oj + (c / WT) % WT - 1, k = ok + c / (WT * WT) - 1;
    const int si = side(i, 0), sj = side(j, 1), sk = side(k, 2);
    if (si || sj || sk) sT[c] = box_bc_p2(sT, c, WT, si, sj, sk);
  }
  __syncthreads();
  // ---- 5. x' on the box: the black half sweep of t, red cells as they are
  const double *rhs = vec_origin(L, box, P.rhs_id), *dinv = vec_origin(L, box, VECTOR_DINV);
  double *out = vec_origin(L, box, P.xout_id);
  for (int c = tid; c < D * D * D; c += NT) {
    const int li = c % D, lj = (c / D) % D, lk = c / (D * D), i = oi + li, j = oj + lj, k = ok + lk;
    const int ct = (li + 1) + (lj + 1) * WT + (lk + 1) * WT * WT, g = i + j * jS + k * kS;
    double v = sT[ct];
    if (!is_red(i, j, k)) {
      const double Ax = apply27_lds<WT>((lds27r)(sT + ct - WT * WT), (lds27r)(sT + ct), (lds27r)(sT + ct + WT * WT), P.a, bh2inv);
      v = v + dinv[g] * (rhs[g] - Ax);
    }
    out[g] = v;
  }
}

}  // namespace hpgmg

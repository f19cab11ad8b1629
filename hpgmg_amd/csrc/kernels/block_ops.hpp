// block_ops.hpp -- entry routines of the block-list executors (ghost-exchange copies and the boundary conditions), shared by the
// streaming kernels of blocks.hip (one workgroup per list entry) and the single-workgroup small-level kernels of stencil.hip (one
// wave per entry).  Reference semantics per entry: operators/blockCopy.c:6-156, boundary_fd.c:6-205, boundary_fv.c:101-569.
#pragma once
#include "common.hpp"

namespace hpgmg {

struct Side { double *p; int jS, kS; };

// resolve one side of an entry to a pointer at its (i,j,k) start plus strides
__device__ __forceinline__ Side resolve_read(const hpgmg_hip_level &L, int id, const blockCopy_type &e) {
  Side s;
  if (e.read.box >= 0) { s.jS = L.jStride; s.kS = L.kStride; s.p = vec_origin(L, e.read.box, id); }
  else { s.jS = e.read.jStride; s.kS = e.read.kStride; s.p = e.read.ptr; }
  s.p += e.read.i + e.read.j * s.jS + e.read.k * s.kS;
  return s;
}
__device__ __forceinline__ Side resolve_write(const hpgmg_hip_level &L, int id, const blockCopy_type &e) {
  Side s;
  if (e.write.box >= 0) { s.jS = L.jStride; s.kS = L.kStride; s.p = vec_origin(L, e.write.box, id); }
  else { s.jS = e.write.jStride; s.kS = e.write.kStride; s.p = e.write.ptr; }
  s.p += e.write.i + e.write.j * s.jS + e.write.k * s.kS;
  return s;
}

// Every list executor below is an ENTRY routine: lanes tid, tid + nth, ... of the caller work through entry e.  The streaming kernels
// (blocks.hip) give one workgroup to an entry; the single-workgroup small-level kernels (stencil.hip) give a wave to an entry.
template <bool kIncrement>
__device__ __forceinline__ void copy_entry(const hpgmg_hip_level &L, int id, const blockCopy_type &e, double prescale, int tid, int nth) {
  const Side r = resolve_read(L, id, e), w = resolve_write(L, id, e);
  const int di = e.dim.i, dj = e.dim.j, n = di * dj * e.dim.k;
  for (int t = tid; t < n; t += nth) {
    const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
    const double v = r.p[i + j * r.jS + k * r.kS];
    double *dst = &w.p[i + j * w.jS + k * w.kS];
    if (kIncrement) *dst = prescale * (*dst) + v; else *dst = v;
  }
}

// The *_at forms take the origin of the vector (first interior cell of the box) as a pointer of ANY address space: the streaming kernels pass
// device memory, the single-workgroup kernels that work on an image of a box in LDS pass an LDS pointer (ds_read / ds_write, not FLAT).
template <typename P>
__device__ __forceinline__ void bc_p1_entry_at(P x0, const hpgmg_hip_level &L, const blockCopy_type &e, int tid, int nth) {
  const int inward = 26 - e.subtype;                       // direction pointing back into the domain
  const int ni = inward % 3 - 1, nj = (inward % 9) / 3 - 1, nk = inward / 9 - 1;
  const int kind = (ni != 0) + (nj != 0) + (nk != 0);      // 1 face, 2 edge, 3 corner
  const double scale = (kind == 2) ? 1.0 : -1.0;
  const int jS = L.jStride, kS = L.kStride, step = ni + nj * jS + nk * kS;
  P x = x0 + (e.read.i + e.read.j * jS + e.read.k * kS);
  const int di = e.dim.i, dj = e.dim.j, n = di * dj * e.dim.k;
  for (int t = tid; t < n; t += nth) {
    const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
    const int ijk = i + j * jS + k * kS;
    x[ijk] = scale * x[ijk + step];
  }
}
__device__ __forceinline__ void bc_p1_entry(const hpgmg_hip_level &L, int id, const blockCopy_type &e, int tid, int nth) {
  bc_p1_entry_at(vec_origin(L, e.read.box, id), L, e, tid, nth);
}

// boundary_fd.c:93-205 apply_BCs_p2: quadratic extrapolation through a zero on the boundary face;
// faces 2 terms, edges 4, corners 8 (decimal literals of the reference).  One ghost cell x[ijk]; the steps lead back into the domain.
template <int NN, typename CP, typename WP>
__device__ __forceinline__ void bc_p2_cell(CP x, WP xw, int ijk, int s0, int s1, int s2) {
  double v;
  if (NN == 1) {
    v = -2.0 * x[ijk + s0] + 0.333333333333333333 * x[ijk + 2 * s0];
  } else if (NN == 2) {
    const int dr = s0, ds = s1;
    v = 4.000000000000000000 * x[ijk + dr + ds] - 0.666666666666666667 * x[ijk + 2 * dr + ds];
    v = v - 0.666666666666666667 * x[ijk + dr + 2 * ds];
    v = v + 0.111111111111111111 * x[ijk + 2 * dr + 2 * ds];
  } else {
    const int di = s0, dj = s1, dk = s2;
    v = -8.000000000000000000 * x[ijk + di + dj + dk] + 1.333333333333333333 * x[ijk + 2 * di + dj + dk];
    v = v + 1.333333333333333333 * x[ijk + di + 2 * dj + dk];
    v = v + 1.333333333333333333 * x[ijk + di + dj + 2 * dk];
    v = v - 0.222222222222222222 * x[ijk + 2 * di + 2 * dj + dk];
    v = v - 0.222222222222222222 * x[ijk + di + 2 * dj + 2 * dk];
    v = v - 0.222222222222222222 * x[ijk + 2 * di + dj + 2 * dk];
    v = v + 0.037037037037037037 * x[ijk + 2 * di + 2 * dj + 2 * dk];
  }
  xw[ijk] = v;
}
template <typename P>
__device__ __forceinline__ void bc_p2_entry_at(P x0, const hpgmg_hip_level &L, const blockCopy_type &e, int tid, int nth) {
  const int jS = L.jStride, kS = L.kStride, inward = 26 - e.subtype;
  const int di = (inward % 3 - 1) * 1, dj = ((inward % 9) / 3 - 1) * jS, dk = (inward / 9 - 1) * kS;
  const int kind = (di != 0) + (dj != 0) + (dk != 0);
  P x = x0 + (e.read.i + e.read.j * jS + e.read.k * kS);
  const int ni = e.dim.i, nj = e.dim.j, n = ni * nj * e.dim.k;
  int dr = 0, ds = 0;
  if (di == 0) { dr = dj; ds = dk; }
  if (dj == 0) { dr = di; ds = dk; }
  if (dk == 0) { dr = di; ds = dj; }
  for (int t = tid; t < n; t += nth) {
    const int i = t % ni, j = (t / ni) % nj, k = t / (ni * nj);
    const int ijk = i + j * jS + k * kS;
    if (kind == 1)      bc_p2_cell<1>(x, x, ijk, di + dj + dk, 0, 0);
    else if (kind == 2) bc_p2_cell<2>(x, x, ijk, dr, ds, 0);
    else                bc_p2_cell<3>(x, x, ijk, di, dj, dk);
  }
}
__device__ __forceinline__ void bc_p2_entry(const hpgmg_hip_level &L, int id, const blockCopy_type &e, int tid, int nth) {
  bc_p2_entry_at(vec_origin(L, e.read.box, id), L, e, tid, nth);
}


// ---- finite-volume boundary conditions (reference operators/boundary_fv.c) -------------------
// Geometry of one BC list entry: axes whose DOMAIN normal component is non-zero (in i<j<k order)
// sit at ghost index -1 / dim and step inward; the remaining axes run over the entry's extent.
struct BcGeom { int nn, pos[3], step[3], lo[2], len[2], fstride[2]; };
__device__ __forceinline__ BcGeom bc_geometry(const hpgmg_hip_level &L, const blockCopy_type &e) {
  // (no array is indexed by a value known only at run time: such an index sends the whole descriptor through scratch memory -- a store and
  // a load round trip per entry, which was most of the boundary stage of the single-workgroup kernels)
  BcGeom g;
  const int s0 = 1, s1 = L.jStride, s2 = L.kStride;
  const int d0 = e.subtype % 3 - 1, d1 = (e.subtype % 9) / 3 - 1, d2 = e.subtype / 9 - 1;
  const bool n0 = d0 != 0, n1 = d1 != 0, n2 = d2 != 0;
  const int p0 = (d0 < 0 ? -1 : L.dim) * s0, p1 = (d1 < 0 ? -1 : L.dim) * s1, p2 = (d2 < 0 ? -1 : L.dim) * s2;
  const int t0 = -d0 * s0, t1 = -d1 * s1, t2 = -d2 * s2;
  g.nn = (int)n0 + (int)n1 + (int)n2;
  // the axes that leave the domain, in axis order
  g.pos[0]  = n0 ? p0 : n1 ? p1 : n2 ? p2 : 0;
  g.step[0] = n0 ? t0 : n1 ? t1 : n2 ? t2 : 0;
  g.pos[1]  = n0 ? (n1 ? p1 : n2 ? p2 : 0) : (n1 && n2) ? p2 : 0;
  g.step[1] = n0 ? (n1 ? t1 : n2 ? t2 : 0) : (n1 && n2) ? t2 : 0;
  g.pos[2]  = (n0 && n1 && n2) ? p2 : 0;
  g.step[2] = (n0 && n1 && n2) ? t2 : 0;
  // the (first two) axes that stay inside, in axis order
  const int fa = !n0 ? 0 : !n1 ? 1 : !n2 ? 2 : -1;
  const int fb = !n0 ? (!n1 ? 1 : !n2 ? 2 : -1) : (!n1 && !n2) ? 2 : -1;
  g.lo[0]      = fa == 0 ? e.read.i : fa == 1 ? e.read.j : fa == 2 ? e.read.k : 0;
  g.len[0]     = fa == 0 ? e.dim.i  : fa == 1 ? e.dim.j  : fa == 2 ? e.dim.k  : 1;
  g.fstride[0] = fa == 0 ? s0 : fa == 1 ? s1 : fa == 2 ? s2 : 0;
  g.lo[1]      = fb == 1 ? e.read.j : fb == 2 ? e.read.k : 0;
  g.len[1]     = fb == 1 ? e.dim.j  : fb == 2 ? e.dim.k  : 1;
  g.fstride[1] = fb == 1 ? s1 : fb == 2 ? s2 : 0;
  return g;
}
// first pass of the v2 / v4 conditions when the ghost zone is deeper than the condition fills: clear the whole region.  The caller
// puts a barrier between this pass and the extrapolation pass (bc_v2_entry / bc_v4_entry) -- the two touch the same cells.
template <typename P>
__device__ __forceinline__ void bc_zero_entry_at(P x, const hpgmg_hip_level &L, const blockCopy_type &e, int tid, int nth) {
  const int ni = e.dim.i, nj = e.dim.j, n = ni * nj * e.dim.k;
  for (int t = tid; t < n; t += nth) {
    const int i = t % ni, j = (t / ni) % nj, k = t / (ni * nj);
    x[(i + e.read.i) + (j + e.read.j) * L.jStride + (k + e.read.k) * L.kStride] = 0.0;
  }
}
__device__ __forceinline__ void bc_zero_entry(const hpgmg_hip_level &L, int id, const blockCopy_type &e, int tid, int nth) {
  bc_zero_entry_at(vec_origin(L, e.read.box, id), L, e, tid, nth);
}

// boundary_fv.c:101-250 apply_BCs_v2: first ghost layer by quadratic extrapolation of cell averages, deeper layers zero.
// One ghost cell x[ijk] where NN axes leave the domain; steps lead back into it.
template <int NN, typename CP, typename WP>
__device__ __forceinline__ void bc_v2_cell(CP x, WP xw, int ijk, int s0, int s1, int s2) {
  double v;
  if (NN == 1) {
    v = -2.5 * x[ijk + s0] + 0.5 * x[ijk + 2 * s0];
  } else if (NN == 2) {
    const int ds = s0, dt = s1;
    v = 6.25 * x[ijk + ds + dt] - 1.25 * x[ijk + 2 * ds + dt];
    v = v - 1.25 * x[ijk + ds + 2 * dt];
    v = v + 0.25 * x[ijk + 2 * ds + 2 * dt];
  } else {
    const int di = s0, dj = s1, dk = s2;
    v = -15.625 * x[ijk + di + dj + dk] + 3.125 * x[ijk + 2 * di + dj + dk];
    v = v + 3.125 * x[ijk + di + 2 * dj + dk];
    v = v + 3.125 * x[ijk + di + dj + 2 * dk];
    v = v - 0.625 * x[ijk + 2 * di + 2 * dj + dk];
    v = v - 0.625 * x[ijk + di + 2 * dj + 2 * dk];
    v = v - 0.625 * x[ijk + 2 * di + dj + 2 * dk];
    v = v + 0.125 * x[ijk + 2 * di + 2 * dj + 2 * dk];
  }
  xw[ijk] = v;
}
template <typename P>
__device__ __forceinline__ void bc_v2_entry_at(P x, const hpgmg_hip_level &L, const blockCopy_type &e, int tid, int nth) {
  const BcGeom g = bc_geometry(L, e);
  const int n = g.len[0] * g.len[1];
  for (int t = tid; t < n; t += nth) {
    const int r = t % g.len[0], q = t / g.len[0];
    const int ijk = (r + g.lo[0]) * g.fstride[0] + (q + g.lo[1]) * g.fstride[1];
    if (g.nn == 1)      bc_v2_cell<1>(x, x, ijk + g.pos[0], g.step[0], 0, 0);
    else if (g.nn == 2) bc_v2_cell<2>(x, x, ijk + g.pos[0] + g.pos[1], g.step[0], g.step[1], 0);
    else                bc_v2_cell<3>(x, x, ijk + g.pos[0] + g.pos[1] + g.pos[2], g.step[0], g.step[1], g.step[2]);
  }
}
__device__ __forceinline__ void bc_v2_entry(const hpgmg_hip_level &L, int id, const blockCopy_type &e, int tid, int nth) {
  bc_v2_entry_at(vec_origin(L, e.read.box, id), L, e, tid, nth);
}

// boundary_fv.c:262-569 apply_BCs_v4: near/far ghost from the four cells next to the boundary,
// N = (-77 x1 + 43 x2 - 17 x3 + 3 x4)/12, F = (-505 x1 + 335 x2 - 145 x3 + 27 x4)/12, applied axis after axis
__device__ __forceinline__ double v4_near(double x1, double x2, double x3, double x4) { const double w = 1.0 / 12.0; double s = -77.0 * x1 + 43.0 * x2; s = s - 17.0 * x3; s = s + 3.0 * x4; return w * s; }
__device__ __forceinline__ double v4_far(double x1, double x2, double x3, double x4)  { const double w = 1.0 / 12.0; double s = -505.0 * x1 + 335.0 * x2; s = s - 145.0 * x3; s = s + 27.0 * x4; return w * s; }
// the ghost cells behind x[ijk] (the near one on every leaving axis): 2 for a face, 4 for an edge, 8 for a corner
template <int NN, typename CP, typename WP>
__device__ __forceinline__ void bc_v4_cell(CP x, WP xw, int ijk, int s0, int s1, int s2) {
  if (NN == 1) {
    const int dt = s0;
    const double x1 = x[ijk + dt], x2 = x[ijk + 2 * dt], x3 = x[ijk + 3 * dt], x4 = x[ijk + 4 * dt];
    xw[ijk] = v4_near(x1, x2, x3, x4);
    xw[ijk - dt] = v4_far(x1, x2, x3, x4);
  } else if (NN == 2) {
    const int ds = s0, dt = s1;
    double nr[4], fr[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int o = ijk + (m + 1) * dt;
      const double a1 = x[o + ds], a2 = x[o + 2 * ds], a3 = x[o + 3 * ds], a4 = x[o + 4 * ds];
      nr[m] = v4_near(a1, a2, a3, a4); fr[m] = v4_far(a1, a2, a3, a4);
    }
    xw[ijk]           = v4_near(nr[0], nr[1], nr[2], nr[3]);
    xw[ijk - dt]      = v4_far(nr[0], nr[1], nr[2], nr[3]);
    xw[ijk - ds]      = v4_near(fr[0], fr[1], fr[2], fr[3]);
    xw[ijk - ds - dt] = v4_far(fr[0], fr[1], fr[2], fr[3]);
  } else {
    const int di = s0, dj = s1, dk = s2;
    double nn[4], nf[4], fn[4], ff[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      double nj[4], fj[4];
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const int o = ijk + (m + 1) * dj + (p + 1) * dk;
        const double a1 = x[o + di], a2 = x[o + 2 * di], a3 = x[o + 3 * di], a4 = x[o + 4 * di];
        nj[m] = v4_near(a1, a2, a3, a4); fj[m] = v4_far(a1, a2, a3, a4);
      }
      nn[p] = v4_near(nj[0], nj[1], nj[2], nj[3]); nf[p] = v4_far(nj[0], nj[1], nj[2], nj[3]);
      fn[p] = v4_near(fj[0], fj[1], fj[2], fj[3]); ff[p] = v4_far(fj[0], fj[1], fj[2], fj[3]);
    }
    xw[ijk]                = v4_near(nn[0], nn[1], nn[2], nn[3]);
    xw[ijk - dk]           = v4_far(nn[0], nn[1], nn[2], nn[3]);
    xw[ijk - dj]           = v4_near(nf[0], nf[1], nf[2], nf[3]);
    xw[ijk - dj - dk]      = v4_far(nf[0], nf[1], nf[2], nf[3]);
    xw[ijk - di]           = v4_near(fn[0], fn[1], fn[2], fn[3]);
    xw[ijk - di - dk]      = v4_far(fn[0], fn[1], fn[2], fn[3]);
    xw[ijk - di - dj]      = v4_near(ff[0], ff[1], ff[2], ff[3]);
    xw[ijk - di - dj - dk] = v4_far(ff[0], ff[1], ff[2], ff[3]);
  }
}
// A corner (NN = 3) formed by a whole WAVE: the 64 reads and 84 extrapolations of bc_v4_cell<3> are a chain of ~5000 cycles for one lane -- on
// a level of one box, where a wave has an entry to itself, the eight corners set the time of the boundary stage.  Lane l takes row
// (m, p) = (l & 3, (l >> 2) & 3) of the 4 x 4 rows along i, the rows of a p are gathered by shuffles, then the four p: the same
// expression tree.  A corner is the work of SIXTEEN lanes: each aligned group of 16 lanes of the wave forms the corner its lanes were given (all
// four groups the same one when the wave has a single entry: the duplicate stores write the same values), the group's first lane writes.
template <typename CP, typename WP>
__device__ __forceinline__ void bc_v4_corner_wave(CP x, WP xw, int ijk, int di, int dj, int dk, int lane) {
  const int m = lane & 3, p = (lane >> 2) & 3, base = lane & 48;
  const int o = ijk + (m + 1) * dj + (p + 1) * dk;
  const double a1 = x[o + di], a2 = x[o + 2 * di], a3 = x[o + 3 * di], a4 = x[o + 4 * di];
  const double njv = v4_near(a1, a2, a3, a4), fjv = v4_far(a1, a2, a3, a4);
  double nj[4], fj[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { nj[q] = __shfl(njv, base | (p << 2) | q, 64); fj[q] = __shfl(fjv, base | (p << 2) | q, 64); }
  const double nnv = v4_near(nj[0], nj[1], nj[2], nj[3]), nfv = v4_far(nj[0], nj[1], nj[2], nj[3]);
  const double fnv = v4_near(fj[0], fj[1], fj[2], fj[3]), ffv = v4_far(fj[0], fj[1], fj[2], fj[3]);
  double nn[4], nf[4], fn[4], ff[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { nn[q] = __shfl(nnv, base | (q << 2), 64); nf[q] = __shfl(nfv, base | (q << 2), 64); fn[q] = __shfl(fnv, base | (q << 2), 64); ff[q] = __shfl(ffv, base | (q << 2), 64); }
  if ((lane & 15) == 0) {
    xw[ijk]                = v4_near(nn[0], nn[1], nn[2], nn[3]);
    xw[ijk - dk]           = v4_far(nn[0], nn[1], nn[2], nn[3]);
    xw[ijk - dj]           = v4_near(nf[0], nf[1], nf[2], nf[3]);
    xw[ijk - dj - dk]      = v4_far(nf[0], nf[1], nf[2], nf[3]);
    xw[ijk - di]           = v4_near(fn[0], fn[1], fn[2], fn[3]);
    xw[ijk - di - dk]      = v4_far(fn[0], fn[1], fn[2], fn[3]);
    xw[ijk - di - dj]      = v4_near(ff[0], ff[1], ff[2], ff[3]);
    xw[ijk - di - dj - dk] = v4_far(ff[0], ff[1], ff[2], ff[3]);
  }
}
// An edge (NN = 2) of up to 16 cells formed by a whole wave: four lanes per cell, lane m of them takes row m (4 reads, a near and a far
// value), the rows are gathered by shuffles inside the group and lane m forms and writes one of the four ghost cells: same expression tree.
template <typename CP, typename WP>
__device__ __forceinline__ void bc_v4_edge_wave(CP x, WP xw, int ijk, int ds, int dt, int lane, bool live) {
  const int m = lane & 3, o = ijk + (m + 1) * dt;
  const double a1 = x[o + ds], a2 = x[o + 2 * ds], a3 = x[o + 3 * ds], a4 = x[o + 4 * ds];
  const double nrv = v4_near(a1, a2, a3, a4), frv = v4_far(a1, a2, a3, a4);
  double nr[4], fr[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { nr[q] = __shfl(nrv, (lane & ~3) | q, 64); fr[q] = __shfl(frv, (lane & ~3) | q, 64); }
  const double r0 = (m & 2) ? fr[0] : nr[0], r1 = (m & 2) ? fr[1] : nr[1], r2 = (m & 2) ? fr[2] : nr[2], r3 = (m & 2) ? fr[3] : nr[3];
  const double v = (m & 1) ? v4_far(r0, r1, r2, r3) : v4_near(r0, r1, r2, r3);
  if (live) xw[ijk - ((m & 2) ? ds : 0) - ((m & 1) ? dt : 0)] = v;
}
template <typename P>
__device__ __forceinline__ void bc_v4_entry_at(P x, const hpgmg_hip_level &L, const blockCopy_type &e, int tid, int nth) {
  const BcGeom g = bc_geometry(L, e);
  const int n = g.len[0] * g.len[1];
  if (g.nn == 2 && n <= 16 && nth == 64) {                      // a wave to itself (the single-workgroup kernels): four lanes per cell
    const int t = min(tid >> 2, n - 1);                         // (g.len[1] == 1 for an edge: t runs along the one inside axis)
    const int ijk = (t % g.len[0] + g.lo[0]) * g.fstride[0] + (t / g.len[0] + g.lo[1]) * g.fstride[1];
    bc_v4_edge_wave(x, x, ijk + g.pos[0] + g.pos[1], g.step[0], g.step[1], tid, (tid >> 2) < n);
    return;
  }
  if (g.nn == 3 && n == 1 && nth == 64) {                       // a wave to itself (the single-workgroup kernels): the corner in parallel
    bc_v4_corner_wave(x, x, g.lo[0] * g.fstride[0] + g.lo[1] * g.fstride[1] + g.pos[0] + g.pos[1] + g.pos[2], g.step[0], g.step[1], g.step[2], tid);
    return;
  }
  for (int t = tid; t < n; t += nth) {
    const int r = t % g.len[0], q = t / g.len[0];
    const int ijk = (r + g.lo[0]) * g.fstride[0] + (q + g.lo[1]) * g.fstride[1];
    if (g.nn == 1)      bc_v4_cell<1>(x, x, ijk + g.pos[0], g.step[0], 0, 0);
    else if (g.nn == 2) bc_v4_cell<2>(x, x, ijk + g.pos[0] + g.pos[1], g.step[0], g.step[1], 0);
    else                bc_v4_cell<3>(x, x, ijk + g.pos[0] + g.pos[1] + g.pos[2], g.step[0], g.step[1], g.step[2]);
  }
}
// The same entry for a GROUP of lanes of a wave -- 16 for a corner, 32 for an edge (eight cells at a time), 64 for a face -- so that the entries of a
// box (8 corners, 12 edges, 6 faces) fill the lanes of a workgroup instead of taking a wave each: `sub` = the lane's index in its group, `lane`
// = its index in the wave (groups are aligned to their size).
template <typename P>
__device__ __forceinline__ void bc_v4_entry_packed(P x, const hpgmg_hip_level &L, const blockCopy_type &e, int sub, int lane) {
  const BcGeom g = bc_geometry(L, e);
  const int n = g.len[0] * g.len[1];
  if (g.nn == 3) {
    bc_v4_corner_wave(x, x, g.lo[0] * g.fstride[0] + g.lo[1] * g.fstride[1] + g.pos[0] + g.pos[1] + g.pos[2], g.step[0], g.step[1], g.step[2], lane);
  } else if (g.nn == 2) {
    for (int tb = 0; tb < n; tb += 8) {
      const int tc = tb + (sub >> 2), t = min(tc, n - 1);
      const int ijk = (t % g.len[0] + g.lo[0]) * g.fstride[0] + (t / g.len[0] + g.lo[1]) * g.fstride[1];
      bc_v4_edge_wave(x, x, ijk + g.pos[0] + g.pos[1], g.step[0], g.step[1], lane, tc < n);
    }
  } else {
    for (int t = sub; t < n; t += 64) {
      const int r = t % g.len[0], q = t / g.len[0];
      bc_v4_cell<1>(x, x, (r + g.lo[0]) * g.fstride[0] + (q + g.lo[1]) * g.fstride[1] + g.pos[0], g.step[0], 0, 0);
    }
  }
}
__device__ __forceinline__ void bc_v4_entry(const hpgmg_hip_level &L, int id, const blockCopy_type &e, int tid, int nth) {
  bc_v4_entry_at(vec_origin(L, e.read.box, id), L, e, tid, nth);
}

// the same conditions over entries whose geometry the host worked out (hpgmg_hip_bc_entry): one loop per kind, so the
// instructions a workgroup executes are few and contiguous.  ORDER: 2 = v2, 4 = v4, 12 = p2
// ORDER 1 = apply_BCs_p1 (boundary_fd.c:35-65): scale * x[ghost + inward normal], scale -1 / +1 / -1 for a face / an edge / a corner
template <int NN, typename CP, typename WP>
__device__ __forceinline__ void bc_p1_compact_cell(CP x, WP xw, int ijk, int s0, int s1, int s2) {
  const double scale = (NN == 2) ? 1.0 : -1.0;
  xw[ijk] = scale * x[ijk + s0 + (NN >= 2 ? s1 : 0) + (NN == 3 ? s2 : 0)];
}
template <int ORDER, int NN, typename CP, typename WP>
__device__ __forceinline__ void bc_compact_cell(CP x, WP xw, int ijk, int s0, int s1, int s2) {
  if (ORDER == 4) bc_v4_cell<NN>(x, xw, ijk, s0, s1, s2); else if (ORDER == 2) bc_v2_cell<NN>(x, xw, ijk, s0, s1, s2);
  else if (ORDER == 1) bc_p1_compact_cell<NN>(x, xw, ijk, s0, s1, s2); else bc_p2_cell<NN>(x, xw, ijk, s0, s1, s2);
}
template <int ORDER, bool REDIRECT, bool CLEAR = false>
__device__ __forceinline__ void bc_fv_compact_entry(const hpgmg_hip_level &L, int id, const hpgmg_hip_bc_entry &e, int tid, int nth) {
  if (CLEAR) {   // boundary_fv.c: the deeper ghost layers of the block are zero; the whole workgroup works on one entry, so a barrier orders the two passes
    double *z = vec_origin(L, e.box, id) + e.zbase;
    const int nz = e.zi * e.zj * e.zk;
    for (int t = tid; t < nz; t += nth) {
      const int i = t % e.zi, j = (t / e.zi) % e.zj, k = t / (e.zi * e.zj);
      z[i + j * L.jStride + k * L.kStride] = 0.0;
    }
    __syncthreads();
  }
  double *xw = vec_origin(L, e.box, id) + e.base;                                  // where the ghost cells are
  // where the cells they are formed from are read (same offsets): the entry's own box as apply_BCs does, or -- REDIRECT, the one-launch
  // form that does not wait for the exchange -- the box that owns them
  const double *x = REDIRECT ? vec_origin(L, e.src_box, id) + e.src_base : xw;
  const int n = e.len0 * e.len1;
  if (e.nn == 1) {
    for (int t = tid; t < n; t += nth) {
      const int q = t / e.len0, r = t - q * e.len0, ijk = r * e.fs0 + q * e.fs1;
      bc_compact_cell<ORDER, 1>(x, xw, ijk, e.step[0], 0, 0);
    }
  } else if (e.nn == 2) {
    for (int t = tid; t < n; t += nth) {
      const int ijk = t * e.fs0;                             // an edge runs along one axis
      bc_compact_cell<ORDER, 2>(x, xw, ijk, e.step[0], e.step[1], 0);
    }
  } else if (tid == 0) {
    bc_compact_cell<ORDER, 3>(x, xw, 0, e.step[0], e.step[1], e.step[2]);
  }
}


// 1-D rules of the tensor-product interpolations (v[] = coarse line, centre at v[R]):
//   order 2 = p2 interpolation_p2.c:90-92,150-205;  order 3 = v2 interpolation_v2.c:111-113;  order 4 = v4 interpolation_v4.c:96-97,180-240
template <int ORDER>
__device__ __forceinline__ double interp_rule(bool odd, const double *v) {
  if (ORDER == 2) {
    const double w0 = 5.0 / 32.0, w1 = 30.0 / 32.0, w2 = -3.0 / 32.0;
    return odd ? (w1 * v[1] + w2 * v[0] + w0 * v[2]) : (w1 * v[1] + w0 * v[0] + w2 * v[2]);
  } else if (ORDER == 3) {
    const double c1 = 1.0 / 8.0;
    return odd ? (v[1] - c1 * (v[0] - v[2])) : (v[1] + c1 * (v[0] - v[2]));
  } else {
    const double c1 = 22.0 / 128.0, c2 = -3.0 / 128.0;
    return odd ? (v[2] - c1 * (v[1] - v[3]) - c2 * (v[0] - v[4])) : (v[2] + c1 * (v[1] - v[3]) + c2 * (v[0] - v[4]));
  }
}

}  // namespace hpgmg

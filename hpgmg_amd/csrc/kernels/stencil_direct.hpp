// stencil_direct.hpp -- what the translation units of the stencil launchers share (stencil.hip, pair.hip, small_levels.hip): the operating
// modes and argument records of the single-sweep kernels, the per-cell "direct" form of apply_op_ijk for every operator (pointer types are
// template parameters, so the single-workgroup kernels can run it on LDS images), the 3 x 3 plane helpers of the 27-point operator, and
// the launch-profiling hooks.  Reference: operators.7pt.c:49-89, operators.27pt.c:60-91, operators.fv4.c:55-134.
#pragma once
#include <stdlib.h>
#include <cstring>
#include <vector>
#include "common.hpp"
#include "stencil_math.hpp"
#include "fv4_tile.hpp"
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
  const int *order;             // stencil7_wide_kernel, fused residual forms across rank boundaries: dispatch slot -> tile (nullptr: identity) of a
                                //    two-part launch (common.hpp tile_part_order): the tiles at a remote face run after the exchange has landed
};

// The piecewise-constant interpolation_vcycle that precedes a smooth() on the up leg of a V-cycle (mg.c:1160-1161, interpolation_p0.c:43:
// f = 1.0 * f + c[i>>1, j>>1, k>>1]) folded into the first two SINGLE Chebyshev sweeps of that smooth() (the sweep-pair kernel has its own fold):
// which == 1 (sweep 0): every value of x_n the stencil reads is taken as stored + the coarse value above it; which == 2 (sweep 1): x_{n-1}, the old
// value of the cell being written, is.  After sweep 1 the vector holds x2 and nothing remembers that the interpolated vector never existed.
struct InterpFold {
  int which;                       // 0: none
  hpgmg_hip_level Lc; int coarse_id;
  const int *map;                  // map[4 b .. 4 b + 3] = coarse box and coarse (i, j, k) under fine box b's first cell (as the fused restriction uses)
};
__device__ __forceinline__ double fold_coarse(const InterpFold &F, int box, int i, int j, int k) {
  const int *m = F.map + 4 * box;
  return F.Lc.box_base[m[0]][(size_t)F.coarse_id * (size_t)F.Lc.volume + (size_t)F.Lc.ghosts * (size_t)(1 + F.Lc.jStride + F.Lc.kStride)
                            + (size_t)((m[1] + (i >> 1)) + (m[2] + (j >> 1)) * F.Lc.jStride + (m[3] + (k >> 1)) * F.Lc.kStride)];
}

// x may alias the output only for in-place GSRB; everywhere else it is restrict-qualified
// so the loads of plane k+1 can be issued ahead of the store of plane k.
template <bool kMayAlias> struct src_ptr { typedef const double *__restrict__ type; };
template <> struct src_ptr<true> { typedef const double *type; };

int  profile_begin(long long cells);          // stencil.hip: hipEvent pair around a smoother launch (bench.py's roofline)
void profile_end(int p, long long cells, bool first_part = false);
static inline int env_int(const char *name, int dflt) { const char *e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
#ifdef HPGMG_EXP_TIMELINE
extern double *g_exp_timeline;     // experiment build: where a kernel's chosen workgroup records its step timeline (pair.hip)
#endif
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

}  // namespace hpgmg

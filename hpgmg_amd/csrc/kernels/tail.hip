// tail.hip -- the launch-bound tail of a V-cycle as ONE single-workgroup launch per leg.
//
// Below 16^3 cells a level visit of the reference's MGVCycle (mg.c:1147-1163) is a dozen
// dependent kernels of ~6 us each although the whole level fits in one CU's caches.  This
// kernel executes, for a chain of consecutive small levels l0 < l1 < ... < bottom,
//   leg 0 (down):  for each level but the last:  smooth; residual -> TEMP;
//                  restriction(next.R <- TEMP); zero_vector(next.e)
//   leg 1 (up):    for each level but the last, coarsest first:  interpolation_vcycle
//                  (e += P next.e, piecewise constant); smooth
// i.e. exactly the operator sequence the cycle driver would otherwise issue one launch at a
// time, with __syncthreads() where the driver had kernel boundaries.  The coarsest level of
// the chain is the bottom level: it is only the target of the restriction / source of the
// interpolation; its solve (BiCGStab) stays with the host between the two legs.
//
// One workgroup of 1024 lanes owns a level (<= 4096 cells: <= 4 per lane); the iterate and TEMP sit
// in LDS, coefficients in registers.  The arithmetic is the same expression tree as the streaming
// kernels (stencil_math.hpp, restriction.c:54-57, interpolation_p0.c:43), so results are
// bit-identical to the per-operator path -- tests/test_gpu_* run both.
// Ghost handling is the ghost-free form (neighbour box / Dirichlet -x, see stencil.hip), which
// requires every face neighbour to be local: the host only uses this kernel then.
#include "common.hpp"
#include "stencil_math.hpp"
#include "dense_levels.hpp"

namespace hpgmg {

constexpr int kTailMaxLevels = 8;
constexpr int kTailMaxSweeps = 8;
constexpr int kTailThreads = 1024;                  // lanes of a workgroup that owns up to 4 cells per lane (a chain that starts at 16^3)
constexpr int kTailThreadsSmall = 512;              // ... of one whose levels have at most 512 cells (from 8^3 down): one cell per lane, half the waves at every barrier
template <int CPL> constexpr int tail_threads() { return CPL == 1 ? kTailThreadsSmall : kTailThreads; }

struct TailLevel {
  hpgmg_hip_level L;
  double h2inv;
  double c1[kTailMaxSweeps], c2[kTailMaxSweeps];   // Chebyshev coefficients of THIS level (its own eigenvalue bound)
};
struct TailArgs {
  int n;                       // levels in the chain, the last one is only restriction target / interpolation source
  int e_id, R_id, sweeps;
  double a, b;
  int krylov_base;             // first of the 8 BiCGStab work vectors on the bottom level (r0,r,p,q,s,t,Ap,As)
  double bottom_norm;          // desired reduction of the bottom residual norm
  int *krylov_iterations;      // pinned host counter (+= iterations done), may be null
  TailLevel lv[kTailMaxLevels];
  int nops;                    // ftail_kernel: the operation sequence of the F-cycle, one byte each: kind << 4 | level
  unsigned char ops[112];
};
enum { FT_DOWN = 0, FT_UP = 1, FT_BOTTOM = 2, FT_RESTRICT_RHS = 3, FT_ZERO_BOTTOM = 4, FT_INTERP_F = 5 };

enum { SM_CHEBY = 0, SM_GSRB = 1, SM_JACOBI = 2, SM_RESIDUAL = 3 };

// The box-base tables of the chain's levels wait in LDS (CPL == 1 kernels; copied once at the start of the launch): `box_base[box]` read from memory is a
// round trip of its own in FRONT of every load and store of a visit.
constexpr int kTailTabBoxes = 64;                   // boxes per level the tables hold (levels of <= 512 cells; more: the 1024-lane kernels, tables in memory)
template <int CPL>
__device__ __forceinline__ hpgmg_hip_level tail_lvl(const TailArgs &A, int l, double *const *tab) {
  hpgmg_hip_level L = A.lv[l].L;
  if (CPL == 1) L.box_base = tab + l * kTailTabBoxes;
  return L;
}

// The level being worked on lives in LDS as two dense D^3 arrays in GLOBAL cell order (box boundaries
// disappear; a Dirichlet face is the in-register rule ghost = -centre, exactly what apply_BCs_p1 stores):
//   sx = the iterate (e), st = VECTOR_TEMP (Chebyshev/Jacobi ping-pong partner, then the residual).
// Each lane owns up to kCellsPerLane cells for the whole visit and keeps their coefficients
// (beta faces, alpha, Dinv, rhs) in registers, so a sweep is LDS reads + arithmetic + one barrier
// (~0.2 us) instead of a round trip to L2 (~1.3 us).  Global memory is read once and written once per
// level visit, leaving every vector in the state the per-operator sequence would leave it in.
constexpr int kCellsPerLane = 4;                    // 4096 cells / 1024 lanes
constexpr int kTailMaxCells = kCellsPerLane * kTailThreads;

// A x at LDS cell c of a D^3 level; src is the dense iterate
template <int V>
__device__ __forceinline__ double tail_apply(const double *src, int c, int gi, int gj, int gk, int D, const CellCoef<V> &q,
                                             double a, double b, double h2inv) {
  const double xc = src[c];
  const int last = D - 1;
  const double xim = (gi == 0)    ? -xc : src[c - 1];
  const double xip = (gi == last) ? -xc : src[c + 1];
  const double xjm = (gj == 0)    ? -xc : src[c - D];
  const double xjp = (gj == last) ? -xc : src[c + D];
  const double xkm = (gk == 0)    ? -xc : src[c - D * D];
  const double xkp = (gk == last) ? -xc : src[c + D * D];
  return apply_op_7pt<V>(xc, xim, xip, xjm, xjp, xkm, xkp, q.bi0, q.bi1, q.bj0, q.bj1, q.bk0, q.bk1, q.al, a, b, h2inv);
}

#ifdef HPGMG_EXP_TIMELINE      /* experiment builds: lane 0 records the 100 MHz clock at the ends of the routines and at the stages of a visit (tools/exp_tail_timeline.py) */
__device__ unsigned long long *g_tail_tl = nullptr;
__device__ int g_tail_tl_n = 0;
#define TAIL_MARK() do { if (threadIdx.x == 0 && g_tail_tl && g_tail_tl_n < 60) g_tail_tl[g_tail_tl_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TAIL_MARK() do { } while (0)
#endif
// What a visit hands to the NEXT one without a round trip through memory (CPL == 1: chains from 8^3 down, the ones the cycle runs since the 16^3 level went to
// the brick launches): the restricted residual goes to the next visit through LDS (sr, dense order) as well as to memory, and the correction of the coarser
// level (or of the bottom solve) is read from sx, where the visit that formed it left it -- the values the per-operator sequence would have read from memory.
// (Also tried: the next visit's coefficients, VECTOR_TEMP and stored e fetched AHEAD during the current visit -- three tails of config 1 74.6 us instead of
// 66.2, the F-cycle tail 58.9 instead of 53.2: the address arithmetic and a dozen loads more per visit on the one wave that is the critical path.)
struct TailCarry {
  int rhs_level;             // sr holds R of this level, dense order; -1: no
  int e_level;               // sx holds e of this level, dense order; -1: no
};
template <int V>
__device__ __forceinline__ void tail_fetch_cell(const TailArgs &A, const hpgmg_hip_level &L, const LevelGeom &G, int c, bool want_e, bool want_rhs,
                                                CellCoef<V> &q, double &temp, double &e, CellRef &where, int &gi, int &gj, int &gk) {
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  { const int cj = c / G.D; gi = c % G.D; gj = cj % G.D; gk = cj / G.D; }
  where = locate(G, gi, gj, gk);
  const int box = where.box, ijk = where.ijk, jS = L.jStride, kS = L.kStride;
  e = want_e ? gvec_origin(L, box, A.e_id)[ijk] : 0.0;
  temp = gvec_origin(L, box, VECTOR_TEMP)[ijk];
  q.rhs = want_rhs ? gvec_origin(L, box, A.R_id)[ijk] : 0.0;
  q.dinv = gvec_origin(L, box, VECTOR_DINV)[ijk];
  q.bi0 = q.bi1 = q.bj0 = q.bj1 = q.bk0 = q.bk1 = q.al = 0.0;
  if (kVC) {
    const gcptr bi = gvec_origin(L, box, VECTOR_BETA_I), bj = gvec_origin(L, box, VECTOR_BETA_J), bk = gvec_origin(L, box, VECTOR_BETA_K);
    q.bi0 = bi[ijk]; q.bi1 = bi[ijk + 1]; q.bj0 = bj[ijk]; q.bj1 = bj[ijk + jS]; q.bk0 = bk[ijk]; q.bk1 = bk[ijk + kS];
  }
  if (kHelm) q.al = gvec_origin(L, box, VECTOR_ALPHA)[ijk];
}
// one visit of a level: load -> (interpolate) -> smooth -> (residual, restrict, zero) -> store
template <int V, int SM, int CPL>
__device__ void tail_level(const TailArgs &A, double *const *tab, int l, int leg, double *sx, double *st, double *sr, TailCarry &carry) {
  constexpr int NT = tail_threads<CPL>();
  const TailLevel &T = A.lv[l];
  const hpgmg_hip_level L = tail_lvl<CPL>(A, l, tab);
  const int D = L.dim_i, total = D * D * D;
  const LevelGeom G = geom_of(L);
  CellCoef<V> q[CPL];
  int gi[CPL], gj[CPL], gk[CPL];
  CellRef where[CPL];
  double e0[CPL], t0[CPL];
  const bool rhs_here = (CPL == 1) && leg == 0 && carry.rhs_level == l;            // R waits in sr, e is the +0.0 the previous visit stored
  const bool parent_here = (CPL == 1) && leg == 1 && carry.e_level == l + 1;       // the coarser level's correction waits in sx

#pragma unroll
  for (int m = 0; m < CPL; m++) {
    const int c = threadIdx.x + m * NT;
    if (c < total) {
      tail_fetch_cell<V>(A, L, G, c, !rhs_here, !rhs_here, q[m], t0[m], e0[m], where[m], gi[m], gj[m], gk[m]);
      if (rhs_here) q[m].rhs = sr[c];
      if (leg == 1) {          // interpolation_vcycle: e = 1.0*e + (coarse parent), interpolation_p0.c:43
        const hpgmg_hip_level C = tail_lvl<CPL>(A, l + 1, tab);
        double parent;
        if (parent_here) { const int Dc = C.dim_i; parent = sx[(gi[m] >> 1) + Dc * ((gj[m] >> 1) + Dc * (gk[m] >> 1))]; }
        else { const CellRef p = locate(geom_of(C), gi[m] >> 1, gj[m] >> 1, gk[m] >> 1); parent = gvec_origin(C, p.box, A.e_id)[p.ijk]; }
        e0[m] = 1.0 * e0[m] + parent;
      }
    }
  }
  if (parent_here) __syncthreads();                   // every parent has been read: sx may take this level's iterate
#pragma unroll
  for (int m = 0; m < CPL; m++) {
    const int c = threadIdx.x + m * NT;
    if (c < total) { sx[c] = e0[m]; st[c] = t0[m]; }
  }
  __syncthreads();
  TAIL_MARK();

  // smooth(): chebyshev.c:43-99 / gsrb.c:24-132 / jacobi.c:17-62 (even number of sweeps: the result ends in sx)
  for (int s = 0; s < A.sweeps; s++) {
    const double *src = (SM != SM_GSRB && (s & 1)) ? st : sx;
    double *dst = (SM == SM_GSRB) ? sx : ((s & 1) ? sx : st);
#pragma unroll
    for (int m = 0; m < CPL; m++) {
      const int c = threadIdx.x + m * NT;
      if (c < total) {
        if (SM == SM_GSRB) {
          const int colour = (gi[m] ^ gj[m] ^ gk[m] ^ s) & 1;   // global parity: box.low folded in (gsrb.c:55)
          if (colour != 0) continue;
        }
        const double xc = src[c];
        const double Ax = tail_apply<V>(src, c, gi[m], gj[m], gk[m], D, q[m], A.a, A.b, T.h2inv);
        if (SM == SM_CHEBY)     { const double xnm1 = dst[c]; dst[c] = xc + T.c1[s] * (xc - xnm1) + T.c2[s] * q[m].dinv * (q[m].rhs - Ax); }
        else if (SM == SM_GSRB) { dst[c] = xc + q[m].dinv * (q[m].rhs - Ax); }
        else                    { dst[c] = xc + (2.0 / 3.0) * q[m].dinv * (q[m].rhs - Ax); }
      }
    }
    __syncthreads();
  }

  TAIL_MARK();
  if (leg == 0) {                                     // residual -> TEMP (residual.c:42-48)
#pragma unroll
    for (int m = 0; m < CPL; m++) {
      const int c = threadIdx.x + m * NT;
      if (c < total) {
        const double Ax = tail_apply<V>(sx, c, gi[m], gj[m], gk[m], D, q[m], A.a, A.b, T.h2inv);
        st[c] = q[m].rhs - Ax;                        // each lane overwrites only its own TEMP cells: no hazard with the reads of sx
      }
    }
    __syncthreads();
  }

  TAIL_MARK();
  // leave e and TEMP in global memory as the per-operator sequence would
#pragma unroll
  for (int m = 0; m < CPL; m++) {
    const int c = threadIdx.x + m * NT;
    if (c < total) {
      gvec_origin(L, where[m].box, A.e_id)[where[m].ijk] = sx[c];
      gvec_origin(L, where[m].box, VECTOR_TEMP)[where[m].ijk] = st[c];
    }
  }
  carry.e_level = l; carry.rhs_level = -1;

  if (leg == 0) {
    // restriction(next.R <- TEMP): 0.125 * sum of the 8 children in the reference's order (restriction.c:54-57)
    const hpgmg_hip_level C = tail_lvl<CPL>(A, l + 1, tab);
    const LevelGeom GC = geom_of(C);
    const int Dc = D / 2, totc = Dc * Dc * Dc;
    for (int c = threadIdx.x; c < totc; c += NT) {
      const int cjk = c / GC.D, ci = c % GC.D, cj = cjk % GC.D, ck = cjk / GC.D;
      const double *f = st + 2 * ci + 2 * cj * D + 2 * ck * D * D;
      double v = f[0] + f[1]; v = v + f[D]; v = v + f[1 + D]; v = v + f[D * D]; v = v + f[1 + D * D]; v = v + f[D + D * D]; v = v + f[1 + D + D * D];
      const CellRef p = locate(GC, ci, cj, ck);
      gvec_origin(C, p.box, A.R_id)[p.ijk] = v * 0.125;
      if (CPL == 1 && totc <= NT) sr[c] = v * 0.125;
    }
    if (CPL == 1 && totc <= NT) carry.rhs_level = l + 1;
    // zero_vector(next.e): the whole padded box, ghosts included (misc.c:6-44); the alignment padding between
    // rows is never read and was zero-filled at allocation, so the box's slab is cleared as one contiguous run
    for (int box = 0; box < C.num_boxes; box++) {
      const gptr z = as_global(C.box_base[box]) + (size_t)A.e_id * (size_t)C.volume;
      for (int c = threadIdx.x; c < C.volume; c += NT) z[c] = 0.0;
    }
  }
  __syncthreads();                                    // global writes of this level are visible to the next level's loads
}

// ---- the bottom solve: diagonally preconditioned BiCGStab (solvers/bicgstab.c:14-97) --------------
// One cell per lane (<= kBottomMaxCells cells), every vector a register; the vector an operator is
// applied to passes through LDS for its neighbours.  The operation sequence, the expression of every
// BLAS-1 step (misc.c: c = sa*a + sb*b, c = s*a*b) and the break-down tests are those of the host
// driver (host/solvers.c), and the sums keep the reference's order -- one partial per dim x 8 x 8
// tile accumulated k,j,i, partials added in tile order (misc.c:261-269) -- so the iterates, the
// iteration count and the coarse correction are bit-identical to the host-driven solve.
constexpr int kBottomMaxCells = kTailThreads;      // (NT below: the lanes of the workgroup = the most cells a bottom level may have in it)
struct BottomGeom { int D, total, c, gi, gj, gk, bd, nb, tiles_side, tiles_per_box, ntiles; bool active; };

template <int NT>
__device__ double bottom_dot(const BottomGeom &g, double va, double vb, double *scr) {
  if (g.active) scr[g.c] = va * vb;
  __syncthreads();
  const int t = threadIdx.x;
  if (t < g.ntiles) {
    const int box = t / g.tiles_per_box, rem = t % g.tiles_per_box;
    const int k0 = (rem / g.tiles_side) * BLOCKCOPY_TILE_K, j0 = (rem % g.tiles_side) * BLOCKCOPY_TILE_J;
    const int k1 = min(k0 + BLOCKCOPY_TILE_K, g.bd), j1 = min(j0 + BLOCKCOPY_TILE_J, g.bd);
    const int oi = (box % g.nb) * g.bd, oj = ((box / g.nb) % g.nb) * g.bd, ok = (box / (g.nb * g.nb)) * g.bd;
    double acc = 0.0;
    for (int k = k0; k < k1; k++) for (int j = j0; j < j1; j++) {
      const double *row = scr + oi + g.D * ((oj + j) + g.D * (ok + k));
      for (int i = 0; i < g.bd; i++) acc += row[i];
    }
    scr[NT + t] = acc;
  }
  __syncthreads();
  if (t == 0) { double sum = 0.0; for (int q = 0; q < g.ntiles; q++) sum += scr[NT + q]; scr[2 * NT] = sum; }
  __syncthreads();
  return scr[2 * NT];
}
template <int NT>
__device__ double bottom_norm(const BottomGeom &g, double v, double *scr) {          // max |v| (misc.c:303-349)
  double m = 0.0;
  if (g.active) { const double f = fabs(v); m = (f > m) ? f : m; }
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(m, off, 64); m = (o > m) ? o : m; }
  double *w = scr + 2 * NT + 8;
  if (threadIdx.x % 64 == 0) w[threadIdx.x / 64] = m;
  __syncthreads();
  m = w[0];
  for (int q = 1; q < NT / 64; q++) m = (w[q] > m) ? w[q] : m;
  __syncthreads();
  return m;
}
template <int V>
__device__ double bottom_apply(const BottomGeom &g, double v, const CellCoef<V> &q, double a, double b, double h2inv, double *sx) {
  if (g.active) sx[g.c] = v;
  __syncthreads();
  double Ax = 0.0;
  if (g.active) Ax = tail_apply<V>(sx, g.c, g.gi, g.gj, g.gk, g.D, q, a, b, h2inv);
  __syncthreads();
  return Ax;
}

template <int V, int CPL>
__device__ void tail_bottom(const TailArgs &A, double *const *tab, double *sx, double *scr, TailCarry &carry) {
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  const TailLevel &T = A.lv[A.n - 1];
  const hpgmg_hip_level L = tail_lvl<CPL>(A, A.n - 1, tab);
  BottomGeom g;
  g.D = L.dim_i; g.total = g.D * g.D * g.D; g.c = threadIdx.x; g.active = g.c < g.total;
  const LevelGeom G = geom_of(L);
  { const int cj = g.c / G.D; g.gi = g.c % G.D; g.gj = cj % G.D; g.gk = cj / G.D; }
  g.bd = L.dim; g.nb = G.nb.d;
  g.tiles_side = (g.bd + BLOCKCOPY_TILE_J - 1) / BLOCKCOPY_TILE_J; g.tiles_per_box = g.tiles_side * g.tiles_side;
  g.ntiles = g.tiles_per_box * L.num_boxes;
  const int r0_id = A.krylov_base, r_id = r0_id + 1, p_id = r0_id + 2, q_id = r0_id + 3, s_id = r0_id + 4, t_id = r0_id + 5, Ap_id = r0_id + 6, As_id = r0_id + 7;

  CellCoef<V> cf;
  cf.bi0 = cf.bi1 = cf.bj0 = cf.bj1 = cf.bk0 = cf.bk1 = cf.al = cf.dinv = cf.rhs = 0.0;
  CellRef at = {0, 0};
  double x = 0, r0 = 0, r = 0, p = 0, q = 0, sv = 0, tv = 0, Ap = 0, As = 0, tmp = 0;
  if (g.active) {
    at = locate(G, g.gi, g.gj, g.gk);
    const int box = at.box, ijk = at.ijk, jS = L.jStride, kS = L.kStride;
    x = gvec_origin(L, box, A.e_id)[ijk];
    cf.rhs = gvec_origin(L, box, A.R_id)[ijk];
    cf.dinv = gvec_origin(L, box, VECTOR_DINV)[ijk];
    if (kVC) {
      const gcptr bi = gvec_origin(L, box, VECTOR_BETA_I), bj = gvec_origin(L, box, VECTOR_BETA_J), bk = gvec_origin(L, box, VECTOR_BETA_K);
      cf.bi0 = bi[ijk]; cf.bi1 = bi[ijk + 1]; cf.bj0 = bj[ijk]; cf.bj1 = bj[ijk + jS]; cf.bk0 = bk[ijk]; cf.bk1 = bk[ijk + kS];
    }
    if (kHelm) cf.al = gvec_origin(L, box, VECTOR_ALPHA)[ijk];
    // work vectors keep whatever an early exit leaves untouched
    r0 = gvec_origin(L, box, r0_id)[ijk]; r = gvec_origin(L, box, r_id)[ijk]; p = gvec_origin(L, box, p_id)[ijk]; q = gvec_origin(L, box, q_id)[ijk];
    sv = gvec_origin(L, box, s_id)[ijk]; tv = gvec_origin(L, box, t_id)[ijk]; Ap = gvec_origin(L, box, Ap_id)[ijk]; As = gvec_origin(L, box, As_id)[ijk];
    tmp = gvec_origin(L, box, VECTOR_TEMP)[ijk];
  }
  const double a = A.a, b = A.b, h2inv = T.h2inv, want = A.bottom_norm;
  int it = 0;

  // the solver (solvers/bicgstab.c:14-97) written once over three primitives -- operator, dot product, max norm -- so the general form
  // (one cell per lane, block-wide reductions) and the one-cell form below run literally the same sequence of operations
#define HPGMG_BICGSTAB(APPLY, DOT, NORM)                                                                   \
  r0 = cf.rhs - APPLY(x);                                            /* residual(r0, x, R) */               \
  r = 1.0 * r0;                                                                                            \
  p = 1.0 * r0;                                                                                            \
  {                                                                                                        \
    double rho = DOT(r, r0);                                                                               \
    const double r0_norm = NORM(r);                                                                        \
    if (!(rho == 0.0 || r0_norm == 0.0)) {                                                                 \
      while (it < 200) {                                                                                   \
        it++;                                                                                              \
        q = 1.0 * cf.dinv * p;                                                                             \
        Ap = APPLY(q);                                                                                     \
        const double Ap_r0 = DOT(Ap, r0);                                                                  \
        if (Ap_r0 == 0.0) break;                                                                           \
        const double alpha = rho / Ap_r0;                                                                  \
        if (__builtin_isinf(alpha)) break;                                                                 \
        x = 1.0 * x + alpha * q;                                                                           \
        sv = 1.0 * r + (-alpha) * Ap;                                                                      \
        const double s_norm = NORM(sv);                                                                    \
        if (s_norm == 0.0 || s_norm < want * r0_norm) break;                                               \
        tv = 1.0 * cf.dinv * sv;                                                                           \
        As = APPLY(tv);                                                                                    \
        const double As_As = DOT(As, As);                                                                  \
        const double As_s = DOT(As, sv);                                                                   \
        if (As_As == 0.0) break;                                                                           \
        const double omega = As_s / As_As;                                                                 \
        if (omega == 0.0 || __builtin_isinf(omega)) break;                                                 \
        x = 1.0 * x + omega * tv;                                                                          \
        r = 1.0 * sv + (-omega) * As;                                                                      \
        const double r_norm = NORM(r);                                                                     \
        if (r_norm == 0.0 || r_norm < want * r0_norm) break;                                               \
        const double rho_new = DOT(r, r0);                                                                 \
        if (rho_new == 0.0) break;                                                                         \
        const double beta = (rho_new / rho) * (alpha / omega);                                             \
        if (__builtin_isinf(beta)) break;                                                                  \
        tmp = 1.0 * p + (-omega) * Ap;                                                                     \
        p = 1.0 * r + beta * tmp;                                                                          \
        rho = rho_new;                                                                                     \
      }                                                                                                    \
    }                                                                                                      \
  }

  if (g.total == 1) {
    // a 1^3 bottom level (config 2: 256^3 -> ... -> 1^3): lane 0 solves alone, every "reduction" is over one cell.  Same expressions:
    // the operator with all six neighbours = the Dirichlet ghost -v (tail_apply with D = 1); the ordered sum of one product, 0.0 + (0.0 + a b)
    // as bottom_dot forms it; max(0, |v|).  No barriers instead of ~25 block-wide ones per solve.
    if (threadIdx.x == 0) {
      auto apply1 = [&](double v) { return apply_op_7pt<V>(v, -v, -v, -v, -v, -v, -v, cf.bi0, cf.bi1, cf.bj0, cf.bj1, cf.bk0, cf.bk1, cf.al, a, b, h2inv); };
      auto dot1 = [&](double va, double vb) { double acc = 0.0; acc += va * vb; double sum = 0.0; sum += acc; return sum; };
      auto norm1 = [&](double v) { double m = 0.0; const double f = fabs(v); m = (f > m) ? f : m; return m; };
      HPGMG_BICGSTAB(apply1, dot1, norm1)
    }
  } else {
    auto applyN = [&](double v) { return bottom_apply<V>(g, v, cf, a, b, h2inv, sx); };
    auto dotN = [&](double va, double vb) { return bottom_dot<tail_threads<CPL>()>(g, va, vb, scr); };
    auto normN = [&](double v) { return bottom_norm<tail_threads<CPL>()>(g, v, scr); };
    HPGMG_BICGSTAB(applyN, dotN, normN)
  }
#undef HPGMG_BICGSTAB
  if (g.active) {
    const int box = at.box, ijk = at.ijk;
    gvec_origin(L, box, A.e_id)[ijk] = x;
    gvec_origin(L, box, r0_id)[ijk] = r0; gvec_origin(L, box, r_id)[ijk] = r; gvec_origin(L, box, p_id)[ijk] = p; gvec_origin(L, box, q_id)[ijk] = q;
    gvec_origin(L, box, s_id)[ijk] = sv; gvec_origin(L, box, t_id)[ijk] = tv; gvec_origin(L, box, Ap_id)[ijk] = Ap; gvec_origin(L, box, As_id)[ijk] = As;
    gvec_origin(L, box, VECTOR_TEMP)[ijk] = tmp;
  }
  if (threadIdx.x == 0 && A.krylov_iterations) *A.krylov_iterations += it;
  if (g.active) sx[g.c] = x;                          // the correction, where the up leg that follows looks for it first
  carry.e_level = A.n - 1; carry.rhs_level = -1;
  __syncthreads();
}

// ---- the F-cycle on the chain (leg 4): what FMGSolve does below the chain's first level (mg.c:1270-1300) ----
// restriction(next.R <- R, RESTRICT_CELL) of the right-hand side, level by level (restriction.c:54-57)
template <int NT, int CPL>
__device__ void tail_restrict_rhs(const TailArgs &A, double *const *tab, int l) {
  const hpgmg_hip_level F = tail_lvl<CPL>(A, l, tab), C = tail_lvl<CPL>(A, l + 1, tab);
  const LevelGeom GF = geom_of(F), GC = geom_of(C);
  const int Dc = C.dim_i, totc = Dc * Dc * Dc;
  for (int c = threadIdx.x; c < totc; c += NT) {
    const int cjk = c / GC.D, ci = c % GC.D, cj = cjk % GC.D, ck = cjk / GC.D;
    double f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { const CellRef r = locate(GF, 2 * ci + (q & 1), 2 * cj + ((q >> 1) & 1), 2 * ck + (q >> 2)); f[q] = gvec_origin(F, r.box, A.R_id)[r.ijk]; }
    double v = f[0] + f[1]; v = v + f[2]; v = v + f[3]; v = v + f[4]; v = v + f[5]; v = v + f[6]; v = v + f[7];
    const CellRef p = locate(GC, ci, cj, ck);
    gvec_origin(C, p.box, A.R_id)[p.ijk] = v * 0.125;
  }
  __syncthreads();
}
// interpolation_fcycle(level l <- level l+1), piecewise linear (interpolation_p1.c:40-70): f = 0.0 * f + 27/64 c + 9/64 (3 face
// neighbours) + 3/64 (3 edge neighbours) + 1/64 corner, an even fine cell leaning on the coarse neighbour behind it, an odd one on the
// one ahead.  The reference first fills the coarse ghost cells (exchange_boundary + apply_BCs_p1, BOX shape: a ghost cell is
// -, +, - its mirror image for 1, 2, 3 directions leaving the domain, boundary_fd.c:35-38); here the same value is formed on the fly.
template <int NT, int CPL>
__device__ void tail_interp_fcycle(const TailArgs &A, double *const *tab, int l) {
  const hpgmg_hip_level F = tail_lvl<CPL>(A, l, tab), C = tail_lvl<CPL>(A, l + 1, tab);
  const LevelGeom GF = geom_of(F), GC = geom_of(C);
  const int D = F.dim_i, total = D * D * D, Dc = C.dim_i;
  auto coarse = [&](int ci, int cj, int ck) -> double {
    double s = 1.0;
    if (ci < 0) { ci = 0; s = -s; } else if (ci >= Dc) { ci = Dc - 1; s = -s; }
    if (cj < 0) { cj = 0; s = -s; } else if (cj >= Dc) { cj = Dc - 1; s = -s; }
    if (ck < 0) { ck = 0; s = -s; } else if (ck >= Dc) { ck = Dc - 1; s = -s; }
    const CellRef r = locate(GC, ci, cj, ck);
    return s * gvec_origin(C, r.box, A.e_id)[r.ijk];
  };
  for (int c = threadIdx.x; c < total; c += NT) {
    const int cj_ = c / GF.D, gi = c % GF.D, gj = cj_ % GF.D, gk = cj_ / GF.D;
    const int ci = gi >> 1, cj = gj >> 1, ck = gk >> 1, di = (gi & 1) ? 1 : -1, dj = (gj & 1) ? 1 : -1, dk = (gk & 1) ? 1 : -1;
    const CellRef w = locate(GF, gi, gj, gk);
    const gptr fp = gvec_origin(F, w.box, A.e_id) + w.ijk;
    double v = 0.0 * (*fp);
    v = v + 0.421875 * coarse(ci, cj, ck);
    v = v + 0.140625 * coarse(ci, cj, ck + dk);
    v = v + 0.140625 * coarse(ci, cj + dj, ck);
    v = v + 0.046875 * coarse(ci, cj + dj, ck + dk);
    v = v + 0.140625 * coarse(ci + di, cj, ck);
    v = v + 0.046875 * coarse(ci + di, cj, ck + dk);
    v = v + 0.046875 * coarse(ci + di, cj + dj, ck);
    v = v + 0.015625 * coarse(ci + di, cj + dj, ck + dk);
    *fp = v;
  }
  __syncthreads();
}

// The legs of a V-cycle (0: down legs | 1: up legs | 2: down, bottom solve, up | 3: bottom solve only) and the F-cycle on the chain (leg 4) are both a
// SEQUENCE of routines the host wrote out (hpgmg_hip_vcycle_tail); interpreting it keeps ONE call site per routine, so everything inlines and TailArgs stays
// in the kernel-argument segment (several call sites made the compiler copy it to scratch: 3x slower).  CPL: cells per lane -- 1 (512 lanes) for chains from
// 8^3 down (the usual ones), 4 (1024 lanes) for a chain that starts at 16^3.
template <int V, int SM, int CPL>
__global__ __launch_bounds__(tail_threads<CPL>()) void tail_kernel(const TailArgs A) {
  constexpr int NT = tail_threads<CPL>();
  __shared__ double sx[CPL * NT];
  __shared__ double st[(CPL > 3 ? CPL : 3) * NT];      // (the bottom solve's reductions use 2 NT + 24 doubles of it)
  double *sr = nullptr;                                          // (CPL == 1 only: the 64 KB of static LDS are taken otherwise)
  if constexpr (CPL == 1) { __shared__ double sr1[NT]; sr = sr1; }
#ifdef HPGMG_EXP_TIMELINE
  if (threadIdx.x == 0) g_tail_tl_n = 0;
#endif
  double *const *tab = nullptr;
  if constexpr (CPL == 1) {
    // the box-base tables of every level of the chain, and a first touch of every level's descriptor (kernel-argument memory: a scalar-cache miss is ~1 us too)
    __shared__ double *s_tab[kTailMaxLevels * kTailTabBoxes];
    double warm = 0.0;
#pragma unroll
    for (int l = 0; l < kTailMaxLevels; l++)
      if (l < A.n) {
        if ((int)threadIdx.x < A.lv[l].L.num_boxes && (int)threadIdx.x < kTailTabBoxes) s_tab[l * kTailTabBoxes + threadIdx.x] = A.lv[l].L.box_base[threadIdx.x];
        warm += A.lv[l].h2inv + A.lv[l].c1[kTailMaxSweeps - 1] + A.lv[l].c2[kTailMaxSweeps - 1] + (double)A.lv[l].L.box_stride;
      }
    if (warm == 1.2345e300) sx[0] = warm;      // (keeps the touches)
    tab = s_tab;
    __syncthreads();
  }
  TailCarry carry;
  carry.rhs_level = -1; carry.e_level = -1;
  TAIL_MARK();
  for (int i = 0; i < A.nops; i++) {
    const int kind = A.ops[i] >> 4, l = A.ops[i] & 15;
    if (kind == FT_DOWN || kind == FT_UP) tail_level<V, SM, CPL>(A, tab, l, kind, sx, st, sr, carry);
    else tail_bottom<V, CPL>(A, tab, sx, st, carry);
    TAIL_MARK();
  }
#ifdef HPGMG_EXP_TIMELINE
  if (threadIdx.x == 0 && g_tail_tl) g_tail_tl[63] = (unsigned long long)g_tail_tl_n;
#endif
}
// leg 4: the F-cycle on the chain: right-hand side restricted down, bottom solve, then per level upwards interpolation_fcycle + a V-cycle.
// (Its own kernel: folded into tail_kernel the extra code cost every V-cycle launch registers -- 48 -> 130 us per launch.)
template <int V, int SM, int CPL>
__global__ __launch_bounds__(tail_threads<CPL>()) void ftail_kernel(const TailArgs A) {
  constexpr int NT = tail_threads<CPL>();
  __shared__ double sx[CPL * NT];
  __shared__ double st[(CPL > 3 ? CPL : 3) * NT];      // (the bottom solve's reductions use 2 NT + 24 doubles of it)
  double *sr = nullptr;                                          // (CPL == 1 only: the 64 KB of static LDS are taken otherwise)
  if constexpr (CPL == 1) { __shared__ double sr1[NT]; sr = sr1; }
  double *const *tab = nullptr;
  if constexpr (CPL == 1) {
    // the box-base tables of every level of the chain, and a first touch of every level's descriptor (kernel-argument memory: a scalar-cache miss is ~1 us too)
    __shared__ double *s_tab[kTailMaxLevels * kTailTabBoxes];
    double warm = 0.0;
#pragma unroll
    for (int l = 0; l < kTailMaxLevels; l++)
      if (l < A.n) {
        if ((int)threadIdx.x < A.lv[l].L.num_boxes && (int)threadIdx.x < kTailTabBoxes) s_tab[l * kTailTabBoxes + threadIdx.x] = A.lv[l].L.box_base[threadIdx.x];
        warm += A.lv[l].h2inv + A.lv[l].c1[kTailMaxSweeps - 1] + A.lv[l].c2[kTailMaxSweeps - 1] + (double)A.lv[l].L.box_stride;
      }
    if (warm == 1.2345e300) sx[0] = warm;      // (keeps the touches)
    tab = s_tab;
    __syncthreads();
  }
  TailCarry carry;
  carry.rhs_level = -1; carry.e_level = -1;
  for (int i = 0; i < A.nops; i++) {
    const int kind = A.ops[i] >> 4, l = A.ops[i] & 15;
    if (kind == FT_DOWN || kind == FT_UP) tail_level<V, SM, CPL>(A, tab, l, kind, sx, st, sr, carry);
    else if (kind == FT_BOTTOM) tail_bottom<V, CPL>(A, tab, sx, st, carry);
    else if (kind == FT_RESTRICT_RHS) tail_restrict_rhs<NT, CPL>(A, tab, l);
    else if (kind == FT_INTERP_F) tail_interp_fcycle<NT, CPL>(A, tab, l);
    else { // zero_vector(bottom, e): whole padded boxes (misc.c:6-44)
      const hpgmg_hip_level B = tail_lvl<CPL>(A, A.n - 1, tab);
      for (int box = 0; box < B.num_boxes; box++) { const gptr z = as_global(B.box_base[box]) + (size_t)A.e_id * (size_t)B.volume; for (int c = threadIdx.x; c < B.volume; c += NT) z[c] = 0.0; }
      __syncthreads();
    }
  }
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

#ifdef HPGMG_EXP_TIMELINE
void hpgmg_hip_exp_tail_timeline(void *buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tail_tl), &buf, sizeof(buf)); }
#endif
int hpgmg_hip_tail_max_levels(void) { return kTailMaxLevels; }
int hpgmg_hip_tail_max_cells(void) { return kTailMaxCells; }
int hpgmg_hip_tail_bottom_max_cells(void) { return kBottomMaxCells; }

// levels[0..n-1]: finest..coarsest of the chain; per level h2inv and Chebyshev coefficients (sweeps of them).
// leg 0/1: the two legs around a host-driven bottom solve; leg 2: legs and bottom solve in one launch;
// leg 3: the bottom solve alone (n may be 1).
int hpgmg_hip_vcycle_tail(int n, const hpgmg_hip_level *const *levels, const double *h2inv,
                          const double *c1, const double *c2, int sweeps,
                          int variant, int smoother, int e_id, int R_id, double a, double b, int leg,
                          int krylov_base, double bottom_norm, int *krylov_iterations) {
  HPGMG_SKIP_IF_REPLAY();
  if (n < 1 || (n < 2 && leg != 3) || n > kTailMaxLevels || sweeps > kTailMaxSweeps || leg < 0 || leg > 4)
    return record_error(hipErrorInvalidValue, "vcycle_tail: chain length / leg");
  TailArgs A = {};
  A.n = n; A.e_id = e_id; A.R_id = R_id; A.sweeps = sweeps; A.a = a; A.b = b;
  A.krylov_base = krylov_base; A.bottom_norm = bottom_norm; A.krylov_iterations = krylov_iterations;
  for (int l = 0; l < n; l++) {
    A.lv[l].L = *levels[l];
    A.lv[l].h2inv = h2inv[l];
    for (int s = 0; s < sweeps; s++) { A.lv[l].c1[s] = c1[l * sweeps + s]; A.lv[l].c2[s] = c2[l * sweeps + s]; }
    const long long cells = (long long)levels[l]->dim_i * levels[l]->dim_j * levels[l]->dim_k;
    if (l + 1 < n && cells > kTailMaxCells) return record_error(hipErrorInvalidValue, "vcycle_tail: level too large");
    if (l + 1 == n && leg >= 2 && cells > kBottomMaxCells) return record_error(hipErrorInvalidValue, "vcycle_tail: bottom level too large");
  }
  int cpl = 1;      // cells per lane: 1 when every level of the chain has at most 512 cells (a workgroup of 512 lanes), else 4 (1024 lanes)
  for (int l = 0; l < n; l++) if ((long long)levels[l]->dim_i * levels[l]->dim_j * levels[l]->dim_k > kTailThreadsSmall || levels[l]->num_boxes > kTailTabBoxes) cpl = kCellsPerLane;
  if (leg != 4) {   // the legs of a V-cycle, written out as a sequence
    int q = 0;
    if (leg == 0 || leg == 2) for (int l = 0; l + 1 < n; l++) A.ops[q++] = (unsigned char)((FT_DOWN << 4) | l);
    if (leg == 2 || leg == 3) A.ops[q++] = (unsigned char)(FT_BOTTOM << 4);
    if (leg == 1 || leg == 2) for (int l = n - 2; l >= 0; l--) A.ops[q++] = (unsigned char)((FT_UP << 4) | l);
    A.nops = q;
  }
  if (leg == 4) {   // FMGSolve below levels[0] (mg.c:1270-1300), written out as a sequence
    int q = 0;
    for (int l = 0; l + 1 < n; l++) A.ops[q++] = (unsigned char)((FT_RESTRICT_RHS << 4) | l);
    A.ops[q++] = (unsigned char)(FT_ZERO_BOTTOM << 4);
    A.ops[q++] = (unsigned char)(FT_BOTTOM << 4);
    for (int top = n - 2; top >= 0; top--) {
      A.ops[q++] = (unsigned char)((FT_INTERP_F << 4) | top);
      for (int l = top; l + 1 < n; l++) A.ops[q++] = (unsigned char)((FT_DOWN << 4) | l);
      A.ops[q++] = (unsigned char)(FT_BOTTOM << 4);
      for (int l = n - 2; l >= top; l--) A.ops[q++] = (unsigned char)((FT_UP << 4) | l);
    }
    if (q > (int)sizeof(A.ops)) return record_error(hipErrorInvalidValue, "vcycle_tail: F-cycle sequence too long");
    A.nops = q;
  }
#define TAIL_CASE(V, SM) do { if (leg == 4) { if (cpl == 1) hipLaunchKernelGGL((ftail_kernel<V, SM, 1>), dim3(1), dim3(kTailThreadsSmall), 0, g_stream, A); \
                                              else hipLaunchKernelGGL((ftail_kernel<V, SM, kCellsPerLane>), dim3(1), dim3(kTailThreads), 0, g_stream, A); } \
                              else { if (cpl == 1) hipLaunchKernelGGL((tail_kernel<V, SM, 1>), dim3(1), dim3(kTailThreadsSmall), 0, g_stream, A); \
                                     else hipLaunchKernelGGL((tail_kernel<V, SM, kCellsPerLane>), dim3(1), dim3(kTailThreads), 0, g_stream, A); } } while (0)
  const int key = variant * 3 + smoother;
  switch (key) {
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + SM_CHEBY:  TAIL_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ, SM_CHEBY); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + SM_GSRB:   TAIL_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ, SM_GSRB); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + SM_JACOBI: TAIL_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ, SM_JACOBI); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + SM_CHEBY:    TAIL_CASE(HPGMG_HIP_7PT_VC_POISSON, SM_CHEBY); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + SM_GSRB:     TAIL_CASE(HPGMG_HIP_7PT_VC_POISSON, SM_GSRB); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + SM_JACOBI:   TAIL_CASE(HPGMG_HIP_7PT_VC_POISSON, SM_JACOBI); break;
    case HPGMG_HIP_7PT_CC * 3 + SM_CHEBY:            TAIL_CASE(HPGMG_HIP_7PT_CC, SM_CHEBY); break;
    case HPGMG_HIP_7PT_CC * 3 + SM_GSRB:             TAIL_CASE(HPGMG_HIP_7PT_CC, SM_GSRB); break;
    case HPGMG_HIP_7PT_CC * 3 + SM_JACOBI:           TAIL_CASE(HPGMG_HIP_7PT_CC, SM_JACOBI); break;
    default: return record_error(hipErrorInvalidValue, "vcycle_tail: variant/smoother");
  }
#undef TAIL_CASE
  HPGMG_LAUNCH_CHECK("tail_kernel");
  return 0;
}

}  // extern "C"

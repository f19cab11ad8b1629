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

namespace hpgmg {

constexpr int kTailMaxLevels = 8;
constexpr int kTailMaxSweeps = 8;
constexpr int kTailThreads = 1024;

struct TailLevel {
  hpgmg_hip_level L;
  double h2inv;
  double c1[kTailMaxSweeps], c2[kTailMaxSweeps];   // Chebyshev coefficients of THIS level (its own eigenvalue bound)
  const blockCopy_type *restrict_list; int n_restrict;  // local list restricting this level into the next one
  const blockCopy_type *interp_list;   int n_interp;    // local list (owned by the next level) interpolating into this one
};
struct TailArgs {
  int n;                       // levels in the chain, the last one is only restriction target / interpolation source
  int e_id, R_id, sweeps;
  double a, b;
  TailLevel lv[kTailMaxLevels];
};

enum { SM_CHEBY = 0, SM_GSRB = 1, SM_JACOBI = 2, SM_RESIDUAL = 3 };

// The level being worked on lives in LDS as two dense D^3 arrays in GLOBAL cell order (box boundaries
// disappear; a Dirichlet face is the in-register rule ghost = -centre, exactly what apply_BCs_p1 stores):
//   sx = the iterate (e), st = VECTOR_TEMP (Chebyshev/Jacobi ping-pong partner, then the residual).
// Each lane owns up to kCellsPerLane cells for the whole visit and keeps their coefficients
// (beta faces, alpha, Dinv, rhs) in registers, so a sweep is LDS reads + arithmetic + one barrier
// (~0.2 us) instead of a round trip to L2 (~1.3 us).  Global memory is read once and written once per
// level visit, leaving every vector in the state the per-operator sequence would leave it in.
constexpr int kCellsPerLane = 4;                    // 4096 cells / 1024 lanes
constexpr int kTailMaxCells = kCellsPerLane * kTailThreads;

struct CellRef { int box, ijk; };                    // where a global cell lives in the boxed layout
__device__ __forceinline__ CellRef locate(const hpgmg_hip_level &L, int gi, int gj, int gk) {
  const int bd = L.dim, nb = L.dim_i / bd;
  const int bi = gi / bd, bj = gj / bd, bk = gk / bd;
  CellRef r;
  r.box = bi + nb * (bj + nb * bk);
  r.ijk = (gi - bi * bd) + (gj - bj * bd) * L.jStride + (gk - bk * bd) * L.kStride;
  return r;
}

template <int V>
struct CellCoef { double bi0, bi1, bj0, bj1, bk0, bk1, al, dinv, rhs; };

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

// one visit of a level: load -> (interpolate) -> smooth -> (residual, restrict, zero) -> store
template <int V, int SM>
__device__ void tail_level(const TailArgs &A, int l, int leg, double *sx, double *st) {
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  const TailLevel &T = A.lv[l];
  const hpgmg_hip_level &L = T.L;
  const int D = L.dim_i, total = D * D * D;
  CellCoef<V> q[kCellsPerLane];
  int gi[kCellsPerLane], gj[kCellsPerLane], gk[kCellsPerLane];
  CellRef where[kCellsPerLane];

#pragma unroll
  for (int m = 0; m < kCellsPerLane; m++) {
    const int c = threadIdx.x + m * kTailThreads;
    if (c < total) {
      gi[m] = c % D; gj[m] = (c / D) % D; gk[m] = c / (D * D);
      where[m] = locate(L, gi[m], gj[m], gk[m]);
      const int box = where[m].box, ijk = where[m].ijk, jS = L.jStride, kS = L.kStride;
      double e = vec_origin(L, box, A.e_id)[ijk];
      if (leg == 1) {          // interpolation_vcycle: e = 1.0*e + (coarse parent), interpolation_p0.c:43
        const hpgmg_hip_level &C = A.lv[l + 1].L;
        const CellRef p = locate(C, gi[m] >> 1, gj[m] >> 1, gk[m] >> 1);
        e = 1.0 * e + vec_origin(C, p.box, A.e_id)[p.ijk];
      }
      sx[c] = e;
      st[c] = vec_origin(L, box, VECTOR_TEMP)[ijk];
      q[m].rhs = vec_origin(L, box, A.R_id)[ijk];
      q[m].dinv = vec_origin(L, box, VECTOR_DINV)[ijk];
      q[m].bi0 = q[m].bi1 = q[m].bj0 = q[m].bj1 = q[m].bk0 = q[m].bk1 = q[m].al = 0.0;
      if (kVC) {
        const double *bi = vec_origin(L, box, VECTOR_BETA_I), *bj = vec_origin(L, box, VECTOR_BETA_J), *bk = vec_origin(L, box, VECTOR_BETA_K);
        q[m].bi0 = bi[ijk]; q[m].bi1 = bi[ijk + 1]; q[m].bj0 = bj[ijk]; q[m].bj1 = bj[ijk + jS]; q[m].bk0 = bk[ijk]; q[m].bk1 = bk[ijk + kS];
      }
      if (kHelm) q[m].al = vec_origin(L, box, VECTOR_ALPHA)[ijk];
    }
  }
  __syncthreads();

  // smooth(): chebyshev.c:43-99 / gsrb.c:24-132 / jacobi.c:17-62 (even number of sweeps: the result ends in sx)
  for (int s = 0; s < A.sweeps; s++) {
    const double *src = (SM != SM_GSRB && (s & 1)) ? st : sx;
    double *dst = (SM == SM_GSRB) ? sx : ((s & 1) ? sx : st);
#pragma unroll
    for (int m = 0; m < kCellsPerLane; m++) {
      const int c = threadIdx.x + m * kTailThreads;
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

  if (leg == 0) {                                     // residual -> TEMP (residual.c:42-48)
#pragma unroll
    for (int m = 0; m < kCellsPerLane; m++) {
      const int c = threadIdx.x + m * kTailThreads;
      if (c < total) {
        const double Ax = tail_apply<V>(sx, c, gi[m], gj[m], gk[m], D, q[m], A.a, A.b, T.h2inv);
        st[c] = q[m].rhs - Ax;                        // each lane overwrites only its own TEMP cells: no hazard with the reads of sx
      }
    }
    __syncthreads();
  }

  // leave e and TEMP in global memory as the per-operator sequence would
#pragma unroll
  for (int m = 0; m < kCellsPerLane; m++) {
    const int c = threadIdx.x + m * kTailThreads;
    if (c < total) {
      vec_origin(L, where[m].box, A.e_id)[where[m].ijk] = sx[c];
      vec_origin(L, where[m].box, VECTOR_TEMP)[where[m].ijk] = st[c];
    }
  }

  if (leg == 0) {
    // restriction(next.R <- TEMP): 0.125 * sum of the 8 children in the reference's order (restriction.c:54-57)
    const hpgmg_hip_level &C = A.lv[l + 1].L;
    const int Dc = D / 2, totc = Dc * Dc * Dc;
    for (int c = threadIdx.x; c < totc; c += kTailThreads) {
      const int ci = c % Dc, cj = (c / Dc) % Dc, ck = c / (Dc * Dc);
      const double *f = st + 2 * ci + 2 * cj * D + 2 * ck * D * D;
      double v = f[0] + f[1]; v = v + f[D]; v = v + f[1 + D]; v = v + f[D * D]; v = v + f[1 + D * D]; v = v + f[D + D * D]; v = v + f[1 + D + D * D];
      const CellRef p = locate(C, ci, cj, ck);
      vec_origin(C, p.box, A.R_id)[p.ijk] = v * 0.125;
    }
    // zero_vector(next.e): whole padded boxes, ghosts included (misc.c:6-44)
    const int side = C.dim + 2 * C.ghosts, per_box = side * side * side, all = per_box * C.num_boxes;
    for (int c = threadIdx.x; c < all; c += kTailThreads) {
      const int box = c / per_box, r = c - box * per_box;
      const int k = r / (side * side), j = (r / side) % side, i = r % side;
      (C.box_base[box] + (size_t)A.e_id * (size_t)C.volume)[i + j * C.jStride + k * C.kStride] = 0.0;
    }
  }
  __syncthreads();                                    // global writes of this level are visible to the next level's loads
}

template <int V, int SM>
__global__ __launch_bounds__(kTailThreads) void tail_kernel(const TailArgs A, int leg) {
  __shared__ double sx[kTailMaxCells];
  __shared__ double st[kTailMaxCells];
  if (leg == 0) { for (int l = 0; l + 1 < A.n; l++) tail_level<V, SM>(A, l, 0, sx, st); }
  else          { for (int l = A.n - 2; l >= 0; l--) tail_level<V, SM>(A, l, 1, sx, st); }
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

int hpgmg_hip_tail_max_levels(void) { return kTailMaxLevels; }
int hpgmg_hip_tail_max_cells(void) { return 4096; }

// levels[0..n-1]: finest..coarsest of the chain; per level h2inv, Chebyshev coefficients (sweeps of them),
// the LOCAL restriction list into the next level and the LOCAL interpolation list from the next level.
int hpgmg_hip_vcycle_tail(int n, const hpgmg_hip_level *const *levels, const double *h2inv,
                          const double *c1, const double *c2, int sweeps,
                          const blockCopy_type *const *restrict_lists, const int *n_restrict,
                          const blockCopy_type *const *interp_lists, const int *n_interp,
                          int variant, int smoother, int e_id, int R_id, double a, double b, int leg) {
  HPGMG_SKIP_IF_REPLAY();
  if (n < 2 || n > kTailMaxLevels || sweeps > kTailMaxSweeps) return record_error(hipErrorInvalidValue, "vcycle_tail: chain too long");
  TailArgs A = {};
  A.n = n; A.e_id = e_id; A.R_id = R_id; A.sweeps = sweeps; A.a = a; A.b = b;
  for (int l = 0; l < n; l++) {
    A.lv[l].L = *levels[l];
    A.lv[l].h2inv = h2inv[l];
    for (int s = 0; s < sweeps; s++) { A.lv[l].c1[s] = c1[l * sweeps + s]; A.lv[l].c2[s] = c2[l * sweeps + s]; }
    A.lv[l].restrict_list = restrict_lists[l]; A.lv[l].n_restrict = n_restrict[l];
    A.lv[l].interp_list = interp_lists[l];     A.lv[l].n_interp = n_interp[l];
    if (l + 1 < n && !levels[l]->box_nbr) return record_error(hipErrorInvalidValue, "vcycle_tail: neighbour table missing");
  }
#define TAIL_CASE(V, SM) hipLaunchKernelGGL((tail_kernel<V, SM>), dim3(1), dim3(kTailThreads), 0, g_stream, A, leg)
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

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
// One workgroup of 1024 lanes walks the cells of a level (<= 4096 cells: <= 4 per lane); all
// data stays in that CU's L1/L2.  The arithmetic is the same expression tree as the streaming
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

// one sweep (or the residual) over every interior cell of a level, all lanes of the workgroup
template <int V, int MODE>
__device__ void tail_sweep(const hpgmg_hip_level &L, int xn_id, int xout_id, int rhs_id,
                           double a, double b, double h2inv, double c1, double c2, int sweep) {
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  const int dim = L.dim, jS = L.jStride, kS = L.kStride, last = dim - 1;
  const int per_box = dim * dim * dim, total = per_box * L.num_boxes;
  for (int c = threadIdx.x; c < total; c += kTailThreads) {
    const int box = c / per_box, r = c - box * per_box;
    const int k = r / (dim * dim), j = (r / dim) % dim, i = r % dim;
    if (MODE == SM_GSRB) {
      const int colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ sweep) & 1;
      if (((i ^ j ^ k ^ colour000) & 1) != 0) continue;   // in place: the other colour keeps its value
    }
    const double *x = vec_origin(L, box, xn_id);
    double *out = vec_origin(L, box, xout_id);
    const int ijk = i + j * jS + k * kS;
    const double xc = x[ijk];
    auto outside = [&](int dir, int idx_in_neighbour) -> double {
      const int nb = L.box_nbr[6 * box + dir];
      if (nb >= 0) return vec_origin(L, nb, xn_id)[idx_in_neighbour];
      return -xc;                                          // Dirichlet face (the host guarantees no remote faces)
    };
    const double xim = (i == 0)    ? outside(0, last + j * jS + k * kS) : x[ijk - 1];
    const double xip = (i == last) ? outside(1, j * jS + k * kS)        : x[ijk + 1];
    const double xjm = (j == 0)    ? outside(2, i + last * jS + k * kS) : x[ijk - jS];
    const double xjp = (j == last) ? outside(3, i + k * kS)             : x[ijk + jS];
    const double xkm = (k == 0)    ? outside(4, i + j * jS + last * kS) : x[ijk - kS];
    const double xkp = (k == last) ? outside(5, i + j * jS)             : x[ijk + kS];
    double bi0 = 0, bi1 = 0, bj0 = 0, bj1 = 0, bk0 = 0, bk1 = 0, al = 0;
    if (kVC) {
      const double *beta_i = vec_origin(L, box, VECTOR_BETA_I), *beta_j = vec_origin(L, box, VECTOR_BETA_J), *beta_k = vec_origin(L, box, VECTOR_BETA_K);
      bi0 = beta_i[ijk]; bi1 = beta_i[ijk + 1]; bj0 = beta_j[ijk]; bj1 = beta_j[ijk + jS]; bk0 = beta_k[ijk]; bk1 = beta_k[ijk + kS];
    }
    if (kHelm) al = vec_origin(L, box, VECTOR_ALPHA)[ijk];
    const double Ax = apply_op_7pt<V>(xc, xim, xip, xjm, xjp, xkm, xkp, bi0, bi1, bj0, bj1, bk0, bk1, al, a, b, h2inv);
    if (MODE == SM_RESIDUAL) { out[ijk] = vec_origin(L, box, rhs_id)[ijk] - Ax; continue; }
    const double rhs = vec_origin(L, box, rhs_id)[ijk], dinv = vec_origin(L, box, VECTOR_DINV)[ijk];
    if (MODE == SM_CHEBY)      { const double xnm1 = out[ijk]; out[ijk] = xc + c1 * (xc - xnm1) + c2 * dinv * (rhs - Ax); }
    else if (MODE == SM_GSRB)  { out[ijk] = xc + dinv * (rhs - Ax); }
    else                       { out[ijk] = xc + c2 * dinv * (rhs - Ax); }
  }
  __syncthreads();
}

// smooth(): the sweep schedule of chebyshev.c:43-47, gsrb.c:26-34, jacobi.c:17-20 (ping-pong with VECTOR_TEMP)
template <int V, int SM>
__device__ void tail_smooth(const TailLevel &T, int x_id, int rhs_id, double a, double b, int sweeps) {
  for (int s = 0; s < sweeps; s++) {
    if (SM == SM_GSRB) {
      tail_sweep<V, SM_GSRB>(T.L, x_id, x_id, rhs_id, a, b, T.h2inv, 0.0, 0.0, s);
    } else {
      const int src = (s & 1) ? VECTOR_TEMP : x_id, dst = (s & 1) ? x_id : VECTOR_TEMP;
      if (SM == SM_CHEBY) tail_sweep<V, SM_CHEBY>(T.L, src, dst, rhs_id, a, b, T.h2inv, T.c1[s], T.c2[s], s);
      else                tail_sweep<V, SM_JACOBI>(T.L, src, dst, rhs_id, a, b, T.h2inv, 0.0, 2.0 / 3.0, s);
    }
  }
}

__device__ __forceinline__ double *side_ptr(const hpgmg_hip_level &L, int id, int box, int i, int j, int k) {
  return vec_origin(L, box, id) + i + j * L.jStride + k * L.kStride;
}

// restriction.c:49-58 over a local list: coarse = 0.125 * sum of 8 fine cells
__device__ void tail_restrict_cell(const hpgmg_hip_level &Lc, int id_c, const hpgmg_hip_level &Lf, int id_f, const blockCopy_type *list, int n) {
  for (int e = 0; e < n; e++) {
    const blockCopy_type &E = list[e];
    const double *rp = side_ptr(Lf, id_f, E.read.box, E.read.i, E.read.j, E.read.k);
    double *wp = side_ptr(Lc, id_c, E.write.box, E.write.i, E.write.j, E.write.k);
    const int di = E.dim.i, dj = E.dim.j, cells = di * dj * E.dim.k, rj = Lf.jStride, rk = Lf.kStride;
    for (int t = threadIdx.x; t < cells; t += kTailThreads) {
      const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
      const double *f = rp + 2 * i + 2 * j * rj + 2 * k * rk;
      double v = f[0] + f[1]; v = v + f[rj]; v = v + f[1 + rj]; v = v + f[rk]; v = v + f[1 + rk]; v = v + f[rj + rk]; v = v + f[1 + rj + rk];
      wp[i + j * Lc.jStride + k * Lc.kStride] = v * 0.125;
    }
  }
  __syncthreads();
}

// interpolation_p0.c:43 over a local list: fine = prescale*fine + coarse parent
__device__ void tail_interp_p0(const hpgmg_hip_level &Lf, int id_f, double prescale, const hpgmg_hip_level &Lc, int id_c, const blockCopy_type *list, int n) {
  for (int e = 0; e < n; e++) {
    const blockCopy_type &E = list[e];
    const double *rp = side_ptr(Lc, id_c, E.read.box, E.read.i, E.read.j, E.read.k);
    double *wp = side_ptr(Lf, id_f, E.write.box, E.write.i, E.write.j, E.write.k);
    const int di = 2 * E.dim.i, dj = 2 * E.dim.j, cells = di * dj * 2 * E.dim.k;
    for (int t = threadIdx.x; t < cells; t += kTailThreads) {
      const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
      double *fw = wp + i + j * Lf.jStride + k * Lf.kStride;
      *fw = prescale * (*fw) + rp[(i >> 1) + (j >> 1) * Lc.jStride + (k >> 1) * Lc.kStride];
    }
  }
  __syncthreads();
}

// misc.c:6-44 zero_vector: whole padded boxes, ghosts included
__device__ void tail_zero(const hpgmg_hip_level &L, int id) {
  const int side = L.dim + 2 * L.ghosts, per_box = side * side * side, total = per_box * L.num_boxes;
  for (int c = threadIdx.x; c < total; c += kTailThreads) {
    const int box = c / per_box, r = c - box * per_box;
    const int k = r / (side * side), j = (r / side) % side, i = r % side;
    (L.box_base[box] + (size_t)id * (size_t)L.volume)[i + j * L.jStride + k * L.kStride] = 0.0;
  }
  __syncthreads();
}

template <int V, int SM>
__global__ __launch_bounds__(kTailThreads) void tail_kernel(const TailArgs A, int leg) {
  if (leg == 0) {
    for (int l = 0; l + 1 < A.n; l++) {
      const TailLevel &T = A.lv[l];
      tail_smooth<V, SM>(T, A.e_id, A.R_id, A.a, A.b, A.sweeps);
      tail_sweep<V, SM_RESIDUAL>(T.L, A.e_id, VECTOR_TEMP, A.R_id, A.a, A.b, T.h2inv, 0.0, 0.0, 0);
      tail_restrict_cell(A.lv[l + 1].L, A.R_id, T.L, VECTOR_TEMP, T.restrict_list, T.n_restrict);
      tail_zero(A.lv[l + 1].L, A.e_id);
    }
  } else {
    for (int l = A.n - 2; l >= 0; l--) {
      const TailLevel &T = A.lv[l];
      tail_interp_p0(T.L, A.e_id, 1.0, A.lv[l + 1].L, A.e_id, T.interp_list, T.n_interp);
      tail_smooth<V, SM>(T, A.e_id, A.R_id, A.a, A.b, A.sweeps);
    }
  }
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

// blocks.hip -- executors of the reference's block-list "mini programs" on the GPU.
//
// A list (ghost exchange pack/local/unpack, boundary condition, restriction,
// interpolation) is an array of blockCopy_type built once on the host
// (hpgmg_amd/csrc/host/level.c, mg.c) and mirrored to device memory; ONE launch
// executes the whole list, one workgroup per entry, instead of one OpenMP task
// per entry as in the reference.  Semantics per entry (paths relative to
// finite-volume/source/operators/):
//   copy       blockCopy.c:6-105       increment   blockCopy.c:109-156
//   bc_p1      boundary_fd.c:6-90      restrict    restriction.c:6-94
//   interp p0  interpolation_p0.c:6-46 interp p1   interpolation_p1.c:8-65
// All of these are pure data movement or <= 8-term sums: HBM/L2-bound, no LDS.
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

template <bool kIncrement>
__global__ __launch_bounds__(256) void copy_blocks_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ list, double prescale) {
  const blockCopy_type &e = list[blockIdx.x];
  const Side r = resolve_read(L, id, e), w = resolve_write(L, id, e);
  const int di = e.dim.i, dj = e.dim.j, n = di * dj * e.dim.k;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
    const double v = r.p[i + j * r.jS + k * r.kS];
    double *dst = &w.p[i + j * w.jS + k * w.kS];
    if (kIncrement) *dst = prescale * (*dst) + v; else *dst = v;
  }
}

__global__ __launch_bounds__(256) void bc_p1_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ list) {
  const blockCopy_type &e = list[blockIdx.x];
  const int inward = 26 - e.subtype;                       // direction pointing back into the domain
  const int ni = inward % 3 - 1, nj = (inward % 9) / 3 - 1, nk = inward / 9 - 1;
  const int kind = (ni != 0) + (nj != 0) + (nk != 0);      // 1 face, 2 edge, 3 corner
  const double scale = (kind == 2) ? 1.0 : -1.0;
  const int jS = L.jStride, kS = L.kStride, step = ni + nj * jS + nk * kS;
  double *x = vec_origin(L, e.read.box, id) + e.read.i + e.read.j * jS + e.read.k * kS;
  const int di = e.dim.i, dj = e.dim.j, n = di * dj * e.dim.k;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
    const int ijk = i + j * jS + k * kS;
    x[ijk] = scale * x[ijk + step];
  }
}

// boundary_fd.c:93-205 apply_BCs_p2: quadratic extrapolation through a zero on the boundary face;
// faces 2 terms, edges 4, corners 8 (decimal literals of the reference)
__global__ __launch_bounds__(256) void bc_p2_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ list) {
  const blockCopy_type &e = list[blockIdx.x];
  const int jS = L.jStride, kS = L.kStride, inward = 26 - e.subtype;
  const int di = (inward % 3 - 1) * 1, dj = ((inward % 9) / 3 - 1) * jS, dk = (inward / 9 - 1) * kS;
  const int kind = (di != 0) + (dj != 0) + (dk != 0);
  double *x = vec_origin(L, e.read.box, id) + e.read.i + e.read.j * jS + e.read.k * kS;
  const int ni = e.dim.i, nj = e.dim.j, n = ni * nj * e.dim.k;
  int dr = 0, ds = 0;
  if (di == 0) { dr = dj; ds = dk; }
  if (dj == 0) { dr = di; ds = dk; }
  if (dk == 0) { dr = di; ds = dj; }
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int i = t % ni, j = (t / ni) % nj, k = t / (ni * nj);
    const int ijk = i + j * jS + k * kS;
    double v;
    if (kind == 1) {
      const int s1 = di + dj + dk;
      v = -2.0 * x[ijk + s1] + 0.333333333333333333 * x[ijk + 2 * s1];
    } else if (kind == 2) {
      v = 4.000000000000000000 * x[ijk + dr + ds] - 0.666666666666666667 * x[ijk + 2 * dr + ds];
      v = v - 0.666666666666666667 * x[ijk + dr + 2 * ds];
      v = v + 0.111111111111111111 * x[ijk + 2 * dr + 2 * ds];
    } else {
      v = -8.000000000000000000 * x[ijk + di + dj + dk] + 1.333333333333333333 * x[ijk + 2 * di + dj + dk];
      v = v + 1.333333333333333333 * x[ijk + di + 2 * dj + dk];
      v = v + 1.333333333333333333 * x[ijk + di + dj + 2 * dk];
      v = v - 0.222222222222222222 * x[ijk + 2 * di + 2 * dj + dk];
      v = v - 0.222222222222222222 * x[ijk + di + 2 * dj + 2 * dk];
      v = v - 0.222222222222222222 * x[ijk + 2 * di + dj + 2 * dk];
      v = v + 0.037037037037037037 * x[ijk + 2 * di + 2 * dj + 2 * dk];
    }
    x[ijk] = v;
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
// dimension by dimension (i, then j, then k), one fine cell per lane: the same intermediate values
// f?c??, f??c?, f??? the reference forms for the 8 children of a coarse cell
template <int ORDER>
__global__ __launch_bounds__(256) void interp_tensor_kernel(const hpgmg_hip_level Lf, int id_f, double prescale, const hpgmg_hip_level Lc, int id_c,
                                                            const blockCopy_type *__restrict__ list) {
  constexpr int R = (ORDER == 4) ? 2 : 1, W = 2 * R + 1;
  const blockCopy_type &e = list[blockIdx.x];
  const Side r = resolve_read(Lc, id_c, e), w = resolve_write(Lf, id_f, e);
  const int di = 2 * e.dim.i, dj = 2 * e.dim.j, n = di * dj * 2 * e.dim.k, rj = r.jS, rk = r.kS;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
    const double *c = r.p + (i >> 1) + (j >> 1) * rj + (k >> 1) * rk;
    double line[W], tj[W], tk[W];
#pragma unroll
    for (int kk = 0; kk < W; kk++) {
#pragma unroll
      for (int jj = 0; jj < W; jj++) {
#pragma unroll
        for (int ii = 0; ii < W; ii++) line[ii] = c[(ii - R) + (jj - R) * rj + (kk - R) * rk];
        tj[jj] = interp_rule<ORDER>(i & 1, line);
      }
      tk[kk] = interp_rule<ORDER>(j & 1, tj);
    }
    double *fw = &w.p[i + j * w.jS + k * w.kS];
    *fw = prescale * (*fw) + interp_rule<ORDER>(k & 1, tk);
  }
}

template <int TYPE>
__global__ __launch_bounds__(256) void restrict_blocks_kernel(const hpgmg_hip_level Lc, int id_c, const hpgmg_hip_level Lf, int id_f,
                                                              const blockCopy_type *__restrict__ list) {
  const blockCopy_type &e = list[blockIdx.x];
  const Side r = resolve_read(Lf, id_f, e), w = resolve_write(Lc, id_c, e);
  const int di = e.dim.i, dj = e.dim.j, n = di * dj * e.dim.k, rj = r.jS, rk = r.kS;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
    const double *f = r.p + 2 * i + 2 * j * rj + 2 * k * rk;
    double v;
    if (TYPE == RESTRICT_CELL) {
      v = f[0] + f[1]; v = v + f[rj]; v = v + f[1 + rj]; v = v + f[rk]; v = v + f[1 + rk]; v = v + f[rj + rk]; v = v + f[1 + rj + rk];
      v = v * 0.125;
    } else if (TYPE == RESTRICT_FACE_I) {
      v = f[0] + f[rj]; v = v + f[rk]; v = v + f[rj + rk]; v = v * 0.25;
    } else if (TYPE == RESTRICT_FACE_J) {
      v = f[0] + f[1]; v = v + f[rk]; v = v + f[1 + rk]; v = v * 0.25;
    } else {
      v = f[0] + f[1]; v = v + f[rj]; v = v + f[1 + rj]; v = v * 0.25;
    }
    w.p[i + j * w.jS + k * w.kS] = v;
  }
}

template <int ORDER>
__global__ __launch_bounds__(256) void interp_blocks_kernel(const hpgmg_hip_level Lf, int id_f, double prescale, const hpgmg_hip_level Lc, int id_c,
                                                            const blockCopy_type *__restrict__ list) {
  const blockCopy_type &e = list[blockIdx.x];
  const Side r = resolve_read(Lc, id_c, e), w = resolve_write(Lf, id_f, e);
  const int di = 2 * e.dim.i, dj = 2 * e.dim.j, n = di * dj * 2 * e.dim.k, rj = r.jS, rk = r.kS;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int i = t % di, j = (t / di) % dj, k = t / (di * dj);
    double *fw = &w.p[i + j * w.jS + k * w.kS];
    const double *c = r.p + (i >> 1) + (j >> 1) * rj + (k >> 1) * rk;
    double v = prescale * (*fw);
    if (ORDER == 0) {
      v = v + c[0];
    } else {  // even fine cell leans on the coarse neighbour behind it, odd on the one ahead
      const int oi = (i & 1) ? 1 : -1, oj = (j & 1) ? rj : -rj, ok = (k & 1) ? rk : -rk;
      v = v + 0.421875 * c[0];
      v = v + 0.140625 * c[ok];
      v = v + 0.140625 * c[oj];
      v = v + 0.046875 * c[oj + ok];
      v = v + 0.140625 * c[oi];
      v = v + 0.046875 * c[oi + ok];
      v = v + 0.046875 * c[oi + oj];
      v = v + 0.015625 * c[oi + oj + ok];
    }
    *fw = v;
  }
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

int hpgmg_hip_copy_blocks(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int n) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  hipLaunchKernelGGL((copy_blocks_kernel<false>), dim3(n), dim3(256), 0, g_stream, *L, id, blocks, 0.0);
  HPGMG_LAUNCH_CHECK("copy_blocks_kernel");
  return 0;
}
int hpgmg_hip_increment_blocks(const hpgmg_hip_level *L, int id, double prescale, const blockCopy_type *blocks, int n) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  hipLaunchKernelGGL((copy_blocks_kernel<true>), dim3(n), dim3(256), 0, g_stream, *L, id, blocks, prescale);
  HPGMG_LAUNCH_CHECK("increment_blocks_kernel");
  return 0;
}
int hpgmg_hip_apply_bc_p1(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int n) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bc_p1_kernel, dim3(n), dim3(256), 0, g_stream, *L, id, blocks);
  HPGMG_LAUNCH_CHECK("bc_p1_kernel");
  return 0;
}
int hpgmg_hip_apply_bc_p2(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int n) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bc_p2_kernel, dim3(n), dim3(256), 0, g_stream, *L, id, blocks);
  HPGMG_LAUNCH_CHECK("bc_p2_kernel");
  return 0;
}
int hpgmg_hip_restrict_blocks(const hpgmg_hip_level *Lc, int id_c, const hpgmg_hip_level *Lf, int id_f,
                              const blockCopy_type *blocks, int n, int type) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  switch (type) {
    case RESTRICT_CELL:   hipLaunchKernelGGL((restrict_blocks_kernel<RESTRICT_CELL>), dim3(n), dim3(256), 0, g_stream, *Lc, id_c, *Lf, id_f, blocks); break;
    case RESTRICT_FACE_I: hipLaunchKernelGGL((restrict_blocks_kernel<RESTRICT_FACE_I>), dim3(n), dim3(256), 0, g_stream, *Lc, id_c, *Lf, id_f, blocks); break;
    case RESTRICT_FACE_J: hipLaunchKernelGGL((restrict_blocks_kernel<RESTRICT_FACE_J>), dim3(n), dim3(256), 0, g_stream, *Lc, id_c, *Lf, id_f, blocks); break;
    case RESTRICT_FACE_K: hipLaunchKernelGGL((restrict_blocks_kernel<RESTRICT_FACE_K>), dim3(n), dim3(256), 0, g_stream, *Lc, id_c, *Lf, id_f, blocks); break;
    default: return record_error(hipErrorInvalidValue, "restriction type");
  }
  HPGMG_LAUNCH_CHECK("restrict_blocks_kernel");
  return 0;
}
int hpgmg_hip_interpolate_blocks(const hpgmg_hip_level *Lf, int id_f, double prescale, const hpgmg_hip_level *Lc, int id_c,
                                 const blockCopy_type *blocks, int n, int order) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  if (order == 0) hipLaunchKernelGGL((interp_blocks_kernel<0>), dim3(n), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else if (order == 1) hipLaunchKernelGGL((interp_blocks_kernel<1>), dim3(n), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else if (order == 2) hipLaunchKernelGGL((interp_tensor_kernel<2>), dim3(n), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else if (order == 3) hipLaunchKernelGGL((interp_tensor_kernel<3>), dim3(n), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else if (order == 4) hipLaunchKernelGGL((interp_tensor_kernel<4>), dim3(n), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else return record_error(hipErrorInvalidValue, "interpolation order");
  HPGMG_LAUNCH_CHECK("interp_blocks_kernel");
  return 0;
}

}  // extern "C"

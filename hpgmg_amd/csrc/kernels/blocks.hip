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
  else return record_error(hipErrorInvalidValue, "interpolation order");
  HPGMG_LAUNCH_CHECK("interp_blocks_kernel");
  return 0;
}

}  // extern "C"

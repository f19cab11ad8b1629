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
#include "block_ops.hpp"

namespace hpgmg {

template <bool kIncrement>
__global__ __launch_bounds__(256) void copy_blocks_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ list, double prescale) {
  copy_entry<kIncrement>(L, id, list[blockIdx.x], prescale, (int)threadIdx.x, (int)blockDim.x);
}
__global__ __launch_bounds__(256) void bc_p1_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ list) {
  bc_p1_entry(L, id, list[blockIdx.x], (int)threadIdx.x, (int)blockDim.x);
}
__global__ __launch_bounds__(256) void bc_p2_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ list) {
  bc_p2_entry(L, id, list[blockIdx.x], (int)threadIdx.x, (int)blockDim.x);
}
__global__ __launch_bounds__(256) void bc_v2_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ list) {
  if (L.ghosts > 1) { bc_zero_entry(L, id, list[blockIdx.x], (int)threadIdx.x, (int)blockDim.x); __syncthreads(); }
  bc_v2_entry(L, id, list[blockIdx.x], (int)threadIdx.x, (int)blockDim.x);
}
__global__ __launch_bounds__(256) void bc_v4_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ list) {
  if (L.ghosts > 2) { bc_zero_entry(L, id, list[blockIdx.x], (int)threadIdx.x, (int)blockDim.x); __syncthreads(); }
  bc_v4_entry(L, id, list[blockIdx.x], (int)threadIdx.x, (int)blockDim.x);
}
template <int ORDER, bool CLEAR>
__global__ __launch_bounds__(256) void bc_fv_kernel(const hpgmg_hip_level L, int id, const hpgmg_hip_bc_entry *__restrict__ list) {
  bc_fv_compact_entry<ORDER, false, CLEAR>(L, id, list[blockIdx.x], (int)threadIdx.x, 256);
}
// exchange_boundary's box-to-box copies and the boundary conditions of the same vector in one launch: the first n_copy workgroups run
// a copy entry each, the others a condition entry.  The two touch disjoint ghost cells and read interior cells only (the host has
// redirected the conditions that run along another box's face to that box, see hpgmg_hip_bc_entry), so no order is needed between them.
template <int ORDER, bool CLEAR>
__global__ __launch_bounds__(256) void ghost_fill_kernel(const hpgmg_hip_level L, int id, const blockCopy_type *__restrict__ copies, int n_copy,
                                                         const hpgmg_hip_bc_entry *__restrict__ list) {
  if ((int)blockIdx.x < n_copy) copy_entry<false>(L, id, copies[blockIdx.x], 0.0, (int)threadIdx.x, 256);
  else bc_fv_compact_entry<ORDER, true, CLEAR>(L, id, list[(int)blockIdx.x - n_copy], (int)threadIdx.x, 256);
}

// boundary_fv.c:573-681 extrapolate_betas.  The reference updates each block IN PLACE in k,j,i order, so a
// deeper ghost cell may see a shallower one already (high side) or not yet (low side) updated.  One lane per
// list entry walks its block in exactly that order; entries run in parallel like the reference's OpenMP tasks.
// Setup only (called from rebuild_operator), never in the timed cycle.
// Blocks of one box depend on each other (the deeper layer of an edge block reads the shallower layer of a face block), and the reference
// with one thread runs them in list order: per box the 26 directions in the order of level.c:377-379.  The launcher therefore runs one
// launch per direction `phase` (blocks of the same direction never read each other's cells), in that order.
__global__ __launch_bounds__(64) void extrapolate_betas_kernel(const hpgmg_hip_level L, const blockCopy_type *__restrict__ list, int n, int phase) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= n) return;
  const blockCopy_type &e = list[b];
  const int jS = L.jStride, kS = L.kStride, ilo = e.read.i, jlo = e.read.j, klo = e.read.k;
  int subtype = 13;
  if (ilo < 0) subtype -= 1;
  if (jlo < 0) subtype -= 3;
  if (klo < 0) subtype -= 9;
  if (ilo >= L.dim) subtype += 1;
  if (jlo >= L.dim) subtype += 3;
  if (klo >= L.dim) subtype += 9;
  if (subtype != phase) return;
  const int normal = 26 - subtype, di = normal % 3 - 1, dj = (normal % 9) / 3 - 1, dk = normal / 9 - 1;
  const int bs[3] = { dj * jS + dk * kS, di + dk * kS, di + dj * jS };
  const int skip_lo[3] = {12, 10, 4}, skip_hi[3] = {14, 16, 22};
  double *beta[3] = { vec_origin(L, e.read.box, VECTOR_BETA_I), vec_origin(L, e.read.box, VECTOR_BETA_J), vec_origin(L, e.read.box, VECTOR_BETA_K) };
  for (int k = 0; k < e.dim.k; k++) for (int j = 0; j < e.dim.j; j++) for (int i = 0; i < e.dim.i; i++) {
    const int ijk = (i + ilo) + (j + jlo) * jS + (k + klo) * kS;
    for (int c = 0; c < 3; c++) {
      if (subtype == skip_lo[c] || subtype == skip_hi[c]) continue;
      double *bb = beta[c]; const int st = bs[c];
      if (L.dim >= 5) {
        double v = 5.0 * bb[ijk + st] - 10.0 * bb[ijk + 2 * st]; v = v + 10.0 * bb[ijk + 3 * st]; v = v - 5.0 * bb[ijk + 4 * st]; v = v + bb[ijk + 5 * st];
        bb[ijk] = v;
      } else if (L.dim >= 4) {
        double v = 4.0 * bb[ijk + st] - 6.0 * bb[ijk + 2 * st]; v = v + 4.0 * bb[ijk + 3 * st]; v = v - bb[ijk + 4 * st];
        bb[ijk] = v;
      } else if (L.dim >= 2) {
        bb[ijk] = 2.0 * bb[ijk + st] - bb[ijk + 2 * st];
      }
    }
  }
}

// The two children (2 ci, 2 ci + 1) of a coarse cell as ONE 16-byte store to device memory.  (Written as `*(double2 *)fw = ...` next to the scalar form of the other
// branch, the compiler merged the two branches into two 8-byte stores per lane: every fine line was written in two half-filled passes -- the interpolations ran at
// half the device's write rate.  A vector store through a pointer typed as device memory is not something it takes apart.)
typedef double __attribute__((ext_vector_type(2))) pair_v;
template <bool NT>      // NT: non-temporal -- for a vector written once that is larger than the infinity cache (512^3 onto zeros: 478 -> 378 us; read back soon or smaller: slower)
__device__ __forceinline__ void store_pair(double *p, double a, double b) {
  pair_v w; w.x = a; w.y = b;
  if (NT) __builtin_nontemporal_store(w, (pair_v __attribute__((address_space(1))) *)as_global(p));
  else *(pair_v __attribute__((address_space(1))) *)as_global(p) = w;
}

// Tensor-product interpolations (interpolation_p2.c, _v2.c, _v4.c): the 1-D rule is applied along i, then j, then k.
// One lane per COARSE column, marching in k over a chunk of the entry (a wave per coarse row and chunk; grid.y strides over them):
// the i- and j-passes of one coarse plane depend only on that plane -- their four results (child i parity x child j parity) are
// computed once per plane from its (2R+1)^2 neighbourhood and kept in a sliding window of 2R+1 planes, the k-pass then serves the
// cell's 8 children.  (2R+1)^2 loads per coarse cell instead of (2R+1)^3 (the first version was bound by the L1 at 125 loads per
// cell); each child is the same expression tree rule_k(rule_j(rule_i(coarse))) as in the reference.  Children are written as
// 16-byte pairs when the layout allows.
template <int ORDER, bool ZEROED = false, bool NT = false>      // ZEROED: as in interp_blocks_kernel -- the fine vector counts as +0.0 (prescale == 0), never read; NT: store_pair
__global__ __launch_bounds__(256) void interp_tensor_kernel(const hpgmg_hip_level Lf, int id_f, double prescale, const hpgmg_hip_level Lc, int id_c,
                                                            const blockCopy_type *__restrict__ list) {
  constexpr int R = (ORDER == 4) ? 2 : 1, W = 2 * R + 1, KC = 8;
  const blockCopy_type &e = list[blockIdx.x];
  const Side r = resolve_read(Lc, id_c, e), w = resolve_write(Lf, id_f, e);
  const int ci_n = e.dim.i, cj_n = e.dim.j, ck_n = e.dim.k, rj = r.jS, rk = r.kS;
  // a coarse row shorter than a wave (boxes of 32^3 and smaller below the fine level): the wave takes 64 / wci rows of the tile at once, wci lanes each
  const int wci = (ci_n > 32) ? 64 : (ci_n > 16 ? 32 : (ci_n > 8 ? 16 : 8)), rpw = 64 / wci, cjg = (cj_n + rpw - 1) / rpw;
  const int nkc = (ck_n + KC - 1) / KC, units = cjg * nkc;
  const bool pairs = (e.write.box >= 0) && (Lf.flags & 1);
  const int lane = threadIdx.x % 64, lane_i = lane % wci, lane_row = lane / wci;
  for (int u = blockIdx.y * 4 + threadIdx.x / 64; u < units; u += gridDim.y * 4) {
    const int kc = u / cjg, cj = (u - kc * cjg) * rpw + lane_row, k0 = kc * KC, k1 = (k0 + KC < ck_n) ? k0 + KC : ck_n;
    if (cj >= cj_n) continue;
    for (int ci = lane_i; ci < ci_n; ci += wci) {
      const double *c = r.p + ci + cj * rj;
      double tk[2][2][W];                                       // [child i parity][child j parity][coarse plane ck-R .. ck+R]
      auto plane = [&](int ck, int slot) {                      // i-pass and j-pass of coarse plane ck
        const double *cp = c + ck * rk;
        double tj[2][W];                                        // [child i parity][coarse row]
#pragma unroll
        for (int jj = 0; jj < W; jj++) {
          double line[W];
#pragma unroll
          for (int ii = 0; ii < W; ii++) line[ii] = cp[(ii - R) + (jj - R) * rj];
          tj[0][jj] = interp_rule<ORDER>(false, line);
          tj[1][jj] = interp_rule<ORDER>(true, line);
        }
#pragma unroll
        for (int fi = 0; fi < 2; fi++) { tk[fi][0][slot] = interp_rule<ORDER>(false, tj[fi]); tk[fi][1][slot] = interp_rule<ORDER>(true, tj[fi]); }
      };
#pragma unroll
      for (int kk = 0; kk < W - 1; kk++) plane(k0 - R + kk, kk + 1);   // planes k0-R .. k0+R-1 wait one slot up: the loop shifts first
      for (int ck = k0; ck < k1; ck++) {
#pragma unroll
        for (int fi = 0; fi < 2; fi++)
#pragma unroll
          for (int fj = 0; fj < 2; fj++)
#pragma unroll
            for (int kk = 0; kk < W - 1; kk++) tk[fi][fj][kk] = tk[fi][fj][kk + 1];
        plane(ck + R, W - 1);
        double *frow = w.p + 2 * cj * w.jS + 2 * ck * w.kS;
#pragma unroll
        for (int fk = 0; fk < 2; fk++) {
#pragma unroll
          for (int fj = 0; fj < 2; fj++) {
            double *fw = frow + 2 * ci + fj * w.jS + fk * w.kS;
            const double a0 = interp_rule<ORDER>(fk != 0, tk[0][fj]), a1 = interp_rule<ORDER>(fk != 0, tk[1][fj]);
            double f0 = 0.0, f1 = 0.0;
            auto load_fine = [&]() {
              if (pairs) { const double2 t = *reinterpret_cast<const double2 *>(fw); f0 = t.x; f1 = t.y; }
              else { f0 = fw[0]; f1 = fw[1]; }
            };
            // prescale == 0 (interpolation_fcycle): 0 * old + y == y for every finite old value unless y is a zero, whose sign then
            // follows old's -- the old value is fetched only for results that are zeros
            if (prescale != 0.0) load_fine();
            double v0 = prescale * f0 + a0, v1 = prescale * f1 + a1;
            if (!ZEROED && prescale == 0.0 && (v0 == 0.0 || v1 == 0.0)) { load_fine(); v0 = prescale * f0 + a0; v1 = prescale * f1 + a1; }
            if (pairs) store_pair<NT>(fw, v0, v1);
            else { fw[0] = v0; fw[1] = v1; }
          }
        }
      }
    }
  }
}

// restriction.c:49-91 for one list entry (one workgroup)
template <int TYPE>
__device__ __forceinline__ void restrict_entry(const hpgmg_hip_level &Lc, int id_c, const hpgmg_hip_level &Lf, int id_f, const blockCopy_type &e) {

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
template <int TYPE>
__global__ __launch_bounds__(256) void restrict_blocks_kernel(const hpgmg_hip_level Lc, int id_c, const hpgmg_hip_level Lf, int id_f,
                                                              const blockCopy_type *__restrict__ list) {
  restrict_entry<TYPE>(Lc, id_c, Lf, id_f, list[blockIdx.x]);
}
// restriction of a list + zero_vector(coarse, zero_id) in one launch (the down-leg of MGVCycle does both, mg.c:1152-1153):
// workgroups [0, n) take the list entries, the rest clear the coarse vector -- whole padded boxes, ghosts included
// (misc.c:6-44; the alignment padding between rows is never read and already zero, so each box is one contiguous run)
__global__ __launch_bounds__(256) void restrict_cell_zero_kernel(const hpgmg_hip_level Lc, int id_c, const hpgmg_hip_level Lf, int id_f,
                                                                 const blockCopy_type *__restrict__ list, int n, int zero_id, int chunks_per_box) {
  if ((int)blockIdx.x < n) { restrict_entry<RESTRICT_CELL>(Lc, id_c, Lf, id_f, list[blockIdx.x]); return; }
  const int z = (int)blockIdx.x - n, box = z / chunks_per_box, chunk = z - box * chunks_per_box;
  if (box >= Lc.num_boxes) return;
  double *v = Lc.box_base[box] + (size_t)zero_id * (size_t)Lc.volume;
  const int lo = chunk * 4096, hi = (lo + 4096 < Lc.volume) ? lo + 4096 : Lc.volume;
  for (int t = lo + (int)threadIdx.x; t < hi; t += 256) v[t] = 0.0;
}

// interpolation_p0.c:43 (ORDER 0, piecewise constant) and interpolation_p1.c:40-70 (ORDER 1, trilinear).
// One wave per COARSE row of an entry (grid.y strides over the rows), one coarse cell per lane: the lane
// produces the cell's 2x2x2 children, each fine row as one 16-byte read-modify-write when the layout allows.
// The per-child expression (prescale*fine + weighted coarse neighbours, in the reference's order) is unchanged.
// ZEROED (with prescale == 0): the fine vector counts as holding +0.0 everywhere -- zero_vector() followed by interpolation_fcycle() with the fine
// vector neither zeroed nor read (0.0 * 0.0 + y).
template <int ORDER, bool ZEROED = false, bool NT = false>
__global__ __launch_bounds__(256) void interp_blocks_kernel(const hpgmg_hip_level Lf, int id_f, double prescale, const hpgmg_hip_level Lc, int id_c,
                                                            const blockCopy_type *__restrict__ list) {
  const blockCopy_type &e = list[blockIdx.x];
  const Side r = resolve_read(Lc, id_c, e), w = resolve_write(Lf, id_f, e);
  const int ci_n = e.dim.i, cj_n = e.dim.j, rows = cj_n * e.dim.k, rj = r.jS, rk = r.kS;
  const bool pairs = (e.write.box >= 0) && (Lf.flags & 1);      // fine pairs (2ci, 2ci+1) are 16-byte aligned
  const int lane = threadIdx.x % 64;
  for (int row = blockIdx.y * 4 + threadIdx.x / 64; row < rows; row += gridDim.y * 4) {
    const int ck = row / cj_n, cj = row - ck * cj_n;
    const double *crow = r.p + cj * rj + ck * rk;
    double *frow = w.p + 2 * cj * w.jS + 2 * ck * w.kS;
    for (int ci = lane; ci < ci_n; ci += 64) {
      const double *c = crow + ci;
      double nb[3][3][3];                                       // [dk][dj][di], offsets -1,0,+1 (ORDER 1 only)
      if (ORDER == 1) {
#pragma unroll
        for (int dk = 0; dk < 3; dk++)
#pragma unroll
          for (int dj = 0; dj < 3; dj++)
#pragma unroll
            for (int di = 0; di < 3; di++) nb[dk][dj][di] = c[(di - 1) + (dj - 1) * rj + (dk - 1) * rk];
      } else nb[1][1][1] = c[0];
#pragma unroll
      for (int fk = 0; fk < 2; fk++) {
#pragma unroll
        for (int fj = 0; fj < 2; fj++) {
          double *fw = frow + 2 * ci + fj * w.jS + fk * w.kS;
          double v[2];
          auto blend = [&](double f0, double f1) {
            v[0] = prescale * f0; v[1] = prescale * f1;
#pragma unroll
            for (int fi = 0; fi < 2; fi++) {
              if (ORDER == 0) v[fi] = v[fi] + nb[1][1][1];
              else {  // an even fine cell leans on the coarse neighbour behind it, an odd one on the one ahead
                const int oi = fi ? 2 : 0, oj = fj ? 2 : 0, ok = fk ? 2 : 0;
                v[fi] = v[fi] + 0.421875 * nb[1][1][1];
                v[fi] = v[fi] + 0.140625 * nb[ok][1][1];
                v[fi] = v[fi] + 0.140625 * nb[1][oj][1];
                v[fi] = v[fi] + 0.046875 * nb[ok][oj][1];
                v[fi] = v[fi] + 0.140625 * nb[1][1][oi];
                v[fi] = v[fi] + 0.046875 * nb[ok][1][oi];
                v[fi] = v[fi] + 0.046875 * nb[1][oj][oi];
                v[fi] = v[fi] + 0.015625 * nb[ok][oj][oi];
              }
            }
          };
          auto load_fine = [&](double &f0, double &f1) {
            if (pairs) { const double2 t = *reinterpret_cast<const double2 *>(fw); f0 = t.x; f1 = t.y; }
            else { f0 = fw[0]; f1 = fw[1]; }
          };
          // prescale == 0 (interpolation_fcycle, and the send buffers): 0 * old + y == y for every finite old value unless y is a
          // zero, whose sign then follows old's -- so the old value is fetched only for results that are zeros
          double f0 = 0.0, f1 = 0.0;
          if (prescale != 0.0) load_fine(f0, f1);
          blend(f0, f1);
          if (!ZEROED && prescale == 0.0 && (v[0] == 0.0 || v[1] == 0.0)) { load_fine(f0, f1); blend(f0, f1); }
          if (pairs) store_pair<NT>(fw, v[0], v[1]);
          else { fw[0] = v[0]; fw[1] = v[1]; }
        }
      }
    }
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
int hpgmg_hip_apply_bc_v2(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int n) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bc_v2_kernel, dim3(n), dim3(256), 0, g_stream, *L, id, blocks);
  HPGMG_LAUNCH_CHECK("bc_v2_kernel");
  return 0;
}
int hpgmg_hip_apply_bc_v4(const hpgmg_hip_level *L, int id, const blockCopy_type *blocks, int n) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bc_v4_kernel, dim3(n), dim3(256), 0, g_stream, *L, id, blocks);
  HPGMG_LAUNCH_CHECK("bc_v4_kernel");
  return 0;
}
// 0: launch without clearing, 1: with, -1: not supported (p2 is defined for one ghost layer only)
static int bc_fv_clear_mode(const hpgmg_hip_level *L, int order) {
  if (order != 1 && order != 2 && order != 4 && order != 12) return -1;
  const int fills = (order == 4) ? 2 : 1;
  if (L->ghosts <= fills) return 0;
  return (order == 12 || order == 1) ? -1 : 1;
}
int hpgmg_hip_apply_bc_fv(const hpgmg_hip_level *L, int id, const hpgmg_hip_bc_entry *entries, int n, int order) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  const int clear = bc_fv_clear_mode(L, order);
  if (clear < 0) return record_error(hipErrorInvalidValue, "apply_bc_fv: order / ghost depth not supported");
  if (order == 4 && clear)      hipLaunchKernelGGL((bc_fv_kernel<4, true>), dim3(n), dim3(256), 0, g_stream, *L, id, entries);
  else if (order == 4)          hipLaunchKernelGGL((bc_fv_kernel<4, false>), dim3(n), dim3(256), 0, g_stream, *L, id, entries);
  else if (order == 2 && clear) hipLaunchKernelGGL((bc_fv_kernel<2, true>), dim3(n), dim3(256), 0, g_stream, *L, id, entries);
  else if (order == 2)          hipLaunchKernelGGL((bc_fv_kernel<2, false>), dim3(n), dim3(256), 0, g_stream, *L, id, entries);
  else if (order == 1)          hipLaunchKernelGGL((bc_fv_kernel<1, false>), dim3(n), dim3(256), 0, g_stream, *L, id, entries);
  else                          hipLaunchKernelGGL((bc_fv_kernel<12, false>), dim3(n), dim3(256), 0, g_stream, *L, id, entries);
  HPGMG_LAUNCH_CHECK("bc_fv_kernel");
  return 0;
}
int hpgmg_hip_exchange_and_bc(const hpgmg_hip_level *L, int id, const blockCopy_type *copies, int n_copy, const hpgmg_hip_bc_entry *entries, int n, int order) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return hpgmg_hip_copy_blocks(L, id, copies, n_copy);
  const int clear = bc_fv_clear_mode(L, order);
  if (clear < 0) return record_error(hipErrorInvalidValue, "exchange_and_bc: order / ghost depth not supported");
  const dim3 grid(n_copy + n);
  if (order == 4 && clear)      hipLaunchKernelGGL((ghost_fill_kernel<4, true>), grid, dim3(256), 0, g_stream, *L, id, copies, n_copy, entries);
  else if (order == 4)          hipLaunchKernelGGL((ghost_fill_kernel<4, false>), grid, dim3(256), 0, g_stream, *L, id, copies, n_copy, entries);
  else if (order == 2 && clear) hipLaunchKernelGGL((ghost_fill_kernel<2, true>), grid, dim3(256), 0, g_stream, *L, id, copies, n_copy, entries);
  else if (order == 2)          hipLaunchKernelGGL((ghost_fill_kernel<2, false>), grid, dim3(256), 0, g_stream, *L, id, copies, n_copy, entries);
  else if (order == 1)          hipLaunchKernelGGL((ghost_fill_kernel<1, false>), grid, dim3(256), 0, g_stream, *L, id, copies, n_copy, entries);
  else                          hipLaunchKernelGGL((ghost_fill_kernel<12, false>), grid, dim3(256), 0, g_stream, *L, id, copies, n_copy, entries);
  HPGMG_LAUNCH_CHECK("ghost_fill_kernel");
  return 0;
}
int hpgmg_hip_extrapolate_betas(const hpgmg_hip_level *L, const blockCopy_type *blocks, int n) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  for (int phase = 0; phase < 27; phase++)
    if (phase != 13) hipLaunchKernelGGL(extrapolate_betas_kernel, dim3((n + 63) / 64), dim3(64), 0, g_stream, *L, blocks, n, phase);
  HPGMG_LAUNCH_CHECK("extrapolate_betas_kernel");
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
int hpgmg_hip_restrict_cell_and_zero(const hpgmg_hip_level *Lc, int id_c, const hpgmg_hip_level *Lf, int id_f,
                                     const blockCopy_type *blocks, int n, int zero_id) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0 || Lc->num_boxes <= 0) return record_error(hipErrorInvalidValue, "restrict_cell_and_zero: empty list or level");
  const int chunks = (Lc->volume + 4095) / 4096;
  hipLaunchKernelGGL(restrict_cell_zero_kernel, dim3(n + chunks * Lc->num_boxes), dim3(256), 0, g_stream, *Lc, id_c, *Lf, id_f, blocks, n, zero_id, chunks);
  HPGMG_LAUNCH_CHECK("restrict_cell_zero_kernel");
  return 0;
}
int hpgmg_hip_interpolate_blocks(const hpgmg_hip_level *Lf, int id_f, double prescale, const hpgmg_hip_level *Lc, int id_c,
                                 const blockCopy_type *blocks, int n, int order) {
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  // entries are whole coarse tiles; spread each over enough workgroups to fill the chip (4 coarse rows per workgroup pass)
  const int slabs = n >= 4096 ? 1 : (4096 / n > 64 ? 64 : 4096 / n);
  if (order == 0) hipLaunchKernelGGL((interp_blocks_kernel<0>), dim3(n, slabs), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else if (order == 1) hipLaunchKernelGGL((interp_blocks_kernel<1>), dim3(n, slabs), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else if (order >= 17 && order <= 20) {      // order - 16 onto a fine vector that counts as zeroed (zero_vector + interpolation_fcycle, the fine level touched once)
    if (prescale != 0.0) return record_error(hipErrorInvalidValue, "interpolation onto a zeroed vector: prescale 0");
    // a fine vector beyond the infinity cache (256 MB) is written past the caches: nothing of it would still be there when the next kernel reads it
    static const bool nt_allowed = [] { const char *e = getenv("HPGMG_TUNE_INTERP_NT"); return !(e && *e == '0'); }();
    const bool nt = nt_allowed && (double)Lf->num_boxes * Lf->dim * Lf->dim * Lf->dim * sizeof(double) > 256e6;
#define INTERP_ZEROED(KERNEL, ORD) do { if (nt) hipLaunchKernelGGL((KERNEL<ORD, true, true>), dim3(n, slabs), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks); \
                                        else hipLaunchKernelGGL((KERNEL<ORD, true, false>), dim3(n, slabs), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks); } while (0)
    if (order == 17)      INTERP_ZEROED(interp_blocks_kernel, 1);
    else if (order == 18) INTERP_ZEROED(interp_tensor_kernel, 2);
    else if (order == 19) INTERP_ZEROED(interp_tensor_kernel, 3);
    else                  INTERP_ZEROED(interp_tensor_kernel, 4);
#undef INTERP_ZEROED
  }
  else if (order == 2) hipLaunchKernelGGL((interp_tensor_kernel<2>), dim3(n, slabs), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else if (order == 3) hipLaunchKernelGGL((interp_tensor_kernel<3>), dim3(n, slabs), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else if (order == 4) hipLaunchKernelGGL((interp_tensor_kernel<4>), dim3(n, slabs), dim3(256), 0, g_stream, *Lf, id_f, prescale, *Lc, id_c, blocks);
  else return record_error(hipErrorInvalidValue, "interpolation order");
  HPGMG_LAUNCH_CHECK("interp_blocks_kernel");
  return 0;
}

}  // extern "C"

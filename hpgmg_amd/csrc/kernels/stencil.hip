// stencil.hip -- the hot path: Chebyshev / GSRB / Jacobi smoother sweeps and the
// residual / apply_op stencil of the 7-point operator, written for gfx950.
//
// Reference semantics (paths relative to finite-volume/source/):
//   apply_op_ijk      operators.7pt.c:49-89 (VC Helmholtz :51-62, VC Poisson :64-74, CC :77-88)
//   Chebyshev update  operators/chebyshev.c:86-95      GSRB update operators/gsrb.c:90-105
//   Jacobi update     operators/jacobi.c:53-60         residual    operators/residual.c:42-48
// Every floating-point expression keeps the reference's association and the
// file is compiled with -ffp-contract=off, so results are bit-identical to the
// reference's gcc -O2 build (no FMA, no reassociation): see DESIGN.md "parity".
//
// Bound: HBM bandwidth (about 25 flop per 64-72 B per cell).  Design:
//   * one lane per (i,j) column, 64 lanes along the unit-stride i direction so a
//     wave reads 512 contiguous bytes per stream; the block marches in +k and
//     keeps x[k-1], x[k], x[k+1] and beta_k[k], beta_k[k+1] in registers, so each
//     of the 8-9 streams is read from HBM once per sweep;
//   * the +-i / +-j neighbours of x are re-read through the vector L1 (they are
//     the same 128-B lines the neighbouring lanes / waves of the workgroup
//     fetch as their own centre values);
//   * logical tiles are ordered box, k, j, i and dealt to XCDs in contiguous
//     ranges (common.hpp) so halo planes shared by adjacent tiles hit the same L2.
#include "common.hpp"
#include "stencil_math.hpp"

namespace hpgmg {

enum { MODE_CHEBY = 0, MODE_GSRB = 1, MODE_JACOBI = 2, MODE_RESIDUAL = 3, MODE_APPLY = 4 };

struct StencilArgs {
  int xn_id, xout_id, rhs_id;   // xout = x_np1 (smoothers) or res/Ax
  double a, b, h2inv;
  double c1, c2;                // Chebyshev; Jacobi uses c2 = weight
  int sweep;                    // GSRB colour of this half sweep
  int copy_other_colour;        // GSRB out of place
  int tiles_i, tiles_j, chunks_k, kchunk, per_xcd, total_blocks;
  int ghost_free;               // read face neighbours from the adjacent box / apply the Dirichlet BC in registers
};

// x may alias the output only for in-place GSRB; everywhere else it is restrict-qualified
// so the loads of plane k+1 can be issued ahead of the store of plane k.
template <bool kMayAlias> struct src_ptr { typedef const double *__restrict__ type; };
template <> struct src_ptr<true> { typedef const double *type; };

template <int V, int MODE>
__global__ __launch_bounds__(256) void stencil7_kernel(const hpgmg_hip_level L, const StencilArgs P) {
  const int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (logical >= P.total_blocks) return;
  // logical id -> (box, k chunk, j tile, i tile), i fastest
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;

  const int i = ti * (int)blockDim.x + (int)threadIdx.x;
  const int j = tj * (int)blockDim.y + (int)threadIdx.y;
  if (i >= L.dim || j >= L.dim) return;
  const int k0 = ck * P.kchunk;
  const int k1 = (k0 + P.kchunk < L.dim) ? k0 + P.kchunk : L.dim;
  const int jS = L.jStride, kS = L.kStride;

  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  constexpr bool kSmooth = (MODE == MODE_CHEBY || MODE == MODE_GSRB || MODE == MODE_JACOBI);

  typename src_ptr<MODE == MODE_GSRB>::type x = vec_origin(L, box, P.xn_id);
  double *out = vec_origin(L, box, P.xout_id);   // may alias x (in-place GSRB): no __restrict__
  const double *__restrict__ rhs = (MODE == MODE_APPLY) ? nullptr : vec_origin(L, box, P.rhs_id);
  const double *__restrict__ dinv = kSmooth ? vec_origin(L, box, VECTOR_DINV) : nullptr;
  const double *__restrict__ alpha = kHelm ? vec_origin(L, box, VECTOR_ALPHA) : nullptr;
  const double *__restrict__ beta_i = kVC ? vec_origin(L, box, VECTOR_BETA_I) : nullptr;
  const double *__restrict__ beta_j = kVC ? vec_origin(L, box, VECTOR_BETA_J) : nullptr;
  const double *__restrict__ beta_k = kVC ? vec_origin(L, box, VECTOR_BETA_K) : nullptr;

  int colour000 = 0;
  if (MODE == MODE_GSRB) colour000 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;

  // Value of x just outside this box across face `dir` (0..5 = -i,+i,-j,+j,-k,+k): from the
  // neighbouring local box, from the Dirichlet condition (ghost = -centre), or from the ghost zone.
  const int last = L.dim - 1;
  auto outside = [&](int dir, int idx_in_neighbour, int idx_ghost, double centre) -> double {
    const int nb = L.box_nbr[6 * box + dir];
    if (nb >= 0) return vec_origin(L, nb, P.xn_id)[idx_in_neighbour];
    if (nb == -1) return -centre;
    return x[idx_ghost];
  };
  const bool gf = P.ghost_free != 0;

  int ijk = i + j * jS + k0 * kS;
  double xc = x[ijk];
  double xm = (gf && k0 == 0) ? outside(4, i + j * jS + last * kS, ijk - kS, xc) : x[ijk - kS];
  double bk0 = kVC ? beta_k[ijk] : 0.0;

  for (int k = k0; k < k1; k++, ijk += kS) {
    const double xp = (gf && k == last) ? outside(5, i + j * jS, ijk + kS, xc) : x[ijk + kS];
    const double bk1 = kVC ? beta_k[ijk + kS] : 0.0;
    bool update = true;
    if (MODE == MODE_GSRB) update = (((i ^ j ^ k ^ colour000) & 1) == 0);
    if (MODE != MODE_GSRB || update) {
      const double xim = (gf && i == 0)    ? outside(0, last + j * jS + k * kS, ijk - 1, xc)  : x[ijk - 1];
      const double xip = (gf && i == last) ? outside(1, j * jS + k * kS, ijk + 1, xc)          : x[ijk + 1];
      const double xjm = (gf && j == 0)    ? outside(2, i + last * jS + k * kS, ijk - jS, xc) : x[ijk - jS];
      const double xjp = (gf && j == last) ? outside(3, i + k * kS, ijk + jS, xc)              : x[ijk + jS];
      double bi0 = 0.0, bi1 = 0.0, bj0 = 0.0, bj1 = 0.0, al = 0.0;
      if (kVC) { bi0 = beta_i[ijk]; bi1 = beta_i[ijk + 1]; bj0 = beta_j[ijk]; bj1 = beta_j[ijk + jS]; }
      if (kHelm) al = alpha[ijk];
      const double Ax = apply_op_7pt<V>(xc, xim, xip, xjm, xjp, xm, xp, bi0, bi1, bj0, bj1, bk0, bk1, al, P.a, P.b, P.h2inv);
      if (MODE == MODE_CHEBY) {
        const double xnm1 = out[ijk];
        out[ijk] = xc + P.c1 * (xc - xnm1) + P.c2 * dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_GSRB) {
        out[ijk] = xc + dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_JACOBI) {
        out[ijk] = xc + P.c2 * dinv[ijk] * (rhs[ijk] - Ax);
      } else if (MODE == MODE_RESIDUAL) {
        out[ijk] = rhs[ijk] - Ax;
      } else {
        out[ijk] = Ax;
      }
    } else if (P.copy_other_colour) {
      out[ijk] = xc;
    }
    xm = xc; xc = xp; bk0 = bk1;
  }
}

// ---- smoother-kernel profiling: hipEvent pair around every smoother launch ----
static bool g_profile = false;
static long long g_profile_min_cells = 0;   // only launches covering at least this many cells are timed
static const int kMaxPairs = 8192;
static hipEvent_t g_ev[2 * kMaxPairs];
static int g_pairs_alloc = 0, g_pairs_used = 0;
static long long g_prof_cells = 0, g_prof_launches = 0;
static double g_prof_ms_flushed = 0.0;

static void profile_flush() {
  for (int p = 0; p < g_pairs_used; p++) {
    float ms = 0.f;
    hipEventSynchronize(g_ev[2 * p + 1]);
    if (hipEventElapsedTime(&ms, g_ev[2 * p], g_ev[2 * p + 1]) == hipSuccess) g_prof_ms_flushed += ms;
  }
  g_pairs_used = 0;
}
extern "C" int hpgmg_hip_graph_is_open(void);
static int profile_begin(long long cells) {
  if (!g_profile || cells < g_profile_min_cells || hpgmg_hip_graph_is_open()) return -1;
  if (g_pairs_used == kMaxPairs) profile_flush();
  if (g_pairs_used == g_pairs_alloc) { hipEventCreate(&g_ev[2 * g_pairs_alloc]); hipEventCreate(&g_ev[2 * g_pairs_alloc + 1]); g_pairs_alloc++; }
  int p = g_pairs_used++;
  hipEventRecord(g_ev[2 * p], g_stream);
  return p;
}
static void profile_end(int p, long long cells) {
  if (p < 0) return;
  hipEventRecord(g_ev[2 * p + 1], g_stream);
  g_prof_cells += cells; g_prof_launches++;
}

static int g_ghost_free = 0;

static void plan(const hpgmg_hip_level *L, StencilArgs &P, dim3 &block, int &grid) {
  P.ghost_free = (g_ghost_free && L->box_nbr) ? 1 : 0;
  int tx = 64;
  while (tx > 1 && tx / 2 >= L->dim) tx /= 2;           // smallest power of two >= dim, capped at one wave
  int ty = 256 / tx;
  while (ty > 1 && ty / 2 >= L->dim) ty /= 2;
  block = dim3(tx, ty, 1);
  P.tiles_i = (L->dim + tx - 1) / tx;
  P.tiles_j = (L->dim + ty - 1) / ty;
  // k chunk: long enough to amortise the two start-up planes, short enough to expose >= ~8 blocks per CU
  int kchunk = L->dim;
  while (kchunk > 16 && (long long)L->num_boxes * P.tiles_i * P.tiles_j * ((L->dim + kchunk - 1) / kchunk) < 2048) kchunk /= 2;
  P.kchunk = kchunk;
  P.chunks_k = (L->dim + kchunk - 1) / kchunk;
  P.total_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i;
  grid = grid_for(P.total_blocks, &P.per_xcd);
}

template <int MODE>
static int launch(const hpgmg_hip_level *L, int variant, StencilArgs P, bool is_smoother) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  dim3 block; int grid;
  plan(L, P, block, grid);
  const long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  int prof = is_smoother ? profile_begin(cells) : -1;
  switch (variant) {
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    case HPGMG_HIP_7PT_VC_POISSON:   hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_VC_POISSON, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    case HPGMG_HIP_7PT_CC:           hipLaunchKernelGGL((stencil7_kernel<HPGMG_HIP_7PT_CC, MODE>), dim3(grid), block, 0, g_stream, *L, P); break;
    default: return record_error(hipErrorInvalidValue, "stencil variant not implemented");
  }
  profile_end(prof, cells);
  HPGMG_LAUNCH_CHECK("stencil7_kernel");
  return 0;
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

void hpgmg_hip_set_ghost_free(int on) { g_ghost_free = on; }
int hpgmg_hip_get_ghost_free(void) { return g_ghost_free; }

void hpgmg_hip_profile_smoother(int enable) {
  if (enable) { profile_flush(); g_prof_ms_flushed = 0.0; g_prof_cells = 0; g_prof_launches = 0; }
  g_profile = enable != 0;
}
void hpgmg_hip_profile_smoother_min_cells(long long min_cells) { g_profile_min_cells = min_cells; }
int hpgmg_hip_profile_smoother_read(double *total_ms, long long *launches, long long *cells) {
  profile_flush();
  if (total_ms) *total_ms = g_prof_ms_flushed;
  if (launches) *launches = g_prof_launches;
  if (cells) *cells = g_prof_cells;
  return 0;
}

int hpgmg_hip_smooth_cheby(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                           double a, double b, double h2inv, double c1, double c2) {
  StencilArgs P = {}; P.xn_id = xn_id; P.xout_id = xnp1_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv; P.c1 = c1; P.c2 = c2;
  return launch<MODE_CHEBY>(L, variant, P, true);
}
int hpgmg_hip_smooth_gsrb(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                          double a, double b, double h2inv, int sweep) {
  StencilArgs P = {}; P.xn_id = xn_id; P.xout_id = xnp1_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv; P.sweep = sweep;
  P.copy_other_colour = (xn_id != xnp1_id);
  return launch<MODE_GSRB>(L, variant, P, true);
}
int hpgmg_hip_smooth_jacobi(const hpgmg_hip_level *L, int variant, int xn_id, int xnp1_id, int rhs_id,
                            double a, double b, double h2inv, double weight) {
  StencilArgs P = {}; P.xn_id = xn_id; P.xout_id = xnp1_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv; P.c2 = weight;
  return launch<MODE_JACOBI>(L, variant, P, true);
}
int hpgmg_hip_residual(const hpgmg_hip_level *L, int variant, int res_id, int x_id, int rhs_id, double a, double b, double h2inv) {
  StencilArgs P = {}; P.xn_id = x_id; P.xout_id = res_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv;
  if (rhs_id < 0) return launch<MODE_APPLY>(L, variant, P, false);
  return launch<MODE_RESIDUAL>(L, variant, P, false);
}

}  // extern "C"

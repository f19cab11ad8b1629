// fv4_rb.hip -- launchers of the one-pass red + black GSRB sweep of the 4th-order operator (fv4_rb.hpp) and of its pre-pass.
// A translation unit of its own: the kernel lives at the edge of the 256 registers two waves per SIMD leave it, and is compiled with the
// register-minimising scheduler (Makefile: FLAGS_fv4_rb) without changing how the other kernels are scheduled.
#include <stdlib.h>
#include <vector>
#include "common.hpp"
#include "fv4_tile.hpp"
#include "fv4_rb.hpp"

namespace hpgmg {
int  profile_begin(long long cells);          // stencil.hip: hipEvent pair around a smoother launch (bench.py's roofline)
void profile_end(int p, long long cells, bool first_part = false);
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
}
using namespace hpgmg;

extern "C" {

// Both coloured half sweeps (sweep, sweep + 1; sweep even) of an out-of-place GSRB sweep of the 4th-order operator in one pass
// (fv4_rb.hpp).  Vectors are (scratch?, id) pairs: scratch ids address the plugin-private vectors behind scr_base.  Boxes of side 64 m,
// all of them local, two ghost cells; apply_BCs_v4 done on the input; on Dirichlet levels hpgmg_hip_fv4_rb_prepass run before.
static long long g_rb4_launches = 0;
#ifdef HPGMG_EXP_TIMELINE
static unsigned long long *g_fv4rb_timeline = nullptr;
void hpgmg_hip_exp_timeline_fv4(void *buf) { g_fv4rb_timeline = (unsigned long long *)buf; }
#endif
long long hpgmg_hip_rb_fv4_launch_count(void) { return g_rb4_launches; }
int hpgmg_hip_smooth_gsrb_fv4_rb_supported(const hpgmg_hip_level *L, int variant) {
  static const int off = env_int("HPGMG_TUNE_FV4_NO_RB", 0);
  if (variant != HPGMG_HIP_FV4_VC_HELMHOLTZ && variant != HPGMG_HIP_FV4_VC_POISSON) return 0;
  // boxes of side 64 m; 32-wide tiles (two workgroups per CU) also take boxes of 32^3, but measured slower there than two half-sweep launches
  // of the tiled kernel (98 vs 2 x 35 us on a 128^3 level of 64 boxes: the march is too short for its prologue), so only on request
  const int need = 64;                                    // (32-wide tiles, two workgroups per CU, measured slower in two rounds: removed)
  return !off && L->num_boxes > 0 && L->dim % need == 0 && L->box_nbr != nullptr && L->ghosts == 2;
}
// Dispatch order of the tiles.  A tile at a domain wall in i or j forms the boundary values of the intermediate vector in every step
// (two more barriers, a short serial stage): its workgroup takes about a quarter longer, and with 4 workgroups per CU in a launch the
// CUs that drew several of them finish last.  Each XCD keeps its contiguous range of tiles (common.hpp) but starts with the slow ones.
// Built once per (level geometry, tiling) from the box neighbour table; any order gives the same numbers.
struct TileOrder { const int *nbr; int num_boxes, dim, ti, kchunk, grid; int *d_order; };
static std::vector<TileOrder> g_tile_orders;
static const int *tile_order(const hpgmg_hip_level *L, const Fv4RbArgs &A, int TI, int grid) {
  static const int on = env_int("HPGMG_TUNE_FV4_RB_ORDER", 1);
  if (!on || A.total_blocks <= 256) return nullptr;                  // one round: every workgroup starts at once anyway
  for (const TileOrder &o : g_tile_orders)
    if (o.nbr == L->box_nbr && o.num_boxes == L->num_boxes && o.dim == L->dim && o.ti == TI && o.kchunk == A.kchunk && o.grid == grid) return o.d_order;
  std::vector<int> nbr(6 * (size_t)L->num_boxes), order((size_t)grid);
  if (hipMemcpy(nbr.data(), L->box_nbr, nbr.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
  for (int x = 0; x < kXcds; x++) {
    int pos = x * A.per_xcd;
    for (int pass = 0; pass < 2; pass++)
      for (int l = x * A.per_xcd; l < (x + 1) * A.per_xcd; l++) {
        if (l >= A.total_blocks) continue;
        int t = l;
        const int ti = t % A.tiles_i; t /= A.tiles_i;
        const int tj = t % A.tiles_j; t /= A.tiles_j;
        t /= A.chunks_k;
        const int *nb = &nbr[6 * (size_t)t];
        const bool wall = (nb[0] == -1 && ti == 0) || (nb[1] == -1 && ti == A.tiles_i - 1) || (nb[2] == -1 && tj == 0) || (nb[3] == -1 && tj == A.tiles_j - 1);
        if (wall == (pass == 0)) order[pos++] = l;
      }
    while (pos < (x + 1) * A.per_xcd) order[pos++] = A.total_blocks;      // the padding of the grid: nothing to do
  }
  int *d = nullptr;
  if (hipMalloc(&d, order.size() * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  g_tile_orders.push_back(TileOrder{L->box_nbr, L->num_boxes, L->dim, TI, A.kchunk, grid, d});
  return d;
}
static VecSel vec_sel(const hpgmg_hip_level *L, double *const *scr_base, int scratch, int id) { return VecSel{scratch ? scr_base : L->box_base, id}; }
// The ghost planes of the intermediate vector t below / above the domain, into the k ghost zone of scratch vector tg_id: a red half sweep
// of x on the four planes next to each k wall (the tiled kernel, restricted to those planes and boxes) into tg's interior, then
// apply_BCs_v4 of tg over the boundary blocks whose domain normal has a k component (faces and the i-k / j-k edges; entries: the
// host-computed geometry of exactly those blocks, sources read from the box that owns them).
int hpgmg_hip_fv4_rb_prepass(const hpgmg_hip_level *L, int variant, double *const *scr_base, int x_scratch, int x_id, int tg_id, int rhs_id,
                             double a, double b, double h2inv, int sweep, const hpgmg_hip_bc_entry *entries_k, int n_k, const int *special_cells, int n_special) {
  HPGMG_SKIP_IF_REPLAY();
  if (!hpgmg_hip_smooth_gsrb_fv4_rb_supported(L, variant) || (sweep & 1) || !scr_base) return record_error(hipErrorInvalidValue, "fv4_rb_prepass: level / arguments not supported");
  // the cells of internal box faces next to a domain wall: their t with the owning box's coefficients -- in the same launch as the planes
  // next to the k walls when there are any
  Fv4SpecialArgs S = {};
  S.x = vec_sel(L, scr_base, x_scratch, x_id); S.tg = vec_sel(L, scr_base, 1, tg_id); S.rhs_id = rhs_id; S.a = a; S.b = b; S.h2inv = h2inv; S.sweep = sweep;
  S.cells = special_cells; S.n = n_special > 0 ? n_special : 0;
  if (n_special > 0 && n_k <= 0) {
    if (variant == HPGMG_HIP_FV4_VC_HELMHOLTZ) hipLaunchKernelGGL((fv4_special_kernel<HPGMG_HIP_FV4_VC_HELMHOLTZ>), dim3((n_special + 255) / 256), dim3(256), 0, g_stream, *L, S);
    else hipLaunchKernelGGL((fv4_special_kernel<HPGMG_HIP_FV4_VC_POISSON>), dim3((n_special + 255) / 256), dim3(256), 0, g_stream, *L, S);
    HPGMG_LAUNCH_CHECK("fv4_special_kernel");
  }
  if (n_k <= 0) return 0;                                  // no k wall on this rank's boxes (periodic)
  Fv4TileArgs P = {};
  P.xn_id = x_id; P.xout_id = tg_id; P.rhs_id = rhs_id; P.a = a; P.b = b; P.h2inv = h2inv; P.sweep = sweep; P.copy_other_colour = 1; P.ghost_free = 1;
  P.x_base = x_scratch ? scr_base : nullptr; P.out_base = scr_base;
  P.kchunk = 4; P.chunks_k = 2; P.k_origin = 0; P.k_step = L->dim - 4; P.wall_only = 1;
#define FV4_PRE_CASE(VAR, TJ, TI) { \
    P.tiles_i = L->dim / TI; P.tiles_j = L->dim / TJ; P.total_blocks = L->num_boxes * P.chunks_k * P.tiles_j * P.tiles_i; \
    const int grid = grid_for(P.total_blocks, &P.per_xcd); \
    const size_t lds = (size_t)11 * (TI + 4) * (TJ + 4) * sizeof(double); \
    const int sp_blocks = (S.n + TI * TJ - 1) / (TI * TJ); \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)fv4_rb_prepass_kernel<VAR, TJ, TI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; } \
    hipLaunchKernelGGL((fv4_rb_prepass_kernel<VAR, TJ, TI>), dim3(grid + sp_blocks), dim3(TI, TJ), lds, g_stream, *L, P, S, sp_blocks); }
  const bool wide = L->dim % 64 == 0;
  if (variant == HPGMG_HIP_FV4_VC_HELMHOLTZ) { if (wide) FV4_PRE_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ, 8, 64) else FV4_PRE_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ, 16, 32) }
  else                                       { if (wide) FV4_PRE_CASE(HPGMG_HIP_FV4_VC_POISSON, 8, 64) else FV4_PRE_CASE(HPGMG_HIP_FV4_VC_POISSON, 16, 32) }
#undef FV4_PRE_CASE
  HPGMG_LAUNCH_CHECK("fv4_rb_prepass_kernel");
  hpgmg_hip_level Ls = *L;
  Ls.box_base = scr_base;
  return hpgmg_hip_exchange_and_bc(&Ls, tg_id, nullptr, 0, entries_k, n_k, 4);
}
int hpgmg_hip_smooth_gsrb_fv4_rb(const hpgmg_hip_level *L, int variant, double *const *scr_base, int x_scratch, int x_id, int out_scratch, int out_id,
                                 int tg_id, int rhs_id, double a, double b, double h2inv, int sweep) {
  HPGMG_SKIP_IF_REPLAY();
  if (!hpgmg_hip_smooth_gsrb_fv4_rb_supported(L, variant) || (x_scratch == out_scratch && x_id == out_id) || (sweep & 1) || !scr_base)
    return record_error(hipErrorInvalidValue, "smooth_gsrb_fv4_rb: level / arguments not supported");
  Fv4RbArgs A = {};
  A.x = vec_sel(L, scr_base, x_scratch, x_id); A.out = vec_sel(L, scr_base, out_scratch, out_id); A.tg = vec_sel(L, scr_base, 1, tg_id);
  A.rhs_id = rhs_id; A.a = a; A.b = b; A.h2inv = h2inv; A.sweep = sweep;
  const int TI = 64;                                      // one workgroup of 8 waves per CU
  A.tiles_i = L->dim / TI; A.tiles_j = L->dim / fv4rb::TJ;
  int kchunk = L->dim;                                   // 256 (512) resident workgroups fill the chip; every k chunk costs four extra planes of loads and two red stages
  while (kchunk > 16 && (long long)L->num_boxes * A.tiles_i * A.tiles_j * (L->dim / kchunk) < (TI == 64 ? 256 : 512)) kchunk /= 2;
  static const int tune_kc = env_int("HPGMG_TUNE_FV4_RB_KCHUNK", 0);
  if (tune_kc > 0 && L->dim % tune_kc == 0) kchunk = tune_kc;
  A.kchunk = kchunk; A.chunks_k = (L->dim + kchunk - 1) / kchunk;
  A.total_blocks = L->num_boxes * A.chunks_k * A.tiles_j * A.tiles_i;
  int grid = grid_for(A.total_blocks, &A.per_xcd);
  long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  const long long whole_cells = cells;
  if (g_tile_part) {      // one part of the launch: part 1 = tiles that read neither an image of another rank's box nor anything the pre-pass forms (no wall)
    int count = 0;
    A.order = tile_part_order(L, A.tiles_i, A.tiles_j, A.chunks_k, g_tile_part, true, &grid, &A.per_xcd, &count);
    if (grid == 0) return 0;
    if (!A.order) return record_error(hipErrorOutOfMemory, "smooth_gsrb_fv4_rb: dispatch list of a partial launch");
    cells = cells * count / A.total_blocks;
  } else A.order = tile_order(L, A, TI, grid);
#ifdef HPGMG_EXP_TIMELINE
  A.timeline = g_fv4rb_timeline; A.timeline_wg = env_int("HPGMG_EXP_TIMELINE_WG", -1);
#endif
  const int prof = profile_begin(whole_cells);               // (the caller's size threshold is on the whole launch; a part reports its share of the cells)
#define FV4_RB_CASE(VAR, TI_) { \
    constexpr size_t lds = fv4rb::Geom<TI_>::LDS_BYTES; \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)fv4_rb_kernel<VAR, TI_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; } \
    hipLaunchKernelGGL((fv4_rb_kernel<VAR, TI_>), dim3(grid), dim3(TI_, 8), lds, g_stream, *L, A); }
  if (variant == HPGMG_HIP_FV4_VC_HELMHOLTZ) FV4_RB_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ, 64) else FV4_RB_CASE(HPGMG_HIP_FV4_VC_POISSON, 64)
#undef FV4_RB_CASE
  g_rb4_launches++;
  profile_end(prof, 2 * cells, g_tile_part == 1);       // one launch = two half sweeps over every cell
  HPGMG_LAUNCH_CHECK("fv4_rb_kernel");
  return 0;
}

}  // extern "C"

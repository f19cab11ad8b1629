// pair.hip -- launchers of the two-sweeps-per-pass kernel of the 7-point operator (cheby_pair.hpp: Chebyshev sweep pairs, chebyshev.c:43-99, and
// pairs of in-place GSRB half sweeps, gsrb.c:24-132), of its pre-pass, of the packed pre-pass coefficients, of the two-part launches across rank
// boundaries, and the pack / unpack kernel of its halo messages.  A translation unit of its own (split from stencil.hip in round 4).
#include "stencil_direct.hpp"
#include "cheby_pair.hpp"

namespace hpgmg {
// ---- halo of a sweep pair across rank boundaries: regions of x0 / xm1 / rhs (or a level vector) <-> one message buffer ----
// fold: packing x0 (vec 0) of a smooth() whose interpolation_vcycle is folded into its first sweep pair -- the owner adds the coarse parent
// (interpolation_p0.c:43: prescale * f + c[i>>1, j>>1, k>>1]) while packing, so the receiver's ghost zones and deep planes hold the interpolated x0
struct HaloRefs { VecRef x0, xm1; int rhs_id; double *const *scr_base; int fold; hpgmg_hip_level Lc; int coarse_id; double prescale; };
__device__ __forceinline__ double *halo_vec(const hpgmg_hip_level &L, const HaloRefs &R, int vec, int box) {
  const size_t first = (size_t)L.ghosts * (size_t)(1 + L.jStride + L.kStride);
  if (vec >= 16) return L.box_base[box] + (size_t)(vec - 16) * (size_t)L.volume + first;
  if (vec == 2) return L.box_base[box] + (size_t)R.rhs_id * (size_t)L.volume + first;
  const VecRef r = (vec == 0) ? R.x0 : R.xm1;
  return (r.scratch ? R.scr_base[box] : L.box_base[box]) + (size_t)r.id * (size_t)L.volume + first;
}
template <bool kUnpack>
__global__ __launch_bounds__(256) void pair_halo_kernel(const hpgmg_hip_level L, const HaloRefs R, const hpgmg_hip_halo_entry *__restrict__ list,
                                                        double *buf, double *deep, double *deep_beta) {
  const hpgmg_hip_halo_entry e = list[blockIdx.x];
  const int n = e.ni * e.nj * e.nk, jS = L.jStride, kS = L.kStride;
  double *v = halo_vec(L, R, e.vec, e.box) + e.i + e.j * jS + e.k * kS;
  double *b = buf + e.off;
  double *plane = nullptr;
  if (kUnpack && e.deep >= 8) plane = deep_beta + ((size_t)e.box * 3 + (e.deep - 8)) * (size_t)L.dim * L.dim;
  else if (kUnpack && e.deep >= 0) plane = deep + ((size_t)e.box * 6 + e.deep) * (size_t)L.dim * L.dim;
  for (int t = blockIdx.y * 256 + threadIdx.x; t < n; t += gridDim.y * 256) {
    const int ti = t % e.ni, tj = (t / e.ni) % e.nj, tk = t / (e.ni * e.nj);
    if (!kUnpack) {
      double val = v[ti + tj * jS + tk * kS];
      if (R.fold && e.vec == 0) {
        const int ci = e.i + ti, cj = e.j + tj, ck = e.k + tk;              // a cell of the sender's interior
        val = R.prescale * val + vec_origin(R.Lc, e.box, R.coarse_id)[(ci >> 1) + (cj >> 1) * R.Lc.jStride + (ck >> 1) * R.Lc.kStride];
      }
      b[t] = val;
    }
    else if (plane) plane[t] = b[t];
    else v[ti + tj * jS + tk * kS] = b[t];
  }
}
#ifdef HPGMG_EXP_TIMELINE
double *g_exp_timeline = nullptr;     // experiment build: where a kernel's chosen workgroup records its step timeline
#endif
}  // namespace hpgmg
using namespace hpgmg;

extern "C" {
int hpgmg_hip_graph_flush(void);

// Two Chebyshev sweeps in one pass (cheby_pair.hpp).  Vector references are (scratch?, id) pairs: scratch ids 0/1
// address the two plugin-private vectors behind scr_base.  Returns hipErrorNotSupported-like status 1 (no launch,
// no error recorded) when the level does not fit the kernel's assumptions, so the caller can fall back.
static int pair_supported_dims(const hpgmg_hip_level *L, int variant, int Di, int Dj, int Dk) {
  if (variant != HPGMG_HIP_7PT_VC_HELMHOLTZ && variant != HPGMG_HIP_7PT_VC_POISSON && variant != HPGMG_HIP_7PT_CC) return 0;
  if (L->num_boxes <= 0 || L->num_boxes > kPairMaxBoxes || L->periodic || !(L->flags & 1) || L->ghosts < 1 || Di % 128 != 0) return 0;
  // a wave owns a 128-cell row: whole multiples of 128 per box, or several boxes (consecutive in one slab) per row
  if (L->dim % 128 != 0 && !(128 % L->dim == 0 && L->dim >= 16 && (L->box_stride > 0 || L->num_boxes == 1))) return 0;
  if (L->jStride % 2 || L->kStride % 2 || L->volume % 2) return 0;
  if (Di % L->dim || Dj % L->dim || Dk % L->dim) return 0;
  if ((long long)(Di / L->dim) * (Dj / L->dim) * (Dk / L->dim) != L->num_boxes) return 0;
  return 1;
}
int hpgmg_hip_smooth_cheby_pair_supported(const hpgmg_hip_level *L, int variant) { return pair_supported_dims(L, variant, L->dim_i, L->dim_j, L->dim_k); }
// several ranks: this rank's boxes form a brick of nbi x nbj x nbk boxes (numbered lexicographically inside it)
int hpgmg_hip_smooth_cheby_pair_supported_brick(const hpgmg_hip_level *L, int variant, int nbi, int nbj, int nbk) {
  return pair_supported_dims(L, variant, nbi * L->dim, nbj * L->dim, nbk * L->dim);
}
int hpgmg_hip_coef32_refresh(const hpgmg_hip_level *L, float *const *c32_base, int num_vectors) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  hipLaunchKernelGGL(coef32_convert_kernel, dim3(512, L->num_boxes), dim3(256), 0, g_stream, *L, c32_base, num_vectors);
  HPGMG_LAUNCH_CHECK("coef32_convert_kernel");
  return 0;
}
// interpolation_vcycle folded into the NEXT sweep-pair launch (consumed by it): x0 := prescale * x0 + parent(coarse_id of Lc)
// remote faces of the NEXT sweep-pair launch (consumed by it): see hpgmg_hip_pair_set_halo
static bool g_pair_halo_set = false;
static int g_pair_rem[6], g_pair_brick[3];
static const double *g_pair_deep = nullptr, *g_pair_deep_beta = nullptr;
static long long g_pair_launches = 0, g_pair_remote_launches = 0;
static int g_pair_discard_x1 = 0;     // consumed by the next Chebyshev pair launch: its out1 vector is scratch, do not store x1
static const hpgmg_hip_level *g_pair_interp_level = nullptr;
static int g_pair_interp_id = 0;
static double g_pair_interp_prescale = 1.0;
// The pre-pass's coefficient values, packed once per operator rebuild (cheby_pair.hpp: PairArgs.edge_coef).  Keyed by the level's box table;
// hpgmg_hip_pair_packed_invalidate() when the coefficients change, _forget() when the level goes away.  HPGMG_TUNE_PAIR_PACKED=0: read in place.
struct EdgePack { const void *key; int variant, Di, Dj, Dk; double *buf; bool valid; };
static std::vector<EdgePack> g_edge_packs;
extern "C" void hpgmg_hip_pair_packed_invalidate(const hpgmg_hip_level *L) { for (EdgePack &e : g_edge_packs) if (!L || e.key == (const void *)L->box_base) e.valid = false; }
extern "C" void hpgmg_hip_pair_packed_forget(const hpgmg_hip_level *L) {
  for (size_t q = 0; q < g_edge_packs.size();) {
    if (!L || g_edge_packs[q].key == (const void *)L->box_base) { (void)hipStreamSynchronize(g_stream); (void)hipFree(g_edge_packs[q].buf); g_edge_packs.erase(g_edge_packs.begin() + (long)q); }
    else q++;
  }
}
static EdgePack *edge_pack_slot(const hpgmg_hip_level *L, int variant, const PairArgs &A) {
  static const int on = env_int("HPGMG_TUNE_PAIR_PACKED", 1);
  if (!on || A.tiles_i < 2 || variant == HPGMG_HIP_7PT_CC) return nullptr;
  for (EdgePack &e : g_edge_packs) if (e.key == (const void *)L->box_base && e.variant == variant && e.Di == A.Di && e.Dj == A.Dj && e.Dk == A.Dk) return &e;
  EdgePack e = { (const void *)L->box_base, variant, A.Di, A.Dj, A.Dk, nullptr, false };
  const size_t n = (size_t)2 * (A.tiles_i - 1) * A.Dk * A.Dj * 8;
  if (hipMalloc((void **)&e.buf, n * sizeof(double)) != hipSuccess) return nullptr;
  g_edge_packs.push_back(e);
  return &g_edge_packs.back();
}
// Two-part launches of the sweep pair across rank boundaries (hpgmg_hip_set_tile_part): part 1 = the workgroups whose slab / chunk / tile touches
// no face another rank owns -- they read nothing the halo exchange delivers (ghost zones, deep planes, ghost columns of the pre-pass) --, part 2 the
// others.  Each part runs the (cheap) pre-pass in full: before the exchange its cells next to remote faces are formed from stale ghost values and
// read by nobody, the second run overwrites them.
struct PairOrder { int ti, sj, ck, rem[6], part; int *d_order; int grid, per_xcd, count; };
static std::vector<PairOrder> g_pair_orders;
static const PairOrder *pair_part_order(const PairArgs &A, int part) {
  for (const PairOrder &o : g_pair_orders)
    if (o.ti == A.tiles_i && o.sj == A.slabs_j && o.ck == A.chunks_k && o.part == part && memcmp(o.rem, A.rem, sizeof o.rem) == 0) return &o;
  PairOrder o = {}; o.ti = A.tiles_i; o.sj = A.slabs_j; o.ck = A.chunks_k; o.part = part; memcpy(o.rem, A.rem, sizeof o.rem);
  std::vector<int> sel;
  for (int l = 0; l < A.total_blocks; l++) {
    int t = l;
    const int ti = t % A.tiles_i; t /= A.tiles_i;
    const int sj = t % A.slabs_j; t /= A.slabs_j;
    const int ck = t;
    const bool later = (ti == 0 && A.rem[0]) || (ti == A.tiles_i - 1 && A.rem[1]) || (sj == 0 && A.rem[2]) || (sj == A.slabs_j - 1 && A.rem[3]) || (ck == 0 && A.rem[4]) || (ck == A.chunks_k - 1 && A.rem[5]);
    if (later == (part == 2)) sel.push_back(l);
  }
  o.count = (int)sel.size();
  if (o.count > 0) {
    o.per_xcd = (o.count + kXcds - 1) / kXcds; o.grid = o.per_xcd * kXcds;
    sel.resize((size_t)o.grid, A.total_blocks);
    if (hipMalloc((void **)&o.d_order, sel.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(o.d_order, sel.data(), sel.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(o.d_order); return nullptr; }
  }
  g_pair_orders.push_back(o);
  return &g_pair_orders.back();
}
static int smooth_pair(const hpgmg_hip_level *L, int variant, int gsrb, int sweep_a, double *const *scr_base, const float *const *c32_base,
                       int x0_scr, int x0_id, int xm1_scr, int xm1_id, int out1_scr, int out1_id, int out2_scr, int out2_id,
                       int rhs_id, double a, double b, double h2inv, double c1a, double c2a, double c1b, double c2b) {
  // the requests set for THIS launch (hpgmg_hip_pair_set_halo / _discard_x1 / _set_interp) are taken here, before any exit: whatever
  // happens below, none of them can stay pending and change a later launch
  const bool remote = g_pair_halo_set;
  const int discard_x1 = g_pair_discard_x1;
  const hpgmg_hip_level *const interp_level = g_pair_interp_level;
  g_pair_halo_set = false; g_pair_discard_x1 = 0; g_pair_interp_level = nullptr;
  HPGMG_SKIP_IF_REPLAY();
  const int Di = remote ? g_pair_brick[0] * L->dim : L->dim_i, Dj = remote ? g_pair_brick[1] * L->dim : L->dim_j, Dk = remote ? g_pair_brick[2] * L->dim : L->dim_k;
  if (!pair_supported_dims(L, variant, Di, Dj, Dk)) return record_error(hipErrorInvalidValue, "smooth_cheby_pair: level not supported");
  if (remote && c32_base) return record_error(hipErrorInvalidValue, "smooth_cheby_pair: remote faces need fp64 coefficients");
  static const int tune_kc = env_int("HPGMG_TUNE_PAIR_KC", 0);
  static const int tune_nw = env_int("HPGMG_TUNE_PAIR_NW", 0);
  // Waves per workgroup (nw - 2 output rows each) and k chunk.  A workgroup occupies a CU (one fits: LDS, registers) and costs KC + 2 plane steps; a step
  // costs about in proportion to its waves (the CU's load path is what a step waits for) -- measured at 256^3: 4.6 / 5.1 / 7.5 us with 10 / 12 / 16 waves (the
  // 16-wave kernel is held to 128 registers and spills 7).  The launch takes ceil(workgroups / 256) rounds x (KC + 2) steps x that: pick the pair that minimises
  // it.  What decides is how the grid FITS the 256 CUs: 256^3 with 16 waves is 2 x 19 x 6 = 228 workgroups of 45 steps (338 us); with 10 waves 2 x 32 x 4 = 256
  // workgroups of 66 shorter steps (303 us); 512^3 runs best with 12 (9 rounds of 49: 2.67 ms against 2.85 with 16).  tools/ab_pair_nw.sh,
  // profiles/r06n_ab_pair_nw.txt.
  int nw = 16, kc = tune_kc;
  {
    const int slots = 256, cand[3] = {10, 12, 16}, step_cost[3] = {46, 51, 75};      // (tenths of a microsecond per step)
    long long best_cost = -1;
    for (int ci = 0; ci < 3; ci++) {
      const int w = cand[ci];
      if (tune_nw > 0 && w != tune_nw) continue;
      if (tune_nw <= 0 && c32_base && w == 10) continue;      // (fp32 coefficient streams: the conversions make a step's cost less a matter of its waves; 256^3: 2.92 ms with 10 as with 16, 2.85 with 12)
      const int per_plane = (Di / 128) * ((Dj + (w - 2) - 1) / (w - 2));
      auto steps = [&](int c) { const long long wgs = (long long)per_plane * ((Dk + c - 1) / c); return ((wgs + slots - 1) / slots) * (c + 2); };
      int k = tune_kc;
      if (k <= 0) {
        long long best = -1;
        for (int c = 8; c <= 64 && c <= Dk; c++) { const long long cost = steps(c); if (best < 0 || cost < best) { best = cost; k = c; } }
        // ... and among the chunk lengths within 3 % of that, the LONGEST: two of every KC + 2 planes a workgroup fetches are halo, and the launch is close enough
        // to the memory system's limit for 5 % fewer bytes to outweigh one more step (tools/ab_pair_kc.sh, profiles/r06e_ab_pair_kc.txt)
        for (int c = k + 1; c <= 64 && c <= Dk; c++) if (steps(c) * 100 <= best * 103) k = c;
      }
      const long long cost = steps(k) * step_cost[ci];
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; nw = w; kc = k; }
    }
  }
  PairArgs A = {};
  A.x0 = VecRef{x0_scr, x0_id}; A.xm1 = VecRef{xm1_scr, xm1_id}; A.out1 = VecRef{out1_scr, out1_id}; A.out2 = VecRef{out2_scr, out2_id};
  A.rhs_id = rhs_id; A.a = a; A.b = b; A.h2inv = h2inv; A.c1a = c1a; A.c2a = c2a; A.c1b = c1b; A.c2b = c2b;
  A.scr_base = scr_base; A.c32_base = c32_base; A.sweep_a = sweep_a;
  A.keep_x1 = discard_x1 ? 0 : 1;
  const bool interp = (interp_level != nullptr);
  if (interp) {
    const hpgmg_hip_level *C = interp_level;
    if ((remote && L->dim % 128 != 0) || C->num_boxes != L->num_boxes || C->num_boxes > kPairMaxBoxes || 2 * C->dim != L->dim) return record_error(hipErrorInvalidValue, "smooth pair with interpolation: level pair not supported");
    A.Lc = *C; A.coarse_id = g_pair_interp_id; A.prescale = g_pair_interp_prescale;
  }
  A.nbi = Di / L->dim; A.nbj = Dj / L->dim;
  A.Di = Di; A.Dj = Dj; A.Dk = Dk;
  if (remote) { for (int d = 0; d < 6; d++) A.rem[d] = g_pair_rem[d]; A.deep = g_pair_deep; A.deep_beta = g_pair_deep_beta; }
#ifdef HPGMG_EXP_TIMELINE
  else A.deep = g_exp_timeline;
#endif
  A.tiles_i = A.Di / 128; A.slabs_j = (A.Dj + (nw - 2) - 1) / (nw - 2); A.KC = kc; A.chunks_k = (A.Dk + kc - 1) / kc;
  A.total_blocks = A.tiles_i * A.slabs_j * A.chunks_k;
  int grid = grid_for(A.total_blocks, &A.per_xcd);
  long long cells = (long long)A.Di * A.Dj * A.Dk;
  const long long whole_cells = cells;
  const int part = remote ? g_tile_part : 0;
  if (part) {
    const PairOrder *o = pair_part_order(A, part);
    if (!o) return record_error(hipErrorOutOfMemory, "smooth pair: dispatch list of a partial launch");
    if (o->count == 0) return 0;
    A.order = o->d_order; grid = o->grid; A.per_xcd = o->per_xcd;
    cells = cells * o->count / A.total_blocks;
  }
  const size_t lds = (size_t)nw * 6 * 64 * sizeof(p2);
  if (!remote && !c32_base) {        // the pre-pass reads its coefficient values packed; (re)pack them after an operator rebuild
    EdgePack *pk = edge_pack_slot(L, variant, A);
    if (pk) {
      A.edge_coef = pk->buf;
      if (!pk->valid) {
        A.edge_blocks = ((A.Dj + 63) / 64) * A.Dk * 2 * (A.tiles_i - 1);
        const dim3 pgrid(grid_for(A.edge_blocks, &A.edge_per_xcd));
        if (variant == HPGMG_HIP_7PT_VC_HELMHOLTZ) hipLaunchKernelGGL((cheby_pair_edge_kernel<HPGMG_HIP_7PT_VC_HELMHOLTZ, false, PAIR_CHEBY, false, false, true>), pgrid, dim3(64), 0, g_stream, *L, A);
        else hipLaunchKernelGGL((cheby_pair_edge_kernel<HPGMG_HIP_7PT_VC_POISSON, false, PAIR_CHEBY, false, false, true>), pgrid, dim3(64), 0, g_stream, *L, A);
        HPGMG_LAUNCH_CHECK("cheby_pair_edge_kernel (packing)");
        pk->valid = true;
      }
    }
  }
  const int prof = profile_begin(whole_cells);
  // (the waves per workgroup are a template parameter of the kernel: one instance per candidate)
#define PAIR_NW_SWITCH(...) switch (nw) { case 10: { constexpr int NWC = 10; __VA_ARGS__ } break; case 12: { constexpr int NWC = 12; __VA_ARGS__ } break; default: { constexpr int NWC = 16; __VA_ARGS__ } break; }
#define PAIR_LAUNCH2(VAR, C32, SM, NARROW, INTERP) PAIR_NW_SWITCH( \
      static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)cheby_pair_kernel<VAR, NWC, C32, SM, NARROW, INTERP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)NWC * 6 * 64 * sizeof(p2)))); once = true; } \
      hipLaunchKernelGGL((cheby_pair_kernel<VAR, NWC, C32, SM, NARROW, INTERP>), dim3(grid), dim3(64, NWC), lds, g_stream, *L, A); )
#define PAIR_LAUNCH_REMOTE_IP(VAR, SM, NRW, IP) { \
      const int ecols = 2 * (A.tiles_i - 1) + (A.rem[0] ? 1 : 0) + (A.rem[1] ? 1 : 0); \
      A.edge_blocks = ((A.Dj + 63) / 64) * A.Dk * ecols; const int egrid_r = grid_for(A.edge_blocks, &A.edge_per_xcd); \
      if (ecols > 0) hipLaunchKernelGGL((cheby_pair_edge_kernel<VAR, false, SM, IP, true>), dim3(egrid_r), dim3(64), 0, g_stream, *L, A); \
      PAIR_NW_SWITCH( \
      static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)cheby_pair_kernel<VAR, NWC, false, SM, NRW, IP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)NWC * 6 * 64 * sizeof(p2)))); once = true; } \
      hipLaunchKernelGGL((cheby_pair_kernel<VAR, NWC, false, SM, NRW, IP, true>), dim3(grid), dim3(64, NWC), lds, g_stream, *L, A); ) }
#define PAIR_LAUNCH_REMOTE(VAR, SM) { if (L->dim % 128 != 0) PAIR_LAUNCH_REMOTE_IP(VAR, SM, true, false) else if (interp) PAIR_LAUNCH_REMOTE_IP(VAR, SM, false, true) else PAIR_LAUNCH_REMOTE_IP(VAR, SM, false, false) }
#define PAIR_LAUNCH(VAR, C32, SM) { \
      A.edge_blocks = ((A.Dj + 63) / 64) * A.Dk * 2 * (A.tiles_i - 1); const dim3 egrid(A.edge_blocks > 0 ? grid_for(A.edge_blocks, &A.edge_per_xcd) : 1); \
      if (remote) PAIR_LAUNCH_REMOTE(VAR, SM) \
      else if (interp) { \
        if (A.tiles_i > 1) hipLaunchKernelGGL((cheby_pair_edge_kernel<VAR, C32, SM, true>), egrid, dim3(64), 0, g_stream, *L, A); \
        if (L->dim % 128 == 0) PAIR_LAUNCH2(VAR, C32, SM, false, true) else PAIR_LAUNCH2(VAR, C32, SM, true, true) \
      } else { \
        if (A.tiles_i > 1) hipLaunchKernelGGL((cheby_pair_edge_kernel<VAR, C32, SM, false>), egrid, dim3(64), 0, g_stream, *L, A); \
        if (L->dim % 128 == 0) PAIR_LAUNCH2(VAR, C32, SM, false, false) else PAIR_LAUNCH2(VAR, C32, SM, true, false) \
      } }
#define PAIR_CASE(VAR) case VAR: \
    if (gsrb) { if (c32_base) PAIR_LAUNCH(VAR, true, PAIR_GSRB) else PAIR_LAUNCH(VAR, false, PAIR_GSRB) } \
    else      { if (c32_base) PAIR_LAUNCH(VAR, true, PAIR_CHEBY) else PAIR_LAUNCH(VAR, false, PAIR_CHEBY) } break;
  switch (variant) {
    PAIR_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ)
    PAIR_CASE(HPGMG_HIP_7PT_VC_POISSON)
    PAIR_CASE(HPGMG_HIP_7PT_CC)
    default: return record_error(hipErrorInvalidValue, "smooth_cheby_pair: variant");
  }
#undef PAIR_CASE
#undef PAIR_LAUNCH
#undef PAIR_LAUNCH2
#undef PAIR_NW_SWITCH
#undef PAIR_LAUNCH_REMOTE
#undef PAIR_LAUNCH_REMOTE_IP
  if (part != 1) { g_pair_launches++; if (remote) g_pair_remote_launches++; }      // the two parts of a launch count once (part 2 is never empty: it holds the workgroups at the remote faces)
  profile_end(prof, 2 * cells, part == 1);           // one launch = two sweeps over every cell
  HPGMG_LAUNCH_CHECK("cheby_pair_kernel");
  return 0;
}
int hpgmg_hip_smooth_cheby_pair(const hpgmg_hip_level *L, int variant, double *const *scr_base, const float *const *c32_base,
                                int x0_scr, int x0_id, int xm1_scr, int xm1_id, int out1_scr, int out1_id, int out2_scr, int out2_id,
                                int rhs_id, double a, double b, double h2inv, double c1a, double c2a, double c1b, double c2b) {
  return smooth_pair(L, variant, 0, 0, scr_base, c32_base, x0_scr, x0_id, xm1_scr, xm1_id, out1_scr, out1_id, out2_scr, out2_id, rhs_id, a, b, h2inv, c1a, c2a, c1b, c2b);
}
void hpgmg_hip_pair_fold_interpolation(const hpgmg_hip_level *Lc, int coarse_id, double prescale) {
  g_pair_interp_level = Lc; g_pair_interp_id = coarse_id; g_pair_interp_prescale = prescale;
}
void hpgmg_hip_pair_set_halo(const int brick_boxes[3], const int remote_face[6], const double *deep, const double *deep_beta) {
  for (int d = 0; d < 3; d++) g_pair_brick[d] = brick_boxes[d];
  for (int d = 0; d < 6; d++) g_pair_rem[d] = remote_face[d];
  g_pair_deep = deep; g_pair_deep_beta = deep_beta; g_pair_halo_set = true;
}
void hpgmg_hip_pair_discard_x1(void) { g_pair_discard_x1 = 1; }
#ifdef HPGMG_EXP_TIMELINE
void hpgmg_hip_exp_timeline(void *buf) { g_exp_timeline = (double *)buf; }
#endif
void hpgmg_hip_pair_launch_counts(long long out[2]) { out[0] = g_pair_launches; out[1] = g_pair_remote_launches; }

// ---- the halo of a sweep pair across rank boundaries: one pack launch, one grouped send/recv, one unpack launch ----
static const hpgmg_hip_level *g_halo_fold_level = nullptr;      // consumed by the next pack (hpgmg_hip_pair_halo_fold_interpolation)
static int g_halo_fold_id = 0;
static double g_halo_fold_prescale = 1.0;
static int pair_halo_move(bool unpack, const hpgmg_hip_level *L, double *const *scr_base, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id,
                          const hpgmg_hip_halo_entry *entries, int n, double *buf, double *deep, double *deep_beta) {
  const hpgmg_hip_level *fold = unpack ? nullptr : g_halo_fold_level;
  if (!unpack) g_halo_fold_level = nullptr;
  HPGMG_SKIP_IF_REPLAY();
  if (n <= 0) return 0;
  HaloRefs R = {}; R.x0 = VecRef{x0_scr, x0_id}; R.xm1 = VecRef{xm1_scr, xm1_id}; R.rhs_id = rhs_id; R.scr_base = scr_base;
  if (fold) { R.fold = 1; R.Lc = *fold; R.coarse_id = g_halo_fold_id; R.prescale = g_halo_fold_prescale; }
  const int slabs = (L->dim * L->dim + 4095) / 4096;                 // a face of dim^2 values: 16 values per lane
  if (unpack) hipLaunchKernelGGL((pair_halo_kernel<true>), dim3(n, slabs), dim3(256), 0, g_stream, *L, R, entries, buf, deep, deep_beta);
  else        hipLaunchKernelGGL((pair_halo_kernel<false>), dim3(n, slabs), dim3(256), 0, g_stream, *L, R, entries, buf, deep, deep_beta);
  HPGMG_LAUNCH_CHECK("pair_halo_kernel");
  return 0;
}
// the NEXT hpgmg_hip_pair_halo_pack adds the coarse parent to the x0 values it packs (a smooth() with interpolation_vcycle folded into its first sweep pair)
void hpgmg_hip_pair_halo_fold_interpolation(const hpgmg_hip_level *Lc, int coarse_id, double prescale) { g_halo_fold_level = Lc; g_halo_fold_id = coarse_id; g_halo_fold_prescale = prescale; }
int hpgmg_hip_pair_halo_pack(const hpgmg_hip_level *L, double *const *scr_base, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id,
                             const hpgmg_hip_halo_entry *entries, int n, double *sendbuf) {
  return pair_halo_move(false, L, scr_base, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id, entries, n, sendbuf, nullptr, nullptr);
}
int hpgmg_hip_pair_halo_unpack(const hpgmg_hip_level *L, double *const *scr_base, int x0_scr, int x0_id, int xm1_scr, int xm1_id, int rhs_id,
                               const hpgmg_hip_halo_entry *entries, int n, double *recvbuf, double *deep, double *deep_beta) {
  return pair_halo_move(true, L, scr_base, x0_scr, x0_id, xm1_scr, xm1_id, rhs_id, entries, n, recvbuf, deep, deep_beta);
}

// two consecutive in-place GSRB half sweeps (sweep, sweep + 1): x2 -> out2; the scratch vector `edge_scr_id` receives the
// few x1 values the kernel exchanges across 128-cell tile edges
int hpgmg_hip_smooth_gsrb_pair(const hpgmg_hip_level *L, int variant, double *const *scr_base, const float *const *c32_base,
                               int x0_scr, int x0_id, int edge_scr_id, int out2_scr, int out2_id, int rhs_id,
                               double a, double b, double h2inv, int sweep) {
  return smooth_pair(L, variant, 1, sweep, scr_base, c32_base, x0_scr, x0_id, x0_scr, x0_id, 1, edge_scr_id, out2_scr, out2_id, rhs_id, a, b, h2inv, 0.0, 0.0, 0.0, 0.0);
}

}  // extern "C"

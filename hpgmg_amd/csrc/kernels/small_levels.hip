// small_levels.hip -- single-workgroup kernels for the levels a chip cannot be filled with (27-point / fv2 / fv4 plugins, and the reference's
// host-driven Krylov solver on any plugin): a whole smooth() / residual() of a level of one box on an LDS image (small_level_kernel), the BiCGStab
// bottom solve (solvers/bicgstab.c:14-97: bottom_bicgstab_kernel), queued BLAS-1 / operator calls ending in a scalar (small_ops_kernel) and the rest
// of a V-cycle below a level of one box (mg.c:1145-1164: small_vtail_kernel).  A translation unit of its own (split from stencil.hip in round 4).
#include "stencil_direct.hpp"

namespace hpgmg {
// ---------------------------------------------------------------------------------------------
// Small levels (<= 4096 cells) of the 27-point / fv2 / fv4 plugins: a whole smooth() -- per sweep exchange_boundary (local copies),
// apply_BCs and the stencil, the three launches operators.27pt.c / .fv2.c / .fv4.c sequence per sweep -- or a whole residual() /
// apply_op() as ONE single-workgroup launch, with barriers where the per-operator path has kernel boundaries.  On these levels a
// launch costs more than the work (fv4: 18 launches of ~6 us per smooth()); the arithmetic is the same entry routines and the same
// per-cell expressions as the streaming kernels, so results stay bit-identical.  (The 7-point plugin has its own, LDS-resident form
// of this idea: tail.hip.)
struct SmallArgs {
  int mode, sweeps, x_id, rhs_id, res_id, out_of_place, bc_kind, zero_first;   // bc_kind: 0 none (periodic), 1 p1, 2 p2, 3 v2, 4 v4
  double a, b, h2inv, c1[8], c2[8];
  const blockCopy_type *copy_list; int n_copy;
  const blockCopy_type *bc_list; int n_bc;
  int lds_resident;                 // a level of ONE box whose vectors fit the LDS: work on an image of the box there (see the kernel)
  unsigned long long *timeline;     // experiment builds (-DHPGMG_EXP_TIMELINE): lane 0 records the clock at the phase boundaries
};
// LDS-resident form (round 3): out of global memory every boundary entry and every stencil read of this one workgroup is a round trip to
// the L2 that nothing hides (measured: slower than the dozen launches it replaces, even on a level of one box).  For a level of ONE box
// the vectors the operator touches -- x, VECTOR_TEMP, rhs, Dinv, alpha, beta_i/j/k, the result -- are copied into LDS with the box's own
// padded layout, the level descriptor is pointed at that image (a one-entry box table in LDS, vector ids renumbered to slots), the very same
// entry routines and per-cell expressions run on it, and the vectors written go back to memory at the end: bit-identical by construction.
constexpr int kSmallSlots = 9;
typedef double __attribute__((address_space(3))) *lds_dptr;       // a pointer into LDS, typed as such: ds_read / ds_write, not FLAT
typedef const int __attribute__((address_space(3))) *lds_iptr;
// a pointer into global memory, typed as such: through a generic pointer a load is a FLAT instruction, which may address LDS -- the compiler
// then keeps every such load and every LDS store of a copy loop in program order, one round trip to memory per element
typedef double __attribute__((address_space(1))) *gbl_dptr;
typedef const double __attribute__((address_space(1))) *gbl_cdptr;
// the eight words of a boundary entry that the entry routines read (subtype, dim, read.box / i / j / k), kept in LDS by the single-workgroup
// kernels: read from the level's list in memory, the descriptor was a round trip per entry and half sweep
// (n <= 32.)  The entries are stored SORTED by kind -- corners, edges, faces; their order is immaterial, every entry reads the interior and writes
// ghost cells of its own -- and words[256..258] hold the three counts: the packed dispatch of apply_BCs_v4 below hands out lanes by kind.
constexpr int kBcWords = 32 * 8 + 4;
__device__ __forceinline__ void lds_bc_words_fill(int *words, const blockCopy_type *list, int n, int tid, int nth, bool sorted) {
  if (!sorted) {                                                  // every lane fetches a word (the sorted form is a wave's serial work: only where it pays)
    for (int t = tid; t < 8 * n; t += nth) {
      const blockCopy_type &g = list[t >> 3];
      const int f = t & 7;
      words[t] = f == 0 ? g.subtype : f == 1 ? g.dim.i : f == 2 ? g.dim.j : f == 3 ? g.dim.k : f == 4 ? g.read.box : f == 5 ? g.read.i : f == 6 ? g.read.j : g.read.k;
    }
    return;
  }
  if (tid >= 64) return;                                          // the first wave: a lane per entry, the positions by ballot
  const bool have = tid < n;
  int w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nn = 0;
  if (have) {
    const blockCopy_type &g = list[tid];
    w[0] = g.subtype; w[1] = g.dim.i; w[2] = g.dim.j; w[3] = g.dim.k; w[4] = g.read.box; w[5] = g.read.i; w[6] = g.read.j; w[7] = g.read.k;
    nn = (w[0] % 3 != 1) + ((w[0] % 9) / 3 != 1) + (w[0] / 9 != 1);
  }
  const unsigned long long m3 = __ballot(have && nn == 3), m2 = __ballot(have && nn == 2), m1 = __ballot(have && nn < 2);
  const unsigned long long below = (1ull << tid) - 1ull;
  const int n3 = __popcll(m3), n2 = __popcll(m2);
  const int pos = (nn == 3) ? __popcll(m3 & below) : (nn == 2) ? n3 + __popcll(m2 & below) : n3 + n2 + __popcll(m1 & below);
  if (have) {
#pragma unroll
    for (int f = 0; f < 8; f++) words[8 * pos + f] = w[f];
  }
  if (tid == 0) { words[256] = n3; words[257] = n2; words[258] = __popcll(m1); }
}
__device__ __forceinline__ blockCopy_type lds_bc_entry(const int *words, int e) {
  const lds_iptr w = (lds_iptr)words + 8 * e;
  blockCopy_type en;
  en.subtype = w[0]; en.dim.i = w[1]; en.dim.j = w[2]; en.dim.k = w[3]; en.read.box = w[4]; en.read.i = w[5]; en.read.j = w[6]; en.read.k = w[7];
  return en;
}
// The sweeps of one launch.  RES: the vectors live in the LDS image (`image`; vector "ids" are slots of it, the only box is box 0) and every
// access to them is an LDS instruction -- through generic pointers each was a FLAT access, and a corner entry of apply_BCs_v4 (64 dependent
// reads by one lane) or the 55 reads of a stencil took microseconds: 6.1 + 3.8 us per half sweep of an 8^3 level, 66 us per smooth().
// coef(s, c1, c2): the Chebyshev / Jacobi coefficients of sweep s (a functor: the caller knows where they live -- kernel arguments, memory)
template <int V, bool RES, typename CoefFn>
__device__ __forceinline__ void small_level_run(const hpgmg_hip_level &L, const SmallArgs &A, double *image, const blockCopy_type *bc_entries, const int *bc_words, const int *ids, unsigned long long *tl, int &tl_n, CoefFn coef) {
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride, per_box = dim * dim * dim, total = per_box * L.num_boxes;
  const size_t vol = (size_t)L.volume, first = (size_t)L.ghosts * (size_t)(1 + jS + kS);
  // origin of vector `id` of box `box`: a slot of the image (LDS) or the level's own storage
  auto vo = [&](int box, int id) {
    if constexpr (RES) { (void)box; return (lds_dptr)image + ((size_t)id * vol + first); }
    else return vec_origin(L, box, id);
  };
  // boundary entry e: from the level's list, or (RES, bc_words != 0) from the eight words of it that the entry routines read, kept in LDS
  auto entry = [&](int e) {
    if constexpr (RES) {
      if (bc_words) return lds_bc_entry(bc_words, e);
    }
    return bc_entries[e];
  };
  // the colour of cell (0,0,0) of the one box of an image (read once: every cell of every sweep asked the level for it)
  int low_parity = 0;
  if constexpr (RES) low_parity = L.box_low[0] ^ L.box_low[1] ^ L.box_low[2];
  const int x_id = ids[0], temp_id = ids[1], rhs_id = ids[2], dinv_id = ids[3], al_id = ids[4], bi_id = ids[5], bj_id = ids[6], bk_id = ids[7], res_id = ids[8];
#ifdef HPGMG_EXP_TIMELINE
#define SL_MARK() do { if (tl && threadIdx.x == 0 && tl_n < 250) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SL_MARK() do { (void)tl; (void)tl_n; } while (0)
#endif
  // (an image is one box of at most 512 cells, a cell per lane: its coordinates are worked out once, not per sweep -- four integer divisions)
  const int own_k = tid / (dim * dim), own_j = (tid / dim) % dim, own_i = tid % dim;
  for (int s = 0; s < A.sweeps; s++) {
    double cs1 = 0.0, cs2 = 0.0;
    coef(s, cs1, cs2);
    int src = x_id, dst = res_id;
    if (A.mode == MODE_CHEBY || A.mode == MODE_JACOBI || (A.mode == MODE_GSRB && A.out_of_place)) { src = (s & 1) ? temp_id : x_id; dst = (s & 1) ? x_id : temp_id; }
    else if (A.mode == MODE_GSRB) { src = x_id; dst = x_id; }
    // exchange_boundary(src): box -> box copies (blockCopy.c:6-105); an image is one box: nothing to copy
    if (!RES) { for (int e = wave; e < A.n_copy; e += nwaves) copy_entry<false>(L, src, A.copy_list[e], 0.0, lane, 64); __syncthreads(); }
    SL_MARK();
    // apply_BCs(src)
    if (A.bc_kind && A.zero_first) { for (int e = wave; e < A.n_bc; e += nwaves) { const blockCopy_type en = entry(e); bc_zero_entry_at(vo(en.read.box, src), L, en, lane, 64); } __syncthreads(); }
    bool packed = false;
    if constexpr (RES) packed = (bc_words != nullptr && A.bc_kind == 4);
    if (packed) {
      // apply_BCs_v4 with the lanes handed out by kind: 16 per corner, 32 per edge, 64 per face -- the 26 entries of a box are 896 lanes of
      // work, one round of a 1024-lane workgroup (two of a 512-lane one) instead of a wave per entry (two / four rounds)
      const lds_iptr cnt = (lds_iptr)bc_words + 256;
      const int n3 = cnt[0], n2 = cnt[1], n1 = cnt[2], lim3 = n3 * 16, lim2 = lim3 + n2 * 32, demand = lim2 + n1 * 64;
      for (int g0 = 0; g0 < demand; g0 += (int)blockDim.x) {
        const int g = g0 + tid;
        if (g < lim3)        bc_v4_entry_packed(vo(0, src), L, lds_bc_entry(bc_words, g >> 4), g & 15, lane);
        else if (g < lim2)   bc_v4_entry_packed(vo(0, src), L, lds_bc_entry(bc_words, n3 + ((g - lim3) >> 5)), (g - lim3) & 31, lane);
        else if (g < demand) bc_v4_entry_packed(vo(0, src), L, lds_bc_entry(bc_words, n3 + n2 + ((g - lim2) >> 6)), (g - lim2) & 63, lane);
      }
    }
    for (int e = wave; e < A.n_bc && !packed; e += nwaves) {
      const blockCopy_type en = entry(e);
      if (A.bc_kind == 1) bc_p1_entry_at(vo(en.read.box, src), L, en, lane, 64);
      else if (A.bc_kind == 2) bc_p2_entry_at(vo(en.read.box, src), L, en, lane, 64);
      else if (A.bc_kind == 3) bc_v2_entry_at(vo(en.read.box, src), L, en, lane, 64);
      else if (A.bc_kind == 4) bc_v4_entry_at(vo(en.read.box, src), L, en, lane, 64);
    }
    __syncthreads();
    SL_MARK();
    // the stencil over every cell (same expressions as stencil_direct_kernel / stencil27_kernel)
    for (int t = tid; t < total; t += (int)blockDim.x) {
      int box = 0, i = own_i, j = own_j, k = own_k;
      if (!RES || total > (int)blockDim.x) { box = t / per_box; const int r = t - box * per_box; k = r / (dim * dim); j = (r / dim) % dim; i = r % dim; }
      const int ijk = i + j * jS + k * kS;
      auto x = vo(box, src);
      auto out = vo(box, dst);
      const double xc = x[ijk];
      bool update = true;
      if (A.mode == MODE_GSRB) {
        const int lp = RES ? low_parity : (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2]);
        update = (((i ^ j ^ k ^ lp ^ s) & 1) == 0);
      }
      if (!update) { if (A.out_of_place) out[ijk] = xc; continue; }
      double Ax;
      if (k27) {
        const plane9 m = load_plane(x + (ijk - kS), jS), c = load_plane(x + ijk, jS), p = load_plane(x + (ijk + kS), jS);
        Ax = apply_op_27pt(m, c, p, A.a, A.b, A.h2inv);
      } else {
        auto none = vo(box, src); none = nullptr;
        // (not the batched form here: with 1024 lanes a lane has 128 registers, the batches then spill, and what the stencil phase gains the
        // boundary phase loses to the reloads -- measured 1.44 + 1.9 us against 1.8 + 1.7 us per half sweep)
        Ax = apply_op_direct<V, decltype(x), decltype(none), false>(x, kHelm ? vo(box, al_id) : none, kVC ? vo(box, bi_id) : none, kVC ? vo(box, bj_id) : none, kVC ? vo(box, bk_id) : none,
                                                                   ijk, jS, kS, A.a, A.b, A.h2inv);
      }
      if (A.mode == MODE_APPLY) { out[ijk] = Ax; continue; }
      const double rhs = vo(box, rhs_id)[ijk];
      if (A.mode == MODE_RESIDUAL) { out[ijk] = rhs - Ax; continue; }
      const double dinv = vo(box, dinv_id)[ijk];
      if (A.mode == MODE_CHEBY)      { const double xnm1 = out[ijk]; out[ijk] = xc + cs1 * (xc - xnm1) + cs2 * dinv * (rhs - Ax); }
      else if (A.mode == MODE_GSRB)  { out[ijk] = xc + dinv * (rhs - Ax); }
      else                           { out[ijk] = xc + cs2 * dinv * (rhs - Ax); }
    }
    __syncthreads();
    SL_MARK();
  }
#undef SL_MARK
}
template <int V>
__global__ __launch_bounds__(1024) void small_level_kernel(const hpgmg_hip_level L, const SmallArgs A) {
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  extern __shared__ double small_lds[];
  const int tid = (int)threadIdx.x;
  int tl_n = 0;
  unsigned long long *tl = nullptr;
#ifdef HPGMG_EXP_TIMELINE
  tl = A.timeline;
  if (tl && tid == 0) tl[tl_n++] = __builtin_amdgcn_s_memrealtime();
#endif
  if (!A.lds_resident) return;                                    // (the launcher only starts it on one box whose vectors fit the LDS)
  // ---- one box, its vectors in LDS: copy in, run, copy what was written back (ghost zones included: the boundary entries filled them)
  const bool smooth = (A.mode == MODE_CHEBY || A.mode == MODE_JACOBI || A.mode == MODE_GSRB);
  const bool uses_temp = smooth && !(A.mode == MODE_GSRB && !A.out_of_place);
  int slot_of[kSmallSlots];                                        // level vector id held in each slot (-1: unused)
  slot_of[0] = A.x_id; slot_of[1] = uses_temp ? VECTOR_TEMP : -1; slot_of[2] = (A.mode == MODE_APPLY) ? -1 : A.rhs_id; slot_of[3] = smooth ? VECTOR_DINV : -1;
  slot_of[4] = kHelm ? VECTOR_ALPHA : -1; slot_of[5] = kVC ? VECTOR_BETA_I : -1; slot_of[6] = kVC ? VECTOR_BETA_J : -1; slot_of[7] = kVC ? VECTOR_BETA_K : -1;
  slot_of[8] = (smooth || A.res_id == A.x_id) ? -1 : A.res_id;
  const size_t vol = (size_t)L.volume;
  lds_dptr img = (lds_dptr)small_lds;
  // the boundary entries of the box (26 at most) wait in LDS too: read from memory, the descriptor was a round trip per entry and half sweep
  __shared__ int s_bc[kBcWords];
  const bool bc_in_lds = A.n_bc <= 32;
  if (bc_in_lds) lds_bc_words_fill(s_bc, A.bc_list, A.n_bc, tid, (int)blockDim.x, A.bc_kind == 4);
#pragma unroll
  for (int q = 0; q < kSmallSlots; q++) {
    if (slot_of[q] < 0) continue;
    // (a result vector that is written in full needs no load, but its ghost zone must come back as it was: copy it all the same)
    const gbl_cdptr g = (gbl_cdptr)(L.box_base[0] + (size_t)slot_of[q] * vol);
#pragma unroll 8
    for (int t = tid; t < (int)vol; t += (int)blockDim.x) img[(size_t)q * vol + t] = g[t];      // unrolled: eight loads in flight per lane, not one
  }
  __syncthreads();
#ifdef HPGMG_EXP_TIMELINE
  if (tl && tid == 0) tl[tl_n++] = __builtin_amdgcn_s_memrealtime();
#endif
  const int ids[kSmallSlots] = { 0, 1, 2, 3, 4, 5, 6, 7, (smooth || A.res_id == A.x_id) ? 0 : 8 };
  small_level_run<V, true>(L, A, small_lds, A.bc_list, bc_in_lds ? (const int *)s_bc : nullptr, ids, tl, tl_n, [&](int s, double &c1, double &c2) { c1 = A.c1[s]; c2 = A.c2[s]; });
#pragma unroll
  for (int q = 0; q < kSmallSlots; q++) {
    const bool written = smooth ? (q == 0 || (q == 1 && slot_of[1] >= 0)) : (q == 0 || q == 8);      // x's ghost zone was filled too
    if (!written || slot_of[q] < 0) continue;
    const gbl_dptr g = (gbl_dptr)(L.box_base[0] + (size_t)slot_of[q] * vol);
#pragma unroll 8
    for (int t = tid; t < (int)vol; t += (int)blockDim.x) g[t] = img[(size_t)q * vol + t];
  }
#ifdef HPGMG_EXP_TIMELINE
  if (tl && tid == 0) { tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); tl[255] = (unsigned long long)tl_n; }
#endif
}


// ---------------------------------------------------------------------------------------------
// Bottom solve of the 27-point / fv2 / fv4 plugins: diagonally preconditioned BiCGStab (solvers/bicgstab.c:14-97) on a bottom level of ONE
// box as one single-workgroup launch.  Driven from the host it is ~25 launches and ~6 host round trips (the dot products and norms) per
// iteration on a level of 8 cells: 1.5 ms of a 33 ms fv4 F-cycle at 512^3, a quarter of one at 128^3.  One cell per lane, every vector a
// register; the vector the operator is applied to passes through an image of the padded box in LDS, on which the boundary entries of the
// level run (the same entry routines as the streaming kernels) before the stencil (the same per-cell expression).  The operation sequence,
// the expression of every BLAS-1 step (misc.c: c = sa*a + sb*b, c = s*a*b), the break-down tests and the order of the sums -- one partial
// per dim x 8 x 8 tile accumulated k, j, i, partials added in tile order (misc.c:261-269) -- are those of host/solvers.c, so the iterates,
// the iteration count and the coarse correction are bit-identical to the host-driven solve (the 7-point plugin's form of this: tail.hip).
struct BottomArgs {
  int e_id, R_id, krylov_base, bc_kind, zero_first, n_bc;
  double a, b, h2inv, want;
  const blockCopy_type *bc_list;
  int *krylov_iterations;
};
template <int V>
__device__ __forceinline__ void bottom_bicgstab_body(const hpgmg_hip_level &L, const BottomArgs &A) {
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  extern __shared__ double bb_lds[];                               // images of the padded box (5 x L.volume doubles: v, alpha, beta_i/j/k), then the reduction scratch
  __shared__ int s_bc[kBcWords];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride, total = dim * dim * dim;
  const bool active = tid < total;
  const int ci = tid % dim, cj = (tid / dim) % dim, ck = tid / (dim * dim), ijk = ci + cj * jS + ck * kS;
  // Every access to the images is an LDS instruction (pointers typed as such; through generic pointers each was a FLAT access), the
  // coefficients wait there too (read from memory they were a round trip to the L2 per apply), and so do the boundary entries.
  const int vol = L.volume, first = L.ghosts * (1 + jS + kS);
  const lds_dptr img = (lds_dptr)bb_lds;
  const lds_dptr xi = img + first;
  const lds_dptr alpha = img + (vol + first), bi = img + (2 * vol + first), bj = img + (3 * vol + first), bk = img + (4 * vol + first);
  const lds_dptr scr = img + 5 * vol;
  const bool bc_in_lds = A.n_bc <= 32;
  if (bc_in_lds) lds_bc_words_fill(s_bc, A.bc_list, A.n_bc, tid, (int)blockDim.x, false);
  {
    const gbl_cdptr g_al = (gbl_cdptr)(L.box_base[0] + (size_t)VECTOR_ALPHA * vol), g_bi = (gbl_cdptr)(L.box_base[0] + (size_t)VECTOR_BETA_I * vol);
    const gbl_cdptr g_bj = (gbl_cdptr)(L.box_base[0] + (size_t)VECTOR_BETA_J * vol), g_bk = (gbl_cdptr)(L.box_base[0] + (size_t)VECTOR_BETA_K * vol);
    for (int t = tid; t < vol; t += (int)blockDim.x) {
      img[t] = 0.0;
      if (kHelm) img[vol + t] = g_al[t];
      if (kVC) { img[2 * vol + t] = g_bi[t]; img[3 * vol + t] = g_bj[t]; img[4 * vol + t] = g_bk[t]; }
    }
  }
  __syncthreads();
  const int r0_id = A.krylov_base, r_id = r0_id + 1, p_id = r0_id + 2, q_id = r0_id + 3, s_id = r0_id + 4, t_id = r0_id + 5, Ap_id = r0_id + 6, As_id = r0_id + 7;
  double x = 0, r0 = 0, r = 0, p = 0, q = 0, sv = 0, tv = 0, Ap = 0, As = 0, tmp = 0, dinv = 0, rhs = 0;
  if (active) {
    x = vec_origin(L, 0, A.e_id)[ijk]; rhs = vec_origin(L, 0, A.R_id)[ijk]; dinv = vec_origin(L, 0, VECTOR_DINV)[ijk];
    r0 = vec_origin(L, 0, r0_id)[ijk]; r = vec_origin(L, 0, r_id)[ijk]; p = vec_origin(L, 0, p_id)[ijk]; q = vec_origin(L, 0, q_id)[ijk];
    sv = vec_origin(L, 0, s_id)[ijk]; tv = vec_origin(L, 0, t_id)[ijk]; Ap = vec_origin(L, 0, Ap_id)[ijk]; As = vec_origin(L, 0, As_id)[ijk];
    tmp = vec_origin(L, 0, VECTOR_TEMP)[ijk];
  }
  auto entry = [&](int e) { if (bc_in_lds) return lds_bc_entry(s_bc, e); return A.bc_list[e]; };
  // apply_op(v): exchange_boundary (nothing to exchange: one box) + apply_BCs + the stencil (operators.*.c: apply_op)
  auto apply = [&](double v) -> double {
    if (active) xi[ijk] = v;
    __syncthreads();
    if (A.bc_kind && A.zero_first) { for (int e = wave; e < A.n_bc; e += nwaves) bc_zero_entry_at(xi, L, entry(e), lane, 64); __syncthreads(); }
    for (int e = wave; e < A.n_bc; e += nwaves) {
      const blockCopy_type en = entry(e);
      if (A.bc_kind == 1) bc_p1_entry_at(xi, L, en, lane, 64);
      else if (A.bc_kind == 2) bc_p2_entry_at(xi, L, en, lane, 64);
      else if (A.bc_kind == 3) bc_v2_entry_at(xi, L, en, lane, 64);
      else if (A.bc_kind == 4) bc_v4_entry_at(xi, L, en, lane, 64);
    }
    __syncthreads();
    double Ax = 0.0;
    if (active) {
      if (k27) {
        const plane9 m = load_plane(xi + (ijk - kS), jS), c = load_plane(xi + ijk, jS), pp = load_plane(xi + (ijk + kS), jS);
        Ax = apply_op_27pt(m, c, pp, A.a, A.b, A.h2inv);
      } else {
        Ax = apply_op_direct<V, lds_dptr, lds_dptr, true>(xi, alpha, bi, bj, bk, ijk, jS, kS, A.a, A.b, A.h2inv);
      }
    }
    __syncthreads();
    return Ax;
  };
  // dot(a, b): misc.c:230-280 -- per tile of 8 x 8 rows a partial accumulated k, j, i; the partials added in tile order
  const int tiles_side = (dim + BLOCKCOPY_TILE_J - 1) / BLOCKCOPY_TILE_J, ntiles = tiles_side * ((dim + BLOCKCOPY_TILE_K - 1) / BLOCKCOPY_TILE_K);
  auto dot = [&](double va, double vb) -> double {
    if (active) scr[tid] = va * vb;
    __syncthreads();
    if (tid < ntiles) {
      const int k0 = (tid / tiles_side) * BLOCKCOPY_TILE_K, j0 = (tid % tiles_side) * BLOCKCOPY_TILE_J;
      const int k1 = min(k0 + BLOCKCOPY_TILE_K, dim), j1 = min(j0 + BLOCKCOPY_TILE_J, dim);
      double acc = 0.0;
      for (int k = k0; k < k1; k++) for (int j = j0; j < j1; j++) { const lds_dptr row = scr + dim * (j + dim * k); for (int i = 0; i < dim; i++) acc += row[i]; }
      scr[512 + tid] = acc;
    }
    __syncthreads();
    if (tid == 0) { double sum = 0.0; for (int t = 0; t < ntiles; t++) sum += scr[512 + t]; scr[1024] = sum; }
    __syncthreads();
    const double v = scr[1024];
    __syncthreads();
    return v;
  };
  auto norm = [&](double v) -> double {                            // max |v| (misc.c:303-349): exact under any order
    double m = 0.0;
    if (active) { const double f = fabs(v); m = (f > m) ? f : m; }
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(m, off, 64); m = (o > m) ? o : m; }
    if (lane == 0) scr[1032 + wave] = m;
    __syncthreads();
    m = scr[1032];
    for (int w = 1; w < nwaves; w++) m = (scr[1032 + w] > m) ? scr[1032 + w] : m;
    __syncthreads();
    return m;
  };
  const double want = A.want;
  int it = 0;
  // host/solvers.c bicgstab(), Dirichlet (no mean to remove)
  r0 = rhs - apply(x);
  r = 1.0 * r0;
  p = 1.0 * r0;
  {
    double rho = dot(r, r0);
    const double r0_norm = norm(r);
    if (!(rho == 0.0 || r0_norm == 0.0)) {
      while (it < 200) {
        it++;
        q = 1.0 * dinv * p;
        Ap = apply(q);
        const double Ap_r0 = dot(Ap, r0);
        if (Ap_r0 == 0.0) break;
        const double al = rho / Ap_r0;
        if (__builtin_isinf(al)) break;
        x = 1.0 * x + al * q;
        sv = 1.0 * r + (-al) * Ap;
        const double s_norm = norm(sv);
        if (s_norm == 0.0 || s_norm < want * r0_norm) break;
        tv = 1.0 * dinv * sv;
        As = apply(tv);
        const double As_As = dot(As, As);
        const double As_s = dot(As, sv);
        if (As_As == 0.0) break;
        const double omega = As_s / As_As;
        if (omega == 0.0 || __builtin_isinf(omega)) break;
        x = 1.0 * x + omega * tv;
        r = 1.0 * sv + (-omega) * As;
        const double r_norm = norm(r);
        if (r_norm == 0.0 || r_norm < want * r0_norm) break;
        const double rho_new = dot(r, r0);
        if (rho_new == 0.0) break;
        const double beta = (rho_new / rho) * (al / omega);
        if (__builtin_isinf(beta)) break;
        tmp = 1.0 * p + (-omega) * Ap;
        p = 1.0 * r + beta * tmp;
        rho = rho_new;
      }
    }
  }
  if (active) {
    vec_origin(L, 0, A.e_id)[ijk] = x;
    vec_origin(L, 0, r0_id)[ijk] = r0; vec_origin(L, 0, r_id)[ijk] = r; vec_origin(L, 0, p_id)[ijk] = p; vec_origin(L, 0, q_id)[ijk] = q;
    vec_origin(L, 0, s_id)[ijk] = sv; vec_origin(L, 0, t_id)[ijk] = tv; vec_origin(L, 0, Ap_id)[ijk] = Ap; vec_origin(L, 0, As_id)[ijk] = As;
    vec_origin(L, 0, VECTOR_TEMP)[ijk] = tmp;
  }
  if (tid == 0 && A.krylov_iterations) *A.krylov_iterations += it;
}
template <int V>
__global__ __launch_bounds__(512) void bottom_bicgstab_kernel(const hpgmg_hip_level L, const BottomArgs A) { bottom_bicgstab_body<V>(L, A); }

// ---------------------------------------------------------------------------------------------
// A queue of BLAS-1 / operator calls on a level of ONE small box (<= 512 cells) as one single-workgroup launch, ending -- if the caller
// wants a value -- in the dot product or norm that made the host ask.  This is what a host-driven Krylov solver does on the bottom level
// (the reference's solvers/bicgstab.c through operators.h, "Route B"): per iteration ~18 launches of an 8-cell kernel and 6 scalars fetched;
// the plugin postpones the void operators and issues them together with the value-returning one: 6 launches.  Every operation is the
// expression of its own kernel: misc.c add_vectors c = sa*a + sb*b, mul_vectors c = s*a*b, scale_vector c = s*a (blas1.hip
// elementwise_kernel); apply_op / residual = apply_BCs on the operand with the level's own boundary entries, then the stencil; dot = the
// products summed k, j, i (one dim x 8 x 8 tile: the order of misc.c:261-269 and tile_sum_kernel); norm = max |a|.
enum { SO_ADD = 1, SO_MUL, SO_SCALE, SO_APPLY, SO_RESIDUAL, SO_DOT, SO_NORM };
constexpr int kSmallOpsMax = 12;
struct SmallOp { int kind, c, a, b; double sa, sb; };
struct SmallOpsArgs {
  int n, bc_kind, zero_first, n_bc;
  const blockCopy_type *bc_list;
  double a, b, h2inv;
  ResultSlot *result; unsigned long long seq;
  SmallOp op[kSmallOpsMax];
};
// a launch may end in TWO value-returning operations (the second one a guess of what the host asks next): value k goes to the k-th double of
// the slot's payload (value, then the word behind the sequence number), the LAST operation of the list publishes the sequence number
__device__ __forceinline__ void small_ops_value(ResultSlot *slot, int k, double v, bool last, unsigned long long seq) {
  double *second = reinterpret_cast<double *>(slot) + 2;
  if (k == 0) slot->value = v; else *second = v;
  if (last) { __threadfence_system(); __hip_atomic_store(&slot->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
template <int V>
__global__ __launch_bounds__(512) void small_ops_kernel(const hpgmg_hip_level L, const SmallOpsArgs A) {
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  __shared__ double part[512 + 8];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride, total = dim * dim * dim;
  const bool active = tid < total;
  const int ci = tid % dim, cj = (tid / dim) % dim, ck = tid / (dim * dim), ijk = ci + cj * jS + ck * kS;
  int nvalues = 0;                                                 // value-returning operations so far: the first goes to result->value, the second behind it
  for (int q = 0; q < A.n; q++) {
    const int kind = A.op[q].kind, idc = A.op[q].c, ida = A.op[q].a, idb = A.op[q].b;
    const double sa = A.op[q].sa, sb = A.op[q].sb;
    if (kind == SO_ADD)        { if (active) vec_origin(L, 0, idc)[ijk] = sa * vec_origin(L, 0, ida)[ijk] + sb * vec_origin(L, 0, idb)[ijk]; }
    else if (kind == SO_MUL)   { if (active) vec_origin(L, 0, idc)[ijk] = sa * vec_origin(L, 0, ida)[ijk] * vec_origin(L, 0, idb)[ijk]; }
    else if (kind == SO_SCALE) { if (active) vec_origin(L, 0, idc)[ijk] = sa * vec_origin(L, 0, ida)[ijk]; }
    else if (kind == SO_APPLY || kind == SO_RESIDUAL) {
      // exchange_boundary: one box, nothing to copy; apply_BCs on the operand, then the stencil (operators.*.c apply_op / residual)
      if (A.bc_kind && A.zero_first) { for (int e = wave; e < A.n_bc; e += nwaves) bc_zero_entry(L, ida, A.bc_list[e], lane, 64); __syncthreads(); }
      for (int e = wave; e < A.n_bc; e += nwaves) {
        if (A.bc_kind == 1) bc_p1_entry(L, ida, A.bc_list[e], lane, 64);
        else if (A.bc_kind == 2) bc_p2_entry(L, ida, A.bc_list[e], lane, 64);
        else if (A.bc_kind == 3) bc_v2_entry(L, ida, A.bc_list[e], lane, 64);
        else if (A.bc_kind == 4) bc_v4_entry(L, ida, A.bc_list[e], lane, 64);
      }
      __syncthreads();
      if (active) {
        const double *x = vec_origin(L, 0, ida);
        double Ax;
        if (k27) {
          const plane9 m = load_plane(x + (ijk - kS), jS), c = load_plane(x + ijk, jS), pp = load_plane(x + (ijk + kS), jS);
          Ax = apply_op_27pt(m, c, pp, A.a, A.b, A.h2inv);
        } else {
          const double *none = nullptr;
          Ax = apply_op_direct<V>(x, kHelm ? (const double *)vec_origin(L, 0, VECTOR_ALPHA) : none, kVC ? (const double *)vec_origin(L, 0, VECTOR_BETA_I) : none,
                                  kVC ? (const double *)vec_origin(L, 0, VECTOR_BETA_J) : none, kVC ? (const double *)vec_origin(L, 0, VECTOR_BETA_K) : none, ijk, jS, kS, A.a, A.b, A.h2inv);
        }
        vec_origin(L, 0, idc)[ijk] = (kind == SO_RESIDUAL) ? vec_origin(L, 0, idb)[ijk] - Ax : Ax;
      }
    } else if (kind == SO_DOT) {
      if (active) part[tid] = vec_origin(L, 0, ida)[ijk] * vec_origin(L, 0, idb)[ijk];
      __syncthreads();
      if (tid == 0) { double acc = 0.0; for (int t = 0; t < total; t++) acc += part[t]; small_ops_value(A.result, nvalues, 0.0 + acc, q == A.n - 1, A.seq); }   // (one tile: its partial added to 0.0, as tile_sum_kernel does)
      nvalues++;
    } else if (kind == SO_NORM) {
      double m = 0.0;
      if (active) { const double f = fabs(vec_origin(L, 0, ida)[ijk]); m = (f > m) ? f : m; }
      for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(m, off, 64); m = (o > m) ? o : m; }
      if (lane == 0) part[512 + wave] = m;
      __syncthreads();
      if (tid == 0) { for (int w = 1; w < nwaves; w++) m = (part[512 + w] > m) ? part[512 + w] : m; small_ops_value(A.result, nvalues, m, q == A.n - 1, A.seq); }
      nvalues++;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The rest of a V-cycle below a level of ONE box, 27-point / fv2 / fv4 plugins (mg.c:1133-1166), as one single-workgroup launch.  Driven
// operator by operator a visit of such a level is 4 launches on the way down (smooth -- itself one launch, above --, the boundary fill and
// the stencil of residual(), restriction + zero_vector) and 3 on the way up, ~5 us each for work of a microsecond.  Here the levels of
// the chain take turns in LDS: an image of the level's box (nine vectors, padded layout) is loaded, small_level_run() smooths it and
// forms the residual, the restriction goes straight from the image to the coarse level's right-hand side in memory, the written vectors go
// back; the bottom solve is bottom_bicgstab_body(); on the way up the coarse correction is staged behind the image, its boundary
// conditions are applied there and the tensor rule adds it to the image's x before the smoother runs.  Every per-cell expression is the one
// of the per-operator kernels (restrict_entry / interp_tensor_kernel in blocks.hip), so the result is bit-identical to the launches it
// replaces (the coarse correction's ghost zone in MEMORY is left as it was: every reader fills it first).
template <int V>
__global__ __launch_bounds__(512) void small_vtail_kernel(const hpgmg_hip_small_tail_args *__restrict__ Tp, unsigned long long *tl) {
#ifdef HPGMG_EXP_TIMELINE
  int tl_n = 0;
#define VT_MARK() do { if (tl && threadIdx.x == 0 && tl_n < 250) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VT_MARK() do { (void)tl; } while (0)
#endif
  VT_MARK();
#ifdef HPGMG_EXP_TIMELINE
  if (tl && threadIdx.x == 0) tl[253] = __builtin_amdgcn_s_memtime();      // shader clock against the 100 MHz real-time marks
#endif
  constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC && !k27);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ || V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  constexpr int ORDER = k27 ? 2 : 3;                               // interpolation_p2.c (27-point) / interpolation_v2.c (fv2, fv4)
  extern __shared__ double vt_lds[];
  __shared__ int s_bc[kBcWords];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6, nth = (int)blockDim.x;
  const int n = Tp->n, e_id = Tp->e_id, R_id = Tp->R_id;
  const lds_dptr img = (lds_dptr)vt_lds;
  // slots of an image: x, VECTOR_TEMP, rhs, Dinv, alpha, beta_i, beta_j, beta_k (small_level_run addresses them by slot number)
  const int slot_vec[8] = { e_id, VECTOR_TEMP, R_id, VECTOR_DINV, kHelm ? VECTOR_ALPHA : -1, kVC ? VECTOR_BETA_I : -1, kVC ? VECTOR_BETA_J : -1, kVC ? VECTOR_BETA_K : -1 };
  int unused_n = 0;
  const int legs = Tp->legs;                                       // bit 0: the way down, bit 1: the bottom solve, bit 2: the way up
  for (int ph = 0; ph < 2 * n - 1; ph++) {
    if (ph < n - 1 ? !(legs & 1) : (ph == n - 1 ? !(legs & 2) : !(legs & 4))) continue;
    if (ph == n - 1) {                                             // ---- the bottom solve (solvers.c IterativeSolver -> BiCGStab)
      const hpgmg_hip_small_tail_level &lb = Tp->lv[n - 1];
      const hpgmg_hip_level Lb = lb.L;
      BottomArgs B;
      B.e_id = e_id; B.R_id = R_id; B.krylov_base = Tp->krylov_base; B.bc_kind = lb.n_bc > 0 ? lb.bc_kind : 0; B.zero_first = lb.zero_first; B.n_bc = lb.n_bc;
      B.a = Tp->a; B.b = Tp->b; B.h2inv = lb.h2inv; B.want = Tp->want; B.bc_list = lb.bc_list; B.krylov_iterations = Tp->krylov_iterations;
      bottom_bicgstab_body<V>(Lb, B);
      __threadfence(); __syncthreads();
      VT_MARK();
      continue;
    }
    const bool down = ph < n - 1;
    const int l = down ? ph : (2 * (n - 1) - ph);
    const hpgmg_hip_small_tail_level &lv = Tp->lv[l];
    const hpgmg_hip_level L = lv.L;
    const int vol = L.volume, jS = L.jStride, kS = L.kStride, dim = L.dim, first = L.ghosts * (1 + jS + kS);
    // ---- the image of level l
    {
      const double *base = L.box_base[0];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (slot_vec[q] < 0) continue;
        const gbl_cdptr g = (gbl_cdptr)(base + (size_t)slot_vec[q] * vol);
#pragma unroll 4
        for (int t = tid; t < vol; t += nth) img[q * vol + t] = g[t];
      }
    }
    VT_MARK();
    if (!down) {
      // ---- interpolation_vcycle(level l, e, 1.0, level l + 1, e): the coarse correction behind the image, its boundary conditions, the rule
      const hpgmg_hip_small_tail_level &lc = Tp->lv[l + 1];
      const hpgmg_hip_level Lc = lc.L;
      const int cvol = Lc.volume, cj = Lc.jStride, ck = Lc.kStride, cfirst = Lc.ghosts * (1 + cj + ck);
      const lds_dptr stage = img + 8 * vol;
      const gbl_cdptr gx = (gbl_cdptr)(Lc.box_base[0] + (size_t)e_id * cvol);
      for (int t = tid; t < cvol; t += nth) stage[t] = gx[t];
      lds_bc_words_fill(s_bc, lc.ibc_list, lc.n_ibc, tid, nth, false);
      __syncthreads();
      const lds_dptr cx = stage + cfirst;
      if (lc.ibc_kind && lc.ibc_zero_first) { for (int e = wave; e < lc.n_ibc; e += nwaves) bc_zero_entry_at(cx, Lc, lds_bc_entry(s_bc, e), lane, 64); __syncthreads(); }
      for (int e = wave; e < lc.n_ibc; e += nwaves) {
        const blockCopy_type en = lds_bc_entry(s_bc, e);
        if (lc.ibc_kind == 1) bc_p1_entry_at(cx, Lc, en, lane, 64);
        else if (lc.ibc_kind == 2) bc_p2_entry_at(cx, Lc, en, lane, 64);
        else if (lc.ibc_kind == 3) bc_v2_entry_at(cx, Lc, en, lane, 64);
      }
      __syncthreads();
      const lds_dptr xf = img + first;
      for (int t = tid; t < dim * dim * dim; t += nth) {
        const int i = t % dim, j = (t / dim) % dim, k = t / (dim * dim);
        const lds_dptr c = cx + ((i >> 1) + (j >> 1) * cj + (k >> 1) * ck);
        double tk[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
          double tj[3];
#pragma unroll
          for (int jj = 0; jj < 3; jj++) {
            double line[3];
#pragma unroll
            for (int ii = 0; ii < 3; ii++) line[ii] = c[(ii - 1) + (jj - 1) * cj + (kk - 1) * ck];
            tj[jj] = interp_rule<ORDER>((i & 1) != 0, line);
          }
          tk[kk] = interp_rule<ORDER>((j & 1) != 0, tj);
        }
        const double add = interp_rule<ORDER>((k & 1) != 0, tk);
        const int ijk = i + j * jS + k * kS;
        xf[ijk] = 1.0 * xf[ijk] + add;
      }
    }
    lds_bc_words_fill(s_bc, lv.bc_list, lv.n_bc, tid, nth, lv.bc_kind == 4);
    __syncthreads();
    VT_MARK();
    // ---- smooth(level l), and on the way down residual(level l, VECTOR_TEMP, e, R)
    for (int pass = 0; pass < (down ? 2 : 1); pass++) {
      SmallArgs A;
      A.mode = pass ? MODE_RESIDUAL : Tp->mode; A.sweeps = pass ? 1 : Tp->sweeps; A.out_of_place = pass ? 0 : Tp->out_of_place;
      A.x_id = 0; A.rhs_id = 2; A.res_id = 1; A.bc_kind = lv.n_bc > 0 ? lv.bc_kind : 0; A.zero_first = lv.zero_first;
      A.a = Tp->a; A.b = Tp->b; A.h2inv = lv.h2inv; A.copy_list = nullptr; A.n_copy = 0; A.bc_list = lv.bc_list; A.n_bc = lv.n_bc; A.lds_resident = 1; A.timeline = nullptr;
      const int ids[kSmallSlots] = { 0, 1, 2, 3, 4, 5, 6, 7, pass ? 1 : 0 };
      small_level_run<V, true>(L, A, vt_lds, lv.bc_list, lv.n_bc <= 32 ? (const int *)s_bc : nullptr, ids, nullptr, unused_n,
                               [&](int s, double &c1, double &c2) { c1 = lv.c1[s]; c2 = lv.c2[s]; });
      VT_MARK();
    }
    if (down) {
      // ---- restriction(level l + 1, R, level l, VECTOR_TEMP, RESTRICT_CELL) and zero_vector(level l + 1, e)
      const hpgmg_hip_level Lc = Tp->lv[l + 1].L;
      const int cdim = Lc.dim, cj = Lc.jStride, ck = Lc.kStride, cvol = Lc.volume;
      const lds_dptr tf = img + (vol + first);
      const gbl_dptr rc = (gbl_dptr)(Lc.box_base[0] + (size_t)R_id * cvol + (size_t)Lc.ghosts * (1 + cj + ck));
      for (int t = tid; t < cdim * cdim * cdim; t += nth) {
        const int i = t % cdim, j = (t / cdim) % cdim, k = t / (cdim * cdim);
        const lds_dptr f = tf + (2 * i + 2 * j * jS + 2 * k * kS);
        double v = f[0] + f[1]; v = v + f[jS]; v = v + f[1 + jS]; v = v + f[kS]; v = v + f[1 + kS]; v = v + f[jS + kS]; v = v + f[1 + jS + kS];
        rc[i + j * cj + k * ck] = v * 0.125;
      }
      const gbl_dptr zc = (gbl_dptr)(Lc.box_base[0] + (size_t)e_id * cvol);
      for (int t = tid; t < cvol; t += nth) zc[t] = 0.0;
    }
    // ---- what was written goes back: x and VECTOR_TEMP, ghost zones included (the boundary entries filled them)
    {
      double *base = L.box_base[0];
      const gbl_dptr gx = (gbl_dptr)(base + (size_t)e_id * vol), gt = (gbl_dptr)(base + (size_t)VECTOR_TEMP * vol);
#pragma unroll 4
      for (int t = tid; t < vol; t += nth) { gx[t] = img[t]; gt[t] = img[vol + t]; }
    }
    __threadfence(); __syncthreads();
    VT_MARK();
  }
#ifdef HPGMG_EXP_TIMELINE
  if (tl && threadIdx.x == 0) { tl[254] = __builtin_amdgcn_s_memtime(); tl[255] = (unsigned long long)tl_n; }
#endif
#undef VT_MARK
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {
int hpgmg_hip_graph_flush(void);

int hpgmg_hip_small_level_max_cells(void) { return 4096; }
// mode: 0 Chebyshev, 1 GSRB, 2 Jacobi (x_id <-> VECTOR_TEMP ping-pong as smooth() does; GSRB in place unless out_of_place), 3 residual
// (res_id = rhs - A x), 4 apply_op (res_id = A x); c1 / c2: per-sweep Chebyshev coefficients (Jacobi: c2 = the weight)
int hpgmg_hip_small_level_op(const hpgmg_hip_level *L, int variant, int mode, int sweeps, int x_id, int rhs_id, int res_id, int out_of_place,
                             double a, double b, double h2inv, const double *c1, const double *c2,
                             const blockCopy_type *copy_list, int n_copy, const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first) {
  HPGMG_SKIP_IF_REPLAY();
  if (L->num_boxes <= 0) return 0;
  if (sweeps < 1 || sweeps > 8 || mode < MODE_CHEBY || mode > MODE_APPLY || (long long)L->num_boxes * L->dim * L->dim * L->dim > 4096)
    return record_error(hipErrorInvalidValue, "small_level_op: arguments");
  SmallArgs A = {};
  A.mode = mode; A.sweeps = sweeps; A.x_id = x_id; A.rhs_id = rhs_id; A.res_id = res_id; A.out_of_place = out_of_place;
  A.bc_kind = n_bc > 0 ? bc_kind : 0; A.zero_first = zero_first; A.a = a; A.b = b; A.h2inv = h2inv;
  for (int q = 0; q < sweeps; q++) { A.c1[q] = c1 ? c1[q] : 0.0; A.c2[q] = c2 ? c2[q] : 0.0; }
  A.copy_list = copy_list; A.n_copy = copy_list ? n_copy : 0; A.bc_list = bc_list; A.n_bc = bc_list ? n_bc : 0;
  const long long cells = (long long)L->num_boxes * L->dim * L->dim * L->dim;
  const int threads = cells >= 1024 ? 1024 : (cells >= 256 ? 256 : 64);
  // one box whose vectors fit the LDS: work on an image of it there
  const size_t image = (size_t)kSmallSlots * (size_t)L->volume * sizeof(double);
  static const int no_lds = env_int("HPGMG_TUNE_SMALL_NO_LDS", 0);
  A.lds_resident = (!no_lds && L->num_boxes == 1 && A.n_copy == 0 && image <= 150 * 1024) ? 1 : 0;
#ifdef HPGMG_EXP_TIMELINE
  A.timeline = (unsigned long long *)g_exp_timeline;
#endif
  const size_t lds = A.lds_resident ? image : 0;
  const int threads_used = A.lds_resident ? 1024 : threads;        // the image is copied by every lane there is
  if (!A.lds_resident) return record_error(hipErrorInvalidValue, "small_level_op: a level of one box whose vectors fit the LDS");
#define SMALL_CASE(VAR) { \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)small_level_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); once = true; } \
    hipLaunchKernelGGL((small_level_kernel<VAR>), dim3(1), dim3(threads_used), lds, g_stream, *L, A); }
  switch (variant) {
    case HPGMG_HIP_27PT_CC:          SMALL_CASE(HPGMG_HIP_27PT_CC) break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: SMALL_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ) break;
    case HPGMG_HIP_FV4_VC_POISSON:   SMALL_CASE(HPGMG_HIP_FV4_VC_POISSON) break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: SMALL_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ) break;
    case HPGMG_HIP_7PT_VC_POISSON:   SMALL_CASE(HPGMG_HIP_7PT_VC_POISSON) break;
    case HPGMG_HIP_7PT_CC:           SMALL_CASE(HPGMG_HIP_7PT_CC) break;
    default: return record_error(hipErrorInvalidValue, "small_level_op: variant");
  }
#undef SMALL_CASE
  HPGMG_LAUNCH_CHECK("small_level_kernel");
  return 0;
}

// BiCGStab bottom solve of the 27-point / fv2 / fv4 plugins on a level of one box (bottom_bicgstab_kernel): x_id holds the initial guess and
// receives the solution; the eight work vectors start at krylov_base; bc_list / bc_kind / zero_first as for hpgmg_hip_small_level_op;
// krylov_iterations: device-visible host counter the kernel adds its iteration count to, or NULL.  Dirichlet only.
int hpgmg_hip_bottom_bicgstab_max_cells(void) { return 512; }
int hpgmg_hip_bottom_bicgstab(const hpgmg_hip_level *L, int variant, int x_id, int rhs_id, int krylov_base, double a, double b, double h2inv, double want,
                              const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first, int *krylov_iterations) {
  HPGMG_SKIP_IF_REPLAY();
  const long long cells = (long long)L->dim * L->dim * L->dim;
  if (L->num_boxes != 1 || cells > 512 || L->periodic) return record_error(hipErrorInvalidValue, "bottom_bicgstab: a level of one box of at most 512 cells, Dirichlet");
  BottomArgs A = {};
  A.e_id = x_id; A.R_id = rhs_id; A.krylov_base = krylov_base; A.a = a; A.b = b; A.h2inv = h2inv; A.want = want;
  A.bc_list = bc_list; A.n_bc = bc_list ? n_bc : 0; A.bc_kind = A.n_bc > 0 ? bc_kind : 0; A.zero_first = zero_first; A.krylov_iterations = krylov_iterations;
  // eight waves whatever the level: the boundary entries (26 of them) are a wave's work each
  static const int tune_threads = env_int("HPGMG_TUNE_BOTTOM_THREADS", 512);
  const int threads = (cells > 256 || tune_threads >= 512) ? 512 : (cells > 64 || tune_threads >= 256 ? 256 : 64);
  const size_t lds = ((size_t)5 * L->volume + 1100) * sizeof(double);
  if (lds > 150 * 1024) return record_error(hipErrorInvalidValue, "bottom_bicgstab: box too large for the LDS image");
#define BOTTOM_CASE(VAR) { \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)bottom_bicgstab_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); once = true; } \
    hipLaunchKernelGGL((bottom_bicgstab_kernel<VAR>), dim3(1), dim3(threads), lds, g_stream, *L, A); }
  switch (variant) {
    case HPGMG_HIP_27PT_CC:          BOTTOM_CASE(HPGMG_HIP_27PT_CC) break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: BOTTOM_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ) break;
    case HPGMG_HIP_FV4_VC_POISSON:   BOTTOM_CASE(HPGMG_HIP_FV4_VC_POISSON) break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: BOTTOM_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ) break;
    case HPGMG_HIP_7PT_VC_POISSON:   BOTTOM_CASE(HPGMG_HIP_7PT_VC_POISSON) break;
    case HPGMG_HIP_7PT_CC:           BOTTOM_CASE(HPGMG_HIP_7PT_CC) break;
    default: return record_error(hipErrorInvalidValue, "bottom_bicgstab: variant");
  }
#undef BOTTOM_CASE
  HPGMG_LAUNCH_CHECK("bottom_bicgstab_kernel");
  return 0;
}
// A queue of BLAS-1 / operator calls on a level of one box of <= 512 cells as one launch (small_ops_kernel).  kinds: 1 add (c = sa*a + sb*b),
// 2 mul (c = sa*a*b), 3 scale (c = sa*a), 4 apply_op (c = A a), 5 residual (c = b - A a), 6 dot (a, b), 7 norm (a); a value-returning
// operation may only come last, its value goes to *value_out (the call then waits for it).
static long long g_small_ops_launches = 0;
long long hpgmg_hip_small_ops_launch_count(void) { return g_small_ops_launches; }
int hpgmg_hip_small_ops_max(void) { return kSmallOpsMax; }
int hpgmg_hip_small_ops(const hpgmg_hip_level *L, int variant, int n, const int *kinds, const int *c, const int *a, const int *b, const double *sa, const double *sb,
                        const blockCopy_type *bc_list, int n_bc, int bc_kind, int zero_first, double op_a, double op_b, double h2inv, double *value_out, double *value2_out) {
  if (int e = hpgmg_hip_graph_flush()) return e;
  if (n < 1 || n > kSmallOpsMax || L->num_boxes != 1 || L->dim > 8 || L->periodic) return record_error(hipErrorInvalidValue, "small_ops: one Dirichlet box of side <= 8, 1..12 operations");
  SmallOpsArgs A = {};
  A.n = n; A.bc_list = bc_list; A.n_bc = bc_list ? n_bc : 0; A.bc_kind = A.n_bc > 0 ? bc_kind : 0; A.zero_first = zero_first; A.a = op_a; A.b = op_b; A.h2inv = h2inv;
  int nvalues = 0;
  for (int q = 0; q < n; q++) {
    const bool is_value = (kinds[q] == SO_DOT || kinds[q] == SO_NORM);
    if (kinds[q] < SO_ADD || kinds[q] > SO_NORM || (is_value && q < n - 2) || (is_value && q == n - 2 && !(kinds[n - 1] == SO_DOT || kinds[n - 1] == SO_NORM)))
      return record_error(hipErrorInvalidValue, "small_ops: operation list (value-returning operations only as the last one or two entries)");
    A.op[q].kind = kinds[q]; A.op[q].c = c[q]; A.op[q].a = a[q]; A.op[q].b = b[q]; A.op[q].sa = sa[q]; A.op[q].sb = sb[q];
    nvalues += is_value ? 1 : 0;
  }
  const bool wants = nvalues > 0;
  if ((nvalues >= 1) != (value_out != nullptr) || (nvalues == 2) != (value2_out != nullptr)) return record_error(hipErrorInvalidValue, "small_ops: one output pointer per value-returning operation");
  if (wants) { A.result = reduction_slot_next(&A.seq); if (!A.result) return record_error(hipErrorOutOfMemory, "small_ops: result slot"); }
  const int cells = L->dim * L->dim * L->dim;
  const int threads = cells > 64 ? 512 : (A.n_bc > 4 ? 512 : 64);           // one cell per lane (no striding); the boundary entries are a wave's work each
#define SMALL_OPS_CASE(VAR) hipLaunchKernelGGL((small_ops_kernel<VAR>), dim3(1), dim3(threads), 0, g_stream, *L, A);
  switch (variant) {
    case HPGMG_HIP_27PT_CC:          SMALL_OPS_CASE(HPGMG_HIP_27PT_CC) break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: SMALL_OPS_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ) break;
    case HPGMG_HIP_FV4_VC_POISSON:   SMALL_OPS_CASE(HPGMG_HIP_FV4_VC_POISSON) break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: SMALL_OPS_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ) break;
    case HPGMG_HIP_7PT_VC_POISSON:   SMALL_OPS_CASE(HPGMG_HIP_7PT_VC_POISSON) break;
    case HPGMG_HIP_7PT_CC:           SMALL_OPS_CASE(HPGMG_HIP_7PT_CC) break;
    default: return record_error(hipErrorInvalidValue, "small_ops: variant");
  }
#undef SMALL_OPS_CASE
  g_small_ops_launches++;
  HPGMG_LAUNCH_CHECK("small_ops_kernel");
  if (wants) { if (int e = reduction_fetch(value_out)) return e; if (value2_out) *value2_out = reduction_second_value(); }
  return 0;
}
// V-cycle tail below a level of one box (small_vtail_kernel).  The argument block lives in device memory: it is the same for every visit of
// a chain in a solve, so a few of them are kept and uploaded only when their contents change.
long long hpgmg_hip_small_vtail_lds_limit(void) { return 150 * 1024 / (long long)sizeof(double); }
long long hpgmg_hip_small_vtail_lds_doubles(const hpgmg_hip_small_tail_args *T) {
  long long need = 0;
  for (int l = 0; l + 1 < T->n; l++) { const long long v = 8LL * T->lv[l].L.volume + T->lv[l + 1].L.volume; if (v > need) need = v; }
  const long long bottom = 5LL * T->lv[T->n - 1].L.volume + 1100;
  return bottom > need ? bottom : need;
}
static long long g_small_vtail_launches = 0;
long long hpgmg_hip_small_vtail_launch_count(void) { return g_small_vtail_launches; }
int hpgmg_hip_small_vtail(const hpgmg_hip_small_tail_args *T, int variant) {
  HPGMG_SKIP_IF_REPLAY();
  if (!T || T->n < 2 || T->n > HPGMG_HIP_SMALL_TAIL_MAX_LEVELS || T->legs < 1 || T->legs > 7 || T->sweeps < 1 || T->sweeps > 8 || T->mode < MODE_CHEBY || T->mode > MODE_JACOBI)
    return record_error(hipErrorInvalidValue, "small_vtail: arguments");
  for (int l = 0; l < T->n; l++) {
    const hpgmg_hip_level &L = T->lv[l].L;
    if (L.num_boxes != 1 || L.periodic || (l > 0 && 2 * L.dim != T->lv[l - 1].L.dim)) return record_error(hipErrorInvalidValue, "small_vtail: a chain of levels of one box, halving, Dirichlet");
  }
  if ((T->legs & 2) && (long long)T->lv[T->n - 1].L.dim * T->lv[T->n - 1].L.dim * T->lv[T->n - 1].L.dim > 512) return record_error(hipErrorInvalidValue, "small_vtail: bottom level too large");
  const long long need = hpgmg_hip_small_vtail_lds_doubles(T);
  if (need > hpgmg_hip_small_vtail_lds_limit()) return record_error(hipErrorInvalidValue, "small_vtail: the chain does not fit the LDS");
  constexpr int kSlots = 8;
  static hpgmg_hip_small_tail_args host_copy[kSlots];
  static hpgmg_hip_small_tail_args *dev_copy[kSlots];
  static int used = 0, next = 0;
  int slot = -1;
  for (int q = 0; q < used; q++) if (memcmp(&host_copy[q], T, sizeof *T) == 0) { slot = q; break; }
  if (slot < 0) {
    slot = (used < kSlots) ? used++ : (next++ % kSlots);
    if (!dev_copy[slot]) HPGMG_CHECK(hipMalloc((void **)&dev_copy[slot], sizeof *T));
    host_copy[slot] = *T;
    HPGMG_CHECK(hipMemcpyAsync(dev_copy[slot], &host_copy[slot], sizeof *T, hipMemcpyHostToDevice, g_stream));   // stream order: after every launch that reads the slot
  }
  const size_t lds = (size_t)need * sizeof(double);
  unsigned long long *tl = nullptr;
#ifdef HPGMG_EXP_TIMELINE
  tl = (unsigned long long *)g_exp_timeline;
#endif
#define VTAIL_CASE(VAR) { \
    static bool once = false; if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)small_vtail_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); once = true; } \
    hipLaunchKernelGGL((small_vtail_kernel<VAR>), dim3(1), dim3(512), lds, g_stream, (const hpgmg_hip_small_tail_args *)dev_copy[slot], tl); }
  switch (variant) {
    case HPGMG_HIP_27PT_CC:          VTAIL_CASE(HPGMG_HIP_27PT_CC) break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ: VTAIL_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ) break;
    case HPGMG_HIP_FV4_VC_POISSON:   VTAIL_CASE(HPGMG_HIP_FV4_VC_POISSON) break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ: VTAIL_CASE(HPGMG_HIP_7PT_VC_HELMHOLTZ) break;
    case HPGMG_HIP_7PT_VC_POISSON:   VTAIL_CASE(HPGMG_HIP_7PT_VC_POISSON) break;
    default: return record_error(hipErrorInvalidValue, "small_vtail: variant");
  }
#undef VTAIL_CASE
  g_small_vtail_launches++;
  HPGMG_LAUNCH_CHECK("small_vtail_kernel");
  return 0;
}

}  // extern "C"

// fv4_rb.hpp -- the two coloured half sweeps of one out-of-place GSRB sweep of the 4th-order operator (reference gsrb.c:24-132 with
// operators.fv4.c:55-134, GSRB_OOP, and apply_BCs_v4 boundary_fv.c:262-569 between the half sweeps) in ONE pass over the level.
//
// The reference runs   exchange + apply_BCs_v4(x);  t = red half sweep of x;  exchange + apply_BCs_v4(t);  x' = black half sweep of t
// -- two passes of 56 B per cell each (x, rhs, Dinv, beta_i/j/k read; the other vector written in full).  Here a workgroup of 64 x 8
// lanes owns a 64 (i) x 16 (j) tile of a box -- a lane owns two vertically adjacent cells, hence one of each colour on every plane --
// and marches in +k with two stages per step:
//   R(q):   the red half sweep on plane q of the tile and of the cells FACE-adjacent to it (a black cell's only red neighbours are its six
//           face neighbours, so nothing further out has to be recomputed: 80 ring cells per plane, done by 80 lanes that each own a red /
//           black pair of them).  Inputs: planes q-1, q, q+1 of x in LDS with a three-cell halo, planes q-1 .. q+1 of beta_i / beta_j and
//           faces q, q+1 of beta_k with a two-cell halo, x[q-2], x[q+2] of the own column in registers.  The result goes to a compact
//           LDS ring that holds ONLY the red cells of t (a black cell of t is the cell of x, which is still in the x ring);
//   B(q-1): the black half sweep on plane q-1 of the tile proper: red neighbours from the t ring, everything else from the x ring; x'
//           is stored (both cells of the pair: the red one is the value R formed a step earlier).
// LDS: x ring 4 planes x 70 x 22, beta rings (3 + 3 + 2) planes x 68 x 20, t ring 3 planes x 18 x 34 doubles = 151 008 B of the CU's
// 160 KiB -- one workgroup per CU.  What makes it fit: the 30 coefficient values B(q) needs (6 face values, 12 differences) are formed one
// step EARLY, while the planes they come from are in the rings for R(q) anyway, and wait in registers; otherwise every coefficient ring
// would need a fourth plane.
// Ghost cells of t outside the domain (the reference's second apply_BCs_v4):
//   * in i and j they are formed in LDS, plane by plane, from t itself (v4_near / v4_far over the four cells next to the wall; the k-edge
//     cell diagonal to a tile corner by the reference's two passes, i then j) and written where B reads them: red-parity positions into
//     the t ring, black-parity positions OVER the x ring's ghost values (R is done with those by then);
//   * in k they are read from the ghost planes of vector P.tg, which a small pre-pass has filled: the existing tiled kernel forms t on
//     the four planes next to the bottom / top of the domain and the existing boundary kernel extrapolates them (faces and the i-k /
//     j-k edges).  Registers and LDS positions of black parity on those planes take the value of P.tg, red-parity ones that of x.
// Cells outside the box but inside the domain are read from the box that owns them (common.hpp gf_column).  Coefficients come from the own
// box's ghost zone, as in the reference; outside the DOMAIN that is this box's own extrapolation, which differs from what a neighbouring box
// holds for the same place (extrapolate_betas works with box-relative normals).  So a recomputed cell that belongs to ANOTHER box and whose
// stencil reaches outside the domain ("special": the lines where an internal box face meets a domain wall) is not recomputed here: its t is
// read from the interior of P.tg, where fv4_special_kernel (below) has formed it with the owning box's coefficients.
// Every update is the expression tree of fv4_tile.hpp (= the reference macro), so x' is bit-identical to the two separate half sweeps;
// the intermediate vector is never materialised: 56 B per cell per SWEEP.  x' must not alias x.
#pragma once
#include "common.hpp"
#include "fv4_tile.hpp"
#include "block_ops.hpp"

namespace hpgmg {

struct VecSel { double *const *base; int id; };      // vector `id` behind a table of box bases: a level vector or a plugin-private scratch vector
__device__ __forceinline__ gptr sel_origin(const hpgmg_hip_level &L, const VecSel &S, int box) {
  return as_global(S.base[box]) + (size_t)S.id * (size_t)L.volume + (size_t)L.ghosts * (size_t)(1 + L.jStride + L.kStride);
}

struct Fv4RbArgs {
  VecSel x, out, tg;
  int rhs_id;
  double a, b, h2inv;
  int sweep;                            // number of the first (even) half sweep: its colour is (i ^ j ^ k ^ sweep) & 1 == 0
  int tiles_i, tiles_j, chunks_k, kchunk, per_xcd, total_blocks;
  const int *order;                     // dispatch slot -> tile (nullptr: identity): the tiles at a domain wall take longer, so each XCD starts with them
  int timeline_wg;                      // ... which workgroup (logical index; < 0: one in the middle of the grid)
  unsigned long long *timeline;         // experiment builds (-DHPGMG_EXP_TIMELINE): where two waves of one workgroup record the clock at the stage boundaries
};

namespace fv4rb {
constexpr int TJ = 16;
// Tile geometry: TI x 16 cells, TI x 8 lanes.  TI = 64: one workgroup of 8 waves fills a CU's LDS; TI = 32: 80 608 B, so TWO workgroups of 4
// waves share a CU and one's barriers / LDS phases run under the other's arithmetic and loads (at 1.57 x instead of 1.36 x the minimum
// traffic); also the form for boxes of 32^3.
template <int TI_> struct Geom {
  static constexpr int TI = TI_, NT = TI * 8;
  static constexpr int WX = TI + 6, HX = TJ + 6, PX = WX * HX;          // x planes: three-cell halo
  static constexpr int WB = TI + 4, HB = TJ + 4, PB = WB * HB;          // coefficient planes: two-cell halo
  static constexpr int ST = TI / 2 + 2, HT = TJ + 2, PT = ST * HT;      // red cells of t on the tile + 1: column c = i + 1 -> c >> 1
  static constexpr int NRING = TI + TJ;                                 // lanes that own a pair of ring cells (the last NRING of the workgroup)
  static constexpr int NH = 6 * TI + 12 + 6 * TJ;                       // halo cells of an x plane (within three steps of the tile, corners cut)
  static constexpr int NBH = 4 * TI + 12 + 4 * TJ;                      // halo cells of a coefficient plane
  static constexpr bool H2 = NH > NT;                                   // lanes 0 .. NH-NT-1 own a second x halo cell
  static constexpr bool PARK = (TI == 64);                              // per-lane "box above" pointers wait in LDS (there is room) instead of being rebuilt
  static constexpr int LDS_DOUBLES = 4 * PX + 8 * PB + 3 * PT + NT / 2 + (PARK ? NT + 4 * NRING : 0);       // + the boundary descriptors (one int per lane)
  static constexpr size_t LDS_BYTES = (size_t)LDS_DOUBLES * sizeof(double);
  static_assert(NBH <= NT - NRING && NH <= 2 * NT && 68 + 2 * (2 * TI + 2) <= NT, "lane roles");
  __device__ __forceinline__ static constexpr int posX(int ci, int cj) { return (cj + 3) * WX + (ci + 3); }
  __device__ __forceinline__ static constexpr int posB(int ci, int cj) { return (cj + 2) * WB + (ci + 2); }
  __device__ __forceinline__ static int posT(int ci, int cj) { return (cj + 1) * ST + ((ci + 1) >> 1); }
  // the n-th halo cell of an x plane / of a coefficient plane, relative to the tile
  __device__ __forceinline__ static void halo_x(int n, int &hi, int &hj) {
    if (n < TI)              { hj = -3; hi = n; }
    else if (n < 2 * TI)     { hj = TJ + 2; hi = n - TI; }
    else if (n < 3 * TI + 2) { hj = -2; hi = -1 + (n - 2 * TI); }
    else if (n < 4 * TI + 4) { hj = TJ + 1; hi = -1 + (n - (3 * TI + 2)); }
    else if (n < 5 * TI + 8) { hj = -1; hi = -2 + (n - (4 * TI + 4)); }
    else if (n < 6 * TI + 12) { hj = TJ; hi = -2 + (n - (5 * TI + 8)); }
    else                     { const int h = n - (6 * TI + 12), c = h % 6; hj = h / 6; hi = (c < 3) ? c - 3 : TI + (c - 3); }
  }
  __device__ __forceinline__ static void halo_b(int n, int &hi, int &hj) {
    if (n < TI + 2)           { hj = -2; hi = -1 + n; }
    else if (n < 2 * TI + 4)  { hj = TJ + 1; hi = -1 + (n - (TI + 2)); }
    else if (n < 3 * TI + 8)  { hj = -1; hi = -2 + (n - (2 * TI + 4)); }
    else if (n < 4 * TI + 12) { hj = TJ; hi = -2 + (n - (3 * TI + 8)); }
    else                      { const int h = n - (4 * TI + 12), c = h % 4; hj = h / 4; hi = (c < 2) ? c - 2 : TI + (c - 2); }
  }
};
#ifdef HPGMG_EXP_STATIC_SLOTS      /* timing experiment only (results are garbage): what the march would cost if every ring slot were a compile-time constant */
__device__ __forceinline__ int slot4(int p) { (void)p; return 1; }
__device__ __forceinline__ int slot3(int p) { (void)p; return 1; }
__device__ __forceinline__ int slot2(int p) { (void)p; return 1; }
#else
__device__ __forceinline__ int slot4(int p) { return p & 3; }
__device__ __forceinline__ int slot3(int p) { return ((p % 3) + 3) % 3; }
__device__ __forceinline__ int slot2(int p) { return p & 1; }
#endif

#define FV4RB_FENCE() __builtin_amdgcn_sched_barrier(0)      /* nothing is scheduled across: reads stay together, ahead of the arithmetic */
}  // namespace fv4rb

template <int V, int TI_>
__global__ __launch_bounds__(TI_ * 8, 2) void fv4_rb_kernel(const hpgmg_hip_level L, const Fv4RbArgs P) {      // two waves per SIMD: 256 registers
  using namespace fv4rb;
  using G = Geom<TI_>;
  constexpr int TI = G::TI, NT = G::NT, WX = G::WX, PX = G::PX, WB = G::WB, PB = G::PB, ST = G::ST, PT = G::PT, NRING = G::NRING;
  constexpr bool H2 = G::H2, PARK = G::PARK;
  constexpr bool kHelm = (V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  auto posX = [](int ci, int cj) { return G::posX(ci, cj); };
  auto posB = [](int ci, int cj) { return G::posB(ci, cj); };
  auto posT = [](int ci, int cj) { return G::posT(ci, cj); };
  extern __shared__ double fv4rb_lds[];
  double *sX = fv4rb_lds, *sBI = sX + 4 * PX, *sBJ = sBI + 3 * PB, *sBK = sBJ + 3 * PB, *sT = sBK + 2 * PB;
  int *sBC = (int *)(sT + 3 * PT);                                             // one boundary-cell descriptor per lane (tiles at a wall)
  unsigned long long *sPtr = (unsigned long long *)(sT + 3 * PT + NT / 2);     // PARK only

  int logical = xcd_logical_block((int)blockIdx.x, P.per_xcd);
  if (P.order) logical = P.order[logical];
  if (logical >= P.total_blocks) return;
  int t = logical;
  const int ti = t % P.tiles_i; t /= P.tiles_i;
  const int tj = t % P.tiles_j; t /= P.tiles_j;
  const int ck = t % P.chunks_k; t /= P.chunks_k;
  const int box = t;
  const int li = (int)threadIdx.x, lj = (int)threadIdx.y, tid = lj * TI + li;     // lj: the pair of rows 2 lj, 2 lj + 1
#ifdef HPGMG_EXP_TIMELINE
  if (P.timeline && tid == 0 && logical < 8192) P.timeline[16384 + 3 * logical] = __builtin_amdgcn_s_memrealtime();     // workgroup start
#endif
  const int i0 = ti * TI, j0 = tj * TJ;
  const int dim = L.dim, jS = L.jStride, kS = L.kStride;
  const int k0 = ck * P.kchunk, k1 = (k0 + P.kchunk < dim) ? k0 + P.kchunk : dim;
  const size_t vol = (size_t)L.volume, first = (size_t)L.ghosts * (size_t)(1 + jS + kS);

  const int *nb = L.box_nbr + 6 * box;
  const bool wall_ilo = nb[0] == -1, wall_ihi = nb[1] == -1, wall_jlo = nb[2] == -1, wall_jhi = nb[3] == -1, wall_klo = nb[4] == -1, wall_khi = nb[5] == -1;
  const bool t_ilo = wall_ilo && i0 == 0, t_ihi = wall_ihi && i0 + TI == dim, t_jlo = wall_jlo && j0 == 0, t_jhi = wall_jhi && j0 + TJ == dim;
  const bool bottom = wall_klo && k0 == 0, top = wall_khi && k1 == dim;      // this chunk starts / ends at the domain boundary
  // how far box-relative coordinate c lies outside the DOMAIN (0: inside)
  auto out_i = [&](int c) { return (c < 0 && wall_ilo) ? -c : ((c >= dim && wall_ihi) ? c - dim + 1 : 0); };
  auto out_j = [&](int c) { return (c < 0 && wall_jlo) ? -c : ((c >= dim && wall_jhi) ? c - dim + 1 : 0); };
  const int par0 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;
  auto is_red = [&](int ci, int cj, int ck2) { return (((ci ^ cj ^ ck2 ^ par0) & 1) == 0); };    // box-relative cell coordinates

  // ---- (1) the own pair of cells (gi, gj), (gi, gj + 1).  Every vector of the own box is reached from one uniform base with the
  // lane's 32-bit offset own_g; planes p >= dim through the base of the box above (or the own ghost zone)
  // Addressing: base pointers that are the same for every lane (they stay in scalar registers) + ONE unsigned 32-bit offset per lane,
  // so a load is "scalar base + lane offset" with no 64-bit address arithmetic per lane: lvb / xb / tgb / outb point at the START of the
  // box's storage and own_o = (first interior cell) + (the lane's lower cell) is never negative, ghost planes included.
  const int gi = i0 + li, gj = j0 + 2 * lj, own_g = gi + gj * jS;
  const unsigned own_o = (unsigned)(first + (size_t)own_g), ujS = (unsigned)jS, ukS = (unsigned)kS;
  const unsigned own_b = own_o * 8u, bjS = ujS * 8u, bkS = ukS * 8u;     // the same in bytes (a box's vector is far smaller than 4 GB)
  const int ownX = posX(li, 2 * lj), ownB = posB(li, 2 * lj), ownT = posT(li, 2 * lj);
  gcptr lvb = as_global(L.box_base[box]), lvb_hi = lvb;                     // level vectors: lvb[id * vol + own_o + p * kS (+ jS)]
  gcptr xb = as_global(P.x.base[box]) + (size_t)P.x.id * vol, xb_hi = xb;
  gcptr tgb = as_global(P.tg.base[box]) + (size_t)P.tg.id * vol;            // its k ghost planes: boundary values of t
  gptr outb = as_global(P.out.base[box]) + (size_t)P.out.id * vol;
  gcptr tgb_hi = (nb[5] >= 0) ? as_global(P.tg.base[nb[5]]) + (size_t)P.tg.id * vol - (long long)dim * kS : tgb;
  if (nb[5] >= 0) { lvb_hi = as_global(L.box_base[nb[5]]) - (long long)dim * kS; xb_hi = as_global(P.x.base[nb[5]]) + (size_t)P.x.id * vol - (long long)dim * kS; }

  // ---- (2) one halo cell of the x planes (the cells within three steps of the tile: NH = 492 / 300); with TI = 32 the first 44 lanes own two
  const bool has_h = tid < G::NH, has_h2 = H2 && tid + NT < G::NH;
  int hi = 0, hj = 0, h2i = 0, h2j = 0;
  // in reverse order: the waves at the end of the workgroup also carry the ring cells, so they get the contiguous rows of the halo and the
  // first waves the columns (one cache line per lane)
  if (has_h) G::halo_x(G::NH - 1 - tid, hi, hj);
  if (has_h2) G::halo_x(G::NH - 1 - (tid + NT), h2i, h2j);
  const int hX = posX(hi, hj), h2X = posX(h2i, h2j);
  const bool h_ok = has_h && out_i(i0 + hi) <= 2 && out_j(j0 + hj) <= 2;          // further out nothing is defined (and nothing is needed)
  const bool h2_ok = has_h2 && out_i(i0 + h2i) <= 2 && out_j(j0 + h2j) <= 2;
  GfColumn hcol = {box, 0}, h2col = {box, 0};
  if (h_ok) hcol = gf_column(L, box, i0 + hi, j0 + hj);
  if (h2_ok) h2col = gf_column(L, box, i0 + h2i, j0 + h2j);
  // a halo cell face-adjacent to the tile: a position B reads on the planes r - 1, r + 1 (the k ghost planes of t need it)
  const bool h_face = h_ok && ((hi >= 0 && hi < TI && (hj == -1 || hj == TJ)) || (hj >= 0 && hj < TJ && (hi == -1 || hi == TI)));
  const bool h2_face = h2_ok && ((h2i >= 0 && h2i < TI && (h2j == -1 || h2j == TJ)) || (h2j >= 0 && h2j < TJ && (h2i == -1 || h2i == TI)));
  // per-lane columns in other boxes are followed by MARCHING pointers (xh_c: the halo column at plane q+2, al_c / xe_c below); the base to
  // continue with above the box (the box above, or the own ghost zone) is needed once per march at most: it waits in LDS where there is
  // room (PARK) and is rebuilt from the column otherwise -- not in registers
  auto x_col = [&](const GfColumn &c) -> gcptr { return sel_origin(L, P.x, c.box) + c.off; };
  auto x_col_up = [&](const GfColumn &c, bool ok) -> gcptr {
    const int m = L.box_nbr[6 * c.box + 5];
    return (ok && m >= 0) ? sel_origin(L, P.x, m) + c.off - (long long)dim * kS : sel_origin(L, P.x, c.box) + c.off;
  };
  if (PARK) sPtr[tid] = (unsigned long long)x_col_up(hcol, h_ok);
  auto xh_up = [&]() -> gcptr { return PARK ? (gcptr)sPtr[tid] : x_col_up(hcol, h_ok); };

  // ---- (3) lanes 0 .. 331: one halo cell of the coefficient planes (what the stencils of the tile and of its face-adjacent ring reach), read
  //      from the OWN box's ghost zone like the reference does: inside the domain that is the neighbour's value (rebuild_operator exchanged
  //      it), outside the domain it is this box's own extrapolation -- which differs from what the neighbouring box holds for the same
  //      place (extrapolate_betas works with box-relative normals), see "special" below
  // ---- (4) lanes 432 .. 511: a red / black pair of ring cells e0 = (ei, ej), e1 = e0 + (edx, edy), in the box that owns them
  const bool has_b = tid < G::NBH, has_e = tid < NRING;
  int bhi = 0, bhj = 0, ei = 0, ej = 0, edx = 0, edy = 0;
  if (has_b) G::halo_b(tid, bhi, bhj);
  if (has_e) {
    const int e = tid;
    // rows -1 and TJ alternate from lane to lane: their red cells sit at columns of opposite parity, so the 32 lanes of a half wave read
    // 32 different LDS banks (one row alone reads every second double: a two-way conflict on every access)
    if (e < TI)          { ej = (e & 1) ? TJ : -1; ei = 2 * (e >> 1); edx = 1; }
    else if (e < TI + 8) { ei = -1; ej = 2 * (e - TI); edy = 1; }
    else                 { ei = TI; ej = 2 * (e - TI - 8); edy = 1; }
  }
  const int bB = posB(bhi, bhj);                                                  // the coefficient halo cell: in LDS, in the own box
  const unsigned b_o = (unsigned)((long long)first + (i0 + bhi) + (long long)(j0 + bhj) * jS);
  const bool e_in = has_e && out_i(i0 + ei) == 0 && out_j(j0 + ej) == 0;          // inside the domain: R forms its red cell
  const bool e_other = e_in && (i0 + ei < 0 || i0 + ei >= dim || j0 + ej < 0 || j0 + ej >= dim);     // ... which belongs to another box
  GfColumn acol = {box, 0};
  if (e_in)  acol = gf_column(L, box, i0 + ei, j0 + ej);
  const int e_step = edx + edy * jS;                                              // second cell of the ring pair, in memory (same box: pairs are aligned)
  const int e_id = has_e ? tid : 0;
  // the ring pair's column: level vectors (al), x (xe), and P.tg, where its t is read when it is "special"; `up`: the base for planes >= dim
  auto al_col = [&](bool up) -> gcptr {
    const int m = L.box_nbr[6 * acol.box + 5];
    return (up && e_in && m >= 0) ? as_global(L.box_base[m]) + first + acol.off - (long long)dim * kS : as_global(L.box_base[acol.box]) + first + acol.off;
  };
  auto xe_col = [&](bool up) -> gcptr { return up ? x_col_up(acol, e_in) : x_col(acol); };
  auto tge_col = [&](bool up) -> gcptr {
    const int m = L.box_nbr[6 * acol.box + 5];
    return (up && e_in && m >= 0) ? sel_origin(L, P.tg, m) + acol.off - (long long)dim * kS : sel_origin(L, P.tg, acol.box) + acol.off;
  };
  if (PARK && has_e) {
    sPtr[NT + e_id] = (unsigned long long)al_col(true); sPtr[NT + NRING + e_id] = (unsigned long long)xe_col(true);
    sPtr[NT + 2 * NRING + e_id] = (unsigned long long)tge_col(false); sPtr[NT + 3 * NRING + e_id] = (unsigned long long)tge_col(true);
  }
  auto al_up = [&]() -> gcptr { return PARK ? (gcptr)sPtr[NT + e_id] : al_col(true); };
  auto xe_up = [&]() -> gcptr { return PARK ? (gcptr)sPtr[NT + NRING + e_id] : xe_col(true); };

  // ---- (5) lanes 0 .. 331: one ghost cell of t outside the domain in i and / or j: kind 1 near, 2 far, 3 the k-edge cell diagonal to a tile corner
  const bool tile_wall = t_ilo || t_ihi || t_jlo || t_jhi;
  const bool iwall_only = (t_ilo || t_ihi) && !(t_jlo || t_jhi);
  // The descriptor (kind, position, inward steps; packed) waits in LDS, not in a register: a value that is live across the loop but used
  // only in this rarely taken block is what the register allocator spills first, and a reload from scratch is a vector-memory load -- it
  // returns behind every prefetch in flight, i.e. costs the wall tiles an HBM round trip per step (they took 1.5 x the time of the
  // others, and with a quarter of the tiles at a wall that set the launch time)
  auto bc_descr = [&](int n) -> int {
    int bc_kind = 0, bc_i = 0, bc_j = 0, bc_si = 0, bc_sj = 0;
    // i walls: near cells of rows -1 .. TJ (a row outside the domain makes it the k-edge cell), far cells of rows 0 .. TJ-1
    for (int side = 0; side < 2 && n >= 0; side++) {
      if (!(side ? t_ihi : t_ilo)) continue;
      if (n < 34) {
        const int col = side ? TI : -1, s = side ? -1 : 1;
        if (n < 18) { const int r = -1 + n; const int oj = out_j(j0 + r);
                      bc_i = col; bc_j = r; bc_si = s;
                      if (oj) { bc_kind = 3; bc_sj = (r < 0) ? 1 : -1; } else bc_kind = 1; }
        else        { bc_kind = 2; bc_i = side ? TI + 1 : -2; bc_j = n - 18; bc_si = s; }
        n = -1;
      } else n -= 34;
    }
    // j walls: near cells of columns -1 .. TI inside the domain, far cells of columns 0 .. TI-1
    for (int side = 0; side < 2 && n >= 0; side++) {
      if (!(side ? t_jhi : t_jlo)) continue;
      if (n < 2 * TI + 2) {
        const int row = side ? TJ : -1, s = side ? -1 : 1;
        if (n < TI + 2) { const int c = -1 + n; if (!out_i(i0 + c)) { bc_kind = 1; bc_i = c; bc_j = row; bc_sj = s; } }
        else            { bc_kind = 2; bc_i = n - (TI + 2); bc_j = side ? TJ + 1 : -2; bc_sj = s; }
        n = -1;
      } else n -= 2 * TI + 2;
    }
    return bc_kind | ((bc_i + 4) << 2) | ((bc_j + 4) << 9) | ((bc_si + 1) << 14) | ((bc_sj + 1) << 16);
  };

  // tiles at an i wall only: lanes 0 .. 3 / TI-4 .. TI-1 of every wave take the near and the far cell of its two rows, lanes 4, 5 /
  // TI-6, TI-5 of the first wave (whose lanes are the ring pairs of rows -1 and TJ) the near cells of those rows
  auto bc_descr_iwall = [&]() -> int {
    const bool lft = t_ilo && li < 6, rgt = t_ihi && li >= TI - 6;
    if (!(lft || rgt)) return 0;
    const int m = lft ? li : TI - 1 - li;
    if (m >= 4 && lj != 0) return 0;
    const int row = (m < 4) ? 2 * lj + (m & 1) : ((m & 1) ? TJ : -1), kind = (m < 2 || m >= 4) ? 1 : 2;
    const int gc = lft ? -kind : TI - 1 + kind, si = lft ? 1 : -1;
    return kind | ((gc + 4) << 2) | ((row + 4) << 9) | ((si + 1) << 14) | ((0 + 1) << 16);
  };
  if (tile_wall) sBC[tid] = iwall_only ? bc_descr_iwall() : bc_descr(tid);        // read after the barriers of the first step
  // t on plane q at a tile / ring cell inside the domain: a red cell from the t ring, a black one is the cell of x
  auto t_at = [&](int ci, int cj, int q) -> double {
    return is_red(i0 + ci, j0 + cj, q) ? sT[slot3(q) * PT + posT(ci, cj)] : sX[slot4(q) * PX + posX(ci, cj)];
  };

  // ---- loaders.  Inside the marching loop only planes >= 0 are fetched (one select between the in-box base and the one for planes
  // >= dim); planes below the box are only met in the prologue of the first chunk and looked up there.
  const int qlo = bottom ? 0 : k0 - 1, qhi = top ? dim - 1 : k1;                  // planes R works on
  auto x_own_any = [&](int cell, int p) -> double {                               // x at the own column, cell 0 / 1 of the pair
    if (p >= 0) return (p > dim + 1 && wall_khi) ? 0.0 : ((p >= dim) ? xb_hi : xb)[own_o + cell * ujS + p * ukS];
    if (nb[4] >= 0) return sel_origin(L, P.x, nb[4])[own_g + cell * jS + (p + dim) * kS];
    return (p < -2) ? 0.0 : xb[(long long)own_o + cell * jS + (long long)p * kS];
  };
  auto lv_any = [&](int id, int cell, int p) -> double {                          // a level vector at the own column, any plane the box or its k neighbours hold
    if (p >= 0) return (((p >= dim) ? lvb_hi : lvb) + (size_t)id * vol)[own_o + cell * ujS + p * ukS];
    if (nb[4] >= 0) return (as_global(L.box_base[nb[4]]) + first)[(size_t)id * vol + own_g + cell * jS + (p + dim) * kS];
    return (lvb + (size_t)id * vol)[(long long)own_o + cell * jS + (long long)p * kS];
  };
  auto xh_any = [&](const GfColumn &c, bool ok, int p) -> double {               // prologue only
    if (!ok) return 0.0;
    if (p >= 0) return (p > dim + 1 && L.box_nbr[6 * c.box + 5] < 0) ? 0.0 : ((p >= dim) ? x_col_up(c, ok) : x_col(c))[p * kS];
    const int m = L.box_nbr[6 * c.box + 4];
    if (m >= 0) return sel_origin(L, P.x, m)[c.off + (p + dim) * kS];
    return (p < -2) ? 0.0 : x_col(c)[p * kS];
  };
  auto al_any = [&](int id, int cell, int p) -> double {                          // prologue only
    if (p >= 0) return al_col(p >= dim)[(size_t)id * vol + cell * e_step + p * kS];
    const int m = L.box_nbr[6 * acol.box + 4];
    if (m >= 0) return (as_global(L.box_base[m]) + first)[(size_t)id * vol + acol.off + cell * e_step + (p + dim) * kS];
    return al_col(false)[(size_t)id * vol + cell * e_step + p * kS];
  };
  auto xe_fwd = [&](int cell, int p) -> double { return xe_col(p >= dim)[cell * e_step + p * kS]; };      // prologue only
  auto e_red = [&](int q) { return is_red(i0 + ei, j0 + ej, q) ? 0 : 1; };         // which cell of the ring pair is red on plane q

  // ---- prologue: planes qlo-2 .. qlo+1 of x, planes qlo-1 .. qlo+1 of beta_i / beta_j, faces qlo, qlo+1 of beta_k into LDS; x[qlo+2] of
  // the own columns and the per-cell streams of plane qlo into registers
  // every load of the prologue is issued before the first value is stored: plane by plane (load, wait, store) the seven planes were seven
  // round trips, ~30 us per workgroup -- 5 % of a 128-plane march, 9 % of a 64-plane one
  {
    double px0[4], px1[4], ph[4], ph2[4];
    double qi0[3], qi1[3], qj0[3], qj1[3], qih[3], qjh[3], qk0[2], qk1[2], qkh[2];
    gcptr gbi = lvb + (size_t)VECTOR_BETA_I * vol, gbj = lvb + (size_t)VECTOR_BETA_J * vol, gbk = lvb + (size_t)VECTOR_BETA_K * vol;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int p = qlo - 2 + m;
      px0[m] = x_own_any(0, p); px1[m] = x_own_any(1, p);
      ph[m] = has_h ? xh_any(hcol, h_ok, p) : 0.0;
      ph2[m] = has_h2 ? xh_any(h2col, h2_ok, p) : 0.0;
    }
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const long long po = (long long)(qlo - 1 + m) * kS;
      qi0[m] = gbi[(long long)own_o + po]; qi1[m] = gbi[(long long)own_o + jS + po];
      qj0[m] = gbj[(long long)own_o + po]; qj1[m] = gbj[(long long)own_o + jS + po];
      qih[m] = has_b ? gbi[(long long)b_o + po] : 0.0; qjh[m] = has_b ? gbj[(long long)b_o + po] : 0.0;
      if (m >= 1) { qk0[m - 1] = gbk[(long long)own_o + po]; qk1[m - 1] = gbk[(long long)own_o + jS + po]; qkh[m - 1] = has_b ? gbk[(long long)b_o + po] : 0.0; }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int s = slot4(qlo - 2 + m) * PX;
      sX[s + ownX] = px0[m]; sX[s + ownX + WX] = px1[m];
      if (has_h) sX[s + hX] = ph[m];
      if (has_h2) sX[s + h2X] = ph2[m];
    }
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const int p = qlo - 1 + m, s = slot3(p) * PB;
      sBI[s + ownB] = qi0[m]; sBI[s + ownB + WB] = qi1[m]; sBJ[s + ownB] = qj0[m]; sBJ[s + ownB + WB] = qj1[m];
      if (has_b) { sBI[s + bB] = qih[m]; sBJ[s + bB] = qjh[m]; }
      if (m >= 1) {
        const int s2 = slot2(p) * PB;
        sBK[s2 + ownB] = qk0[m - 1]; sBK[s2 + ownB + WB] = qk1[m - 1];
        if (has_b) sBK[s2 + bB] = qkh[m - 1];
      }
    }
  }
  double kp2_0 = x_own_any(0, qlo + 2), kp2_1 = x_own_any(1, qlo + 2);           // x two planes above the current one, both cells of the pair
  double c_rhs0 = lv_any(P.rhs_id, 0, qlo), c_rhs1 = lv_any(P.rhs_id, 1, qlo), c_dinv0 = lv_any(VECTOR_DINV, 0, qlo), c_dinv1 = lv_any(VECTOR_DINV, 1, qlo);
  double c_al0 = kHelm ? lv_any(VECTOR_ALPHA, 0, qlo) : 0.0, c_al1 = kHelm ? lv_any(VECTOR_ALPHA, 1, qlo) : 0.0;
  double e_rhs = 0.0, e_dinv = 0.0, e_al = 0.0, e_xp2 = 0.0;                       // ring: the streams of the cell that is red on the current plane, its x two planes up
  if (e_in) {
    const int c = e_red(qlo);
    e_rhs = al_any(P.rhs_id, c, qlo); e_dinv = al_any(VECTOR_DINV, c, qlo); if (kHelm) e_al = al_any(VECTOR_ALPHA, c, qlo);
    e_xp2 = xe_fwd(c, qlo + 2);
  }
  // what B(q) needs from the coefficient planes that will have left the rings by then (beta_i / beta_j plane q-1, beta_k face q), formed a step early
  double pd1 = 0.0, pd3 = 0.0, pd7 = 0.0, pd9 = 0.0, pf4 = 0.0, pd4 = 0.0, pd5 = 0.0;
  double b_rhs = 0.0, b_dinv = 0.0, b_al = 0.0, kmB = 0.0;           // kmB: t three planes below the current one at the cell B works on
  double r_prev = 0.0;                                               // what R formed at the own pair's red cell a step ago
  const double bh2inv = P.b * P.h2inv, nbh2inv = (-P.b) * P.h2inv;

  // marching pointers: the x halo column at plane q+2, the ring pair's level vectors at plane q+1, its x at plane q+3
  gcptr xh_c = ((qlo + 2 >= dim) ? x_col_up(hcol, h_ok) : x_col(hcol)) + (long long)(qlo + 2) * kS;
  gcptr xh2_c = H2 ? ((qlo + 2 >= dim) ? x_col_up(h2col, h2_ok) : x_col(h2col)) + (long long)(qlo + 2) * kS : xh_c;
  gcptr al_c = al_col(qlo + 1 >= dim) + (long long)(qlo + 1) * kS;
  gcptr xe_c = xe_col(qlo + 3 >= dim) + (long long)(qlo + 3) * kS;
  const int qend = top ? dim : qhi;                                                 // at the top of the domain one more step: B(dim-1) after t's ghost plane
#ifdef HPGMG_EXP_TIMELINE
  // wave 0 and the last (ring) wave of one workgroup in the middle of the grid: the 100 MHz clock at nine points of every step
  const bool probe = P.timeline && logical == (P.timeline_wg >= 0 ? P.timeline_wg : P.total_blocks / 2 + 3) && (tid & 63) == 0;
  unsigned long long *tl = P.timeline + (tid >> 6) * 2048;
  int tl_n = 0;
#define TL_MARK() do { if (probe && tl_n < 2040) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TL_MARK() do { } while (0)
#endif
  // the waves that also carry the ring have the longest step and everybody else waits for them at the barriers: they issue first
  if (tid < ((NRING + 63) / 64) * 64) __builtin_amdgcn_s_setprio(2);
  for (int q = qlo; q <= qend; q++) {
    TL_MARK();                                                                      // 0
    __syncthreads();                                                                // [A] the planes stored at the end of the previous step are in place
    TL_MARK();                                                                      // 1
    const bool do_r = q <= qhi, more = q + 1 <= qhi;                                // more: there is an R(q+1)
    const int up = is_red(gi, gj, q) ? 0 : 1;                                      // the red cell of the own pair on plane q: 0 lower, 1 upper
    // ---- "special" red cells of this step (see the head of the file): t comes from P.tg's interior.  These loads are the FIRST of the step, so
    // waiting for them later does not mean waiting for the prefetches issued after them.
    const bool q_out = q < 0 || q >= dim;
    const int row_r = 2 * lj + up, ec = e_red(q), eci = ei + ec * edx, ecj = ej + ec * edy;
    const bool sp_o = do_r && q_out && ((t_ilo && li == 0) || (t_ihi && li == TI - 1) || (t_jlo && row_r == 0) || (t_jhi && row_r == TJ - 1));
    const bool sp_e = do_r && e_in && (e_other || q_out) &&
                      (out_i(i0 + eci - 1) || out_i(i0 + eci + 1) || out_j(j0 + ecj - 1) || out_j(j0 + ecj + 1) || (wall_klo && q == 0) || (wall_khi && q == dim - 1));
    double t_spo = 0.0, t_spe = 0.0;
    if (__builtin_expect(sp_o, 0)) t_spo = (q >= dim) ? tgb_hi[own_o + up * ujS + (long long)q * kS] : sel_origin(L, P.tg, nb[4])[own_g + up * jS + (q + dim) * kS];
    if (__builtin_expect(sp_e, 0)) {
      if (q >= 0) t_spe = (PARK ? (gcptr)sPtr[NT + (q >= dim ? 3 : 2) * NRING + e_id] : tge_col(q >= dim))[ec * e_step + (long long)q * kS];
      else { const int m = L.box_nbr[6 * acol.box + 4]; t_spe = (sel_origin(L, P.tg, m) + acol.off)[ec * e_step + (long long)(q + dim) * kS]; }
    }
    // ---- loads of this step (consumed at its end or in the next step)
    // aux0 .. aux2: the coefficient halo cell (lanes 0 .. NBH-1: beta_i, beta_j, beta_k); eax0 .. eax2: the ring pair's red cell (the last NRING lanes:
    // rhs, Dinv, x two planes up).  Separate registers although no lane has both roles: a second load into a register whose first load
    // may still be in flight makes the wave wait for it -- here the ring waves, at the start of every step, for the prefetches before it
    double n_x0 = 0.0, n_x1 = 0.0, n_hx = 0.0, n_hx2 = 0.0, n_bi0, n_bi1, n_bj0, n_bj1, n_bk0, n_bk1, aux0, aux1, aux2;      // the rest: defined whenever it is used
    double eax0, eax1, eax2;
    double n_rhs0, n_rhs1, n_dinv0, n_dinv1, n_al0 = 0.0, n_al1 = 0.0, ne_al = 0.0;
    // A CU passes these loads on at about 16 B per clock, so the 80 KB of a step take 2 us to ISSUE and the waves served last sat
    // in this block while the first ones computed (timeline: 0.7 us for the first, 2.2 us for the last wave).  So the second half of the
    // workgroup runs its R stage first and issues afterwards, when the first half has moved on: both halves find the path free.
    auto prefetch = [&]() {
      const int p2 = q + 2, p3 = q + 3, p1 = q + 1;
      if (h_ok) n_hx = xh_c[0];
      if (H2 && h2_ok) n_hx2 = xh2_c[0];
      { // coefficients: always the own box's arrays, ghost planes included
        gcptr bi = lvb + (size_t)VECTOR_BETA_I * vol, bj = lvb + (size_t)VECTOR_BETA_J * vol, bk = lvb + (size_t)VECTOR_BETA_K * vol;
        const unsigned o = own_b + (unsigned)p2 * bkS;
        n_bi0 = gld(bi, o); n_bi1 = gld(bi, o + bjS); n_bj0 = gld(bj, o); n_bj1 = gld(bj, o + bjS); n_bk0 = gld(bk, o); n_bk1 = gld(bk, o + bjS);
        if (has_b) { const unsigned ob = b_o * 8u + (unsigned)p2 * bkS; aux0 = gld(bi, ob); aux1 = gld(bj, ob); aux2 = gld(bk, ob); } }
      if (!(wall_khi && p3 > dim + 1)) { gcptr b = (p3 >= dim) ? xb_hi : xb; const unsigned o = own_b + (unsigned)p3 * bkS; n_x0 = gld(b, o); n_x1 = gld(b, o + bjS); }
      { gcptr b = (p1 >= dim) ? lvb_hi : lvb;
        gcptr br = b + (size_t)P.rhs_id * vol, bd = b + (size_t)VECTOR_DINV * vol;
        const unsigned o = own_b + (unsigned)p1 * bkS;
        n_rhs0 = gld(br, o); n_rhs1 = gld(br, o + bjS); n_dinv0 = gld(bd, o); n_dinv1 = gld(bd, o + bjS);
        if (kHelm) { gcptr ba = b + (size_t)VECTOR_ALPHA * vol; n_al0 = gld(ba, o); n_al1 = gld(ba, o + bjS); } }
      if (e_in) {
        const int c = e_red(p1);
        gcptr a = al_c + c * e_step;
        eax0 = a[(size_t)P.rhs_id * vol]; eax1 = a[(size_t)VECTOR_DINV * vol]; if (kHelm) ne_al = a[(size_t)VECTOR_ALPHA * vol];
        eax2 = xe_c[c * e_step];
      }
    };
    const bool late = tid >= NT / 2;                                                // the same for every lane of a wave
    double r_new = 0.0;                                                             // t at the own pair's red cell of plane q
    ldsr X0 = (ldsr)(sX + slot4(q) * PX), Xm = (ldsr)(sX + slot4(q - 1) * PX), Xp = (ldsr)(sX + slot4(q + 1) * PX), Xmm = (ldsr)(sX + slot4(q - 2) * PX);
    ldsr I0 = (ldsr)(sBI + slot3(q) * PB), Im = (ldsr)(sBI + slot3(q - 1) * PB), Ip = (ldsr)(sBI + slot3(q + 1) * PB);
    ldsr J0 = (ldsr)(sBJ + slot3(q) * PB), Jm = (ldsr)(sBJ + slot3(q - 1) * PB), Jp = (ldsr)(sBJ + slot3(q + 1) * PB);
    ldsr K0 = (ldsr)(sBK + slot2(q) * PB), K1 = (ldsr)(sBK + slot2(q + 1) * PB);
    auto gather = [&](X25 &x, int o, double kp2) {                               // the 25 values around position o of the x planes
      constexpr int W = WX;
      x.c = X0[o]; x.im1 = X0[o - 1]; x.ip1 = X0[o + 1]; x.im2 = X0[o - 2]; x.ip2 = X0[o + 2];
      x.jm1 = X0[o - W]; x.jp1 = X0[o + W]; x.jm2 = X0[o - 2 * W]; x.jp2 = X0[o + 2 * W];
      x.km1 = Xm[o]; x.kp1 = Xp[o]; x.km2 = Xmm[o]; x.kp2 = kp2;
      x.mm = X0[o - 1 - W]; x.pm = X0[o + 1 - W]; x.mp = X0[o - 1 + W]; x.pp = X0[o + 1 + W];
      x.m_im = Xm[o - 1]; x.m_ip = Xm[o + 1]; x.m_jm = Xm[o - W]; x.m_jp = Xm[o + W];
      x.p_im = Xp[o - 1]; x.p_ip = Xp[o + 1]; x.p_jm = Xp[o - W]; x.p_jp = Xp[o + W];
    };
    if (more && !late) prefetch();
    TL_MARK();                                                                      // 2: loads issued (first half of the workgroup)
    if (do_r) {
      { // ---- R(q) at the red cell of the own pair (gsrb.c:100-104)
        X25 x; BG1 g1; BG2 g2; Br18 br;
        const int oX = ownX + up * WX, oB = ownB + up * WB;
        FV4RB_FENCE();
        gather(x, oX, up ? kp2_1 : kp2_0);
        beta_g1<WB>(g1, I0 + oB, Im + oB, Ip + oB, J0 + oB, Jm + oB, Jp + oB, K0 + oB, K1 + oB);
        FV4RB_FENCE();
        fv4_brackets(br, x);                    // under the arrival of group 1
        FV4RB_FENCE();
        beta_g2<WB>(g2, I0 + oB, Im + oB, Ip + oB, J0 + oB, Jm + oB, Jp + oB, K0 + oB, K1 + oB);
        FV4RB_FENCE();
        double s1, s2;
        fv4_combine_a(s1, s2, br, g1);          // under the arrival of group 2
        FV4RB_FENCE();
        const double sum = fv4_combine_b(s1, s2, br, g2);
        const double alv = up ? c_al1 : c_al0, rhs = up ? c_rhs1 : c_rhs0, dinv = up ? c_dinv1 : c_dinv0;
        const double Ax = kHelm ? (P.a * alv) * x.c - bh2inv * sum : nbh2inv * sum;
        r_new = x.c + dinv * (rhs - Ax);
        // a special cell (a plane of another box next to a wall).  A real branch (the empty asm keeps it from becoming a select): only there
        // does anything wait for t_spo, which means waiting for every prefetch issued after it
        if (__builtin_expect(sp_o, 0)) { asm volatile("" ::: "memory"); r_new = t_spo; }
        sT[slot3(q) * PT + ownT + up * ST] = r_new;
        // the pair's red cell on plane q is final (x' = t there); it is stored a step later, together with the black cell of its plane
      }
    }
    TL_MARK();                                                                      // 3: R own
    if (more && late) prefetch();
    if (do_r) {
      if (e_in) { // ---- R(q) at the red cell of the ring pair
        X25 x; BG1 g1; BG2 g2; Br18 br;
        const int ci = eci, cj = ecj;
        const int oX = posX(ci, cj), oB = posB(ci, cj);
        FV4RB_FENCE();
        gather(x, oX, e_xp2);
        beta_g1<WB>(g1, I0 + oB, Im + oB, Ip + oB, J0 + oB, Jm + oB, Jp + oB, K0 + oB, K1 + oB);
        FV4RB_FENCE();
        fv4_brackets(br, x);
        FV4RB_FENCE();
        beta_g2<WB>(g2, I0 + oB, Im + oB, Ip + oB, J0 + oB, Jm + oB, Jp + oB, K0 + oB, K1 + oB);
        FV4RB_FENCE();
        double s1, s2;
        fv4_combine_a(s1, s2, br, g1);
        FV4RB_FENCE();
        const double sum = fv4_combine_b(s1, s2, br, g2);
        const double Ax = kHelm ? (P.a * e_al) * x.c - bh2inv * sum : nbh2inv * sum;
        double tv = x.c + e_dinv * (e_rhs - Ax);
        if (__builtin_expect(sp_e, 0)) { asm volatile("" ::: "memory"); tv = t_spe; }
        sT[slot3(q) * PT + posT(ci, cj)] = tv;
      }
    }
    // B(q-1), next, reads nothing of plane q that ANOTHER lane has just written: from that plane it takes its own column's t (r_new) and
    // black-parity cells, which are x.  So there is no barrier between R and B -- the waves of the workgroup may drift between the two stages,
    // one wave's LDS reads under another's arithmetic -- except where the ghost cells of t have to be formed from the neighbours' results:
    // ghost values of t: in i / j on plane q (tiles at a wall); below the domain after R(0); above it after R(dim-1) (the x ring's part:
    // B(dim-2) reads it in this step) and in the extra step (the t ring's part, whose slot B(dim-2) still needed)
    TL_MARK();                                                                      // 4: R ring
    const int kfill = (bottom && q == 0) ? 1 : ((top && q == dim - 1) ? 2 : ((top && q == dim) ? 3 : 0));
    // A tile at an i wall only (a quarter of all tiles; j walls: one in sixteen) needs no barriers for this: its ghost cells are dealt to the
    // lanes so that the four cells one is formed from are results of the SAME wave (bc_descr_iwall; LDS serves a wave's accesses in order);
    // what B(q-1) reads of plane q outside its own column is the near cell of its own row; every other ghost cell of plane q is first read
    // in the next step, behind its barriers; and the positions written (black parity in the x ring, the t ring) are none R(q) of another
    // wave still reads (those have red parity).
    const bool bc_sync = (tile_wall && !iwall_only && do_r) || kfill;
    if (__builtin_expect((tile_wall && do_r) || kfill, 0)) {
      if (bc_sync) __syncthreads();                                                 // [B] t on plane q is complete inside the domain
      TL_MARK();
      const int bc_pack = (tile_wall && do_r) ? sBC[tid] : 0;
      if (bc_pack & 3) {
        const int bc_kind = bc_pack & 3, bc_i = ((bc_pack >> 2) & 127) - 4, bc_j = ((bc_pack >> 9) & 31) - 4, bc_si = ((bc_pack >> 14) & 3) - 1, bc_sj = ((bc_pack >> 16) & 3) - 1;
        // ---- ghost cells of t on plane q outside the domain in i / j: apply_BCs_v4 (boundary_fv.c:262-425) from t itself
        const bool red = is_red(i0 + bc_i, j0 + bc_j, q);
        if (bc_kind == 3) {
          if (!red) {                                                               // read by the diagonal black cell only
            double n4[4];
#pragma unroll 1
            for (int m = 0; m < 4; m++) {
              const int cj = bc_j + (m + 1) * bc_sj;
              n4[m] = v4_near(t_at(bc_i + bc_si, cj, q), t_at(bc_i + 2 * bc_si, cj, q), t_at(bc_i + 3 * bc_si, cj, q), t_at(bc_i + 4 * bc_si, cj, q));
            }
            sX[slot4(q) * PX + posX(bc_i, bc_j)] = v4_near(n4[0], n4[1], n4[2], n4[3]);
          }
        } else {
          const int d = bc_kind;                                                    // the first cell inside the domain is d steps away
          const int wi = bc_i + d * bc_si, wj = bc_j + d * bc_sj;
          const double x1 = t_at(wi, wj, q), x2 = t_at(wi + bc_si, wj + bc_sj, q), x3 = t_at(wi + 2 * bc_si, wj + 2 * bc_sj, q), x4 = t_at(wi + 3 * bc_si, wj + 3 * bc_sj, q);
          if (bc_kind == 1) { const double v = v4_near(x1, x2, x3, x4); if (red) sT[slot3(q) * PT + posT(bc_i, bc_j)] = v; else sX[slot4(q) * PX + posX(bc_i, bc_j)] = v; }
          else if (!red)    sX[slot4(q) * PX + posX(bc_i, bc_j)] = v4_far(x1, x2, x3, x4);
        }
      }
      if (__builtin_expect(kfill != 0, 0)) {
        // ---- the ghost plane of t below / above the domain (P.tg): red-parity own cells into the t ring, black-parity positions over x's
        const int pg = (kfill == 1) ? -1 : dim;
        double t0 = tgb[(long long)own_o + (long long)pg * kS], t1 = tgb[(long long)own_o + jS + (long long)pg * kS];
        asm volatile("" : "+v"(t0), "+v"(t1));             // landed here, on every path: a load still pending at the join would make the common path wait
        const bool red0 = is_red(gi, gj, pg);
        if (kfill != 3) {
          sX[slot4(pg) * PX + ownX + (red0 ? WX : 0)] = red0 ? t1 : t0;
          if (h_face && !is_red(i0 + hi, j0 + hj, pg)) sX[slot4(pg) * PX + hX] = (sel_origin(L, P.tg, hcol.box) + hcol.off)[pg * kS];
          if (H2 && h2_face && !is_red(i0 + h2i, j0 + h2j, pg)) sX[slot4(pg) * PX + h2X] = (sel_origin(L, P.tg, h2col.box) + h2col.off)[pg * kS];
        }
        if (kfill != 2) sT[slot3(pg) * PT + ownT + (red0 ? 0 : ST)] = red0 ? t0 : t1;
        if (kfill == 1) sX[slot4(-2) * PX + ownX + (red0 ? 0 : WX)] = tgb[(long long)own_o + (red0 ? 0 : jS) - 2LL * kS];   // two below: black parity where plane -1 is red
      }
      TL_MARK();
      if (bc_sync) __syncthreads();                                                 // [C]
      TL_MARK();
    } else { TL_MARK(); TL_MARK(); TL_MARK(); }

    // ---- B(r), r = q - 1: the black half sweep at the cell of the own pair that is red on plane q (black on plane r)
    const int r = q - 1;
    if (r >= k0 && r < k1) {
      X25 x; B18 bt;
      FV4RB_FENCE();
      ldsr X0 = (ldsr)(sX + slot4(r) * PX), Xm = (ldsr)(sX + slot4(r - 1) * PX), Xp = (ldsr)(sX + slot4(r + 1) * PX);
      ldsr T0 = (ldsr)(sT + slot3(r) * PT), Tm = (ldsr)(sT + slot3(r - 1) * PT), Tp = (ldsr)(sT + slot3(r + 1) * PT);
      const int row = 2 * lj + up;
      const int o = posX(li, row), oB = posB(li, row);
      constexpr int W = WX, V2 = WB;
      x.c = X0[o];
      x.km2 = kmB;
      x.kp2 = ((ldsr)sX)[slot4(q + 1) * PX + o];
      if (__builtin_expect(top && q == dim, 0)) {                                  // landed before the branch is left: no pending load may reach the common path
        double v = tgb[own_o + up * ujS + (unsigned)(dim + 1) * ukS];
        asm volatile("" : "+v"(v));
        x.kp2 = v;
      }
      x.im1 = T0[posT(li - 1, row)]; x.ip1 = T0[posT(li + 1, row)]; x.jm1 = T0[posT(li, row - 1)]; x.jp1 = T0[posT(li, row + 1)];
      x.km1 = Tm[posT(li, row)];
      x.kp1 = do_r ? r_new : Tp[posT(li, row)];                                     // the own column on plane q: what R has just formed (the extra step at the top of the domain: its ghost value)
      x.im2 = X0[o - 2]; x.ip2 = X0[o + 2]; x.jm2 = X0[o - 2 * W]; x.jp2 = X0[o + 2 * W];
      x.mm = X0[o - 1 - W]; x.pm = X0[o + 1 - W]; x.mp = X0[o - 1 + W]; x.pp = X0[o + 1 + W];
      x.m_im = Xm[o - 1]; x.m_ip = Xm[o + 1]; x.m_jm = Xm[o - W]; x.m_jp = Xm[o + W];
      x.p_im = Xp[o - 1]; x.p_ip = Xp[o + 1]; x.p_jm = Xp[o - W]; x.p_jp = Xp[o + W];
      // coefficients: beta_i / beta_j plane r and beta_k face r+1 are still in the rings, the rest was formed in the previous step
      ldsr I0 = (ldsr)(sBI + slot3(r) * PB + oB), J0 = (ldsr)(sBJ + slot3(r) * PB + oB), K1 = (ldsr)(sBK + slot2(r + 1) * PB + oB);
      const double i00 = I0[0], i01 = I0[1], j00 = J0[0], jw0 = J0[V2], k10 = K1[0], i0p = I0[V2], i0m = I0[-V2], j01 = J0[1], j0m = J0[-1];
      FV4RB_FENCE();
      Br18 br;
      fv4_brackets(br, x);
      FV4RB_FENCE();
      const double i0pp = I0[1 + V2], i0mp = I0[1 - V2], jwp = J0[V2 + 1], jwm = J0[V2 - 1], k1p = K1[1], k1m = K1[-1], k1w = K1[V2], k1mw = K1[-V2];
      FV4RB_FENCE();
      bt.f[0] = i00; bt.f[1] = i01; bt.f[2] = j00; bt.f[3] = jw0; bt.f[4] = pf4; bt.f[5] = k10;
      bt.d[0] = i0p - i0m; bt.d[1] = pd1; bt.d[2] = j01 - j0m; bt.d[3] = pd3; bt.d[4] = pd4; bt.d[5] = pd5;
      bt.d[6] = i0pp - i0mp; bt.d[7] = pd7; bt.d[8] = jwp - jwm; bt.d[9] = pd9;
      bt.d[10] = k1p - k1m; bt.d[11] = k1w - k1mw;
      const double sum = fv4_combine(br, bt);
      const double Ax = kHelm ? (P.a * b_al) * x.c - bh2inv * sum : nbh2inv * sum;
      // both cells of the pair on plane r in two stores that each cover a whole row of the tile: the black one just formed, the red one from
      // R(r) a step ago.  (Stored with its own stage, each colour wrote every second cell of a line, a step apart: the lines went to memory
      // twice -- WRITE_SIZE was 2.14 GB per 512^3 pass for 1.07 GB of output.)
      const double bv = x.c + b_dinv * (b_rhs - Ax);
      gst(outb, own_b + (unsigned)r * bkS, up ? r_prev : bv);
      gst(outb, own_b + bjS + (unsigned)r * bkS, up ? bv : r_prev);
                               // the pair's red cell on plane r: what R(r) formed
    }
    TL_MARK();                                                                      // 5: B
    if (do_r) {
      // for B(q) in the next step, at the OTHER cell of the pair (black on plane q): the terms whose planes will have left the rings by then
      // (after B(q-1), which has just used the previous set)
      ldsr Im = (ldsr)(sBI + slot3(q - 1) * PB), Ip = (ldsr)(sBI + slot3(q + 1) * PB), Jm = (ldsr)(sBJ + slot3(q - 1) * PB), Jp = (ldsr)(sBJ + slot3(q + 1) * PB), K0 = (ldsr)(sBK + slot2(q) * PB);
      const int o2 = ownB + (1 - up) * WB;
      FV4RB_FENCE();
      const double ip0 = Ip[o2], im0 = Im[o2], ip1 = Ip[o2 + 1], im1 = Im[o2 + 1], jp0 = Jp[o2], jm0 = Jm[o2], jpw = Jp[o2 + WB], jmw = Jm[o2 + WB];
      const double k00 = K0[o2], k0p = K0[o2 + 1], k0m = K0[o2 - 1], k0w = K0[o2 + WB], k0mw = K0[o2 - WB];
      FV4RB_FENCE();
      pd1 = ip0 - im0; pd7 = ip1 - im1; pd3 = jp0 - jm0; pd9 = jpw - jmw;
      pf4 = k00; pd4 = k0p - k0m; pd5 = k0w - k0mw;
    }
    TL_MARK();                                                                      // 6: coefficient terms for the next B
    __syncthreads();                                                                // [D] B is done with the plane that is overwritten now
    TL_MARK();                                                                      // 7
    // t three planes below the NEXT plane at the cell B works on then (the other cell of the pair): still in the x ring, in a slot this lane itself overwrites next
    kmB = sX[slot4(q - 2) * PX + ownX + (1 - up) * WX];
    if (more) {
      const int p2 = q + 2, s = slot4(p2) * PX, sb = slot3(p2) * PB, sk = slot2(p2) * PB;
      sX[s + ownX] = kp2_0; sX[s + ownX + WX] = kp2_1;
      if (has_h) sX[s + hX] = n_hx;
      if (has_h2) sX[s + h2X] = n_hx2;
      sBI[sb + ownB] = n_bi0; sBI[sb + ownB + WB] = n_bi1; sBJ[sb + ownB] = n_bj0; sBJ[sb + ownB + WB] = n_bj1; sBK[sk + ownB] = n_bk0; sBK[sk + ownB + WB] = n_bk1;
      if (has_b) { sBI[sb + bB] = aux0; sBJ[sb + bB] = aux1; sBK[sk + bB] = aux2; }
    }
    // t_spo / t_spe stay live to this point, where every load of the step has landed: were their registers reused earlier, the reuse would have
    // to wait for the (possibly still outstanding) special loads, i.e. for all prefetches, in the middle of the step
    asm volatile("" :: "v"(t_spo), "v"(t_spe));
#ifdef HPGMG_EXP_TIMELINE
    if (probe) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    TL_MARK();                                                                      // 8: prefetched planes in LDS
    // ---- next plane
    b_rhs = up ? c_rhs0 : c_rhs1; b_dinv = up ? c_dinv0 : c_dinv1; b_al = up ? c_al0 : c_al1;      // the black cell of plane q: B(q) in the next step
    kp2_0 = n_x0; kp2_1 = n_x1;
    r_prev = r_new;
    c_rhs0 = n_rhs0; c_rhs1 = n_rhs1; c_dinv0 = n_dinv0; c_dinv1 = n_dinv1; c_al0 = n_al0; c_al1 = n_al1;
    if (has_e) { e_rhs = eax0; e_dinv = eax1; e_xp2 = eax2; }
    e_al = ne_al;
    // the marching pointers: one plane up; where a column leaves the top of its box, continue in the box above (parked in LDS)
#ifdef HPGMG_EXP_TIMELINE
    if (P.timeline && tid == 0 && logical < 8192) {
      if (q == qlo) P.timeline[16384 + 3 * logical + 1] = __builtin_amdgcn_s_memrealtime();     // first step done (prologue + one step)
      if (q == qend) P.timeline[16384 + 3 * logical + 2] = __builtin_amdgcn_s_memrealtime();    // last step done
    }
#endif
    xh_c += kS; al_c += kS; xe_c += kS;
    if (H2) xh2_c += kS;
    if (__builtin_expect(q + 3 == dim, 0)) { xh_c = xh_up() + (long long)dim * kS; if (H2) xh2_c = x_col_up(h2col, h2_ok) + (long long)dim * kS; }
    if (__builtin_expect(q + 2 == dim, 0)) al_c = al_up() + (long long)dim * kS;
    if (__builtin_expect(q + 4 == dim, 0)) xe_c = xe_up() + (long long)dim * kS;
  }
}

// ---- the "special" cells (head of the file): t = the red half sweep of x at the listed cells of this rank's boxes, formed exactly as the
// reference forms it for a cell of that box (its own coefficient arrays, ghost zones included; x outside the box from the box that owns it),
// written into the interior of P.tg.  A few thousand cells per level (lines where an internal box face meets a domain wall): one lane per
// cell, plain global loads.
struct Fv4SpecialArgs {
  VecSel x, tg;
  int rhs_id;
  double a, b, h2inv;
  int sweep;
  const int *cells;      // 4 ints per cell: box, i, j, k
  int n;
};
template <int V>
__device__ __forceinline__ void fv4_special_cell(const hpgmg_hip_level &L, const Fv4SpecialArgs &P, int idx) {
  using namespace fv4rb;
  constexpr bool kHelm = (V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  if (idx >= P.n) return;
  const int box = P.cells[4 * idx], i = P.cells[4 * idx + 1], j = P.cells[4 * idx + 2], k = P.cells[4 * idx + 3];
  const int dim = L.dim, jS = L.jStride, kS = L.kStride;
  const size_t vol = (size_t)L.volume, first = (size_t)L.ghosts * (size_t)(1 + jS + kS);
  // x at (i + di, j + dj, k + dk): in the box that owns the place (one hop per axis), its ghost zone where the domain ends
  // the first hop from the cell's own box comes from a copy of its six neighbours read up front (one round trip instead of one per access)
  int nb0[6];
#pragma unroll
  for (int d = 0; d < 6; d++) nb0[d] = L.box_nbr[6 * box + d];
  auto nbr = [&](int b, int d) -> int { return (b == box) ? nb0[d] : L.box_nbr[6 * b + d]; };
  auto xat = [&](int di, int dj, int dk) -> double {
    int b = box, I = i + di, J = j + dj, K = k + dk;
    if (I < 0)         { const int n = nbr(b, 0); if (n >= 0) { b = n; I += dim; } }
    else if (I >= dim) { const int n = nbr(b, 1); if (n >= 0) { b = n; I -= dim; } }
    if (J < 0)         { const int n = nbr(b, 2); if (n >= 0) { b = n; J += dim; } }
    else if (J >= dim) { const int n = nbr(b, 3); if (n >= 0) { b = n; J -= dim; } }
    if (K < 0)         { const int n = nbr(b, 4); if (n >= 0) { b = n; K += dim; } }
    else if (K >= dim) { const int n = nbr(b, 5); if (n >= 0) { b = n; K -= dim; } }
    return sel_origin(L, P.x, b)[I + J * jS + K * kS];
  };
  X25 x;
  x.c = xat(0, 0, 0); x.im1 = xat(-1, 0, 0); x.ip1 = xat(1, 0, 0); x.im2 = xat(-2, 0, 0); x.ip2 = xat(2, 0, 0);
  x.jm1 = xat(0, -1, 0); x.jp1 = xat(0, 1, 0); x.jm2 = xat(0, -2, 0); x.jp2 = xat(0, 2, 0);
  x.km1 = xat(0, 0, -1); x.kp1 = xat(0, 0, 1); x.km2 = xat(0, 0, -2); x.kp2 = xat(0, 0, 2);
  x.mm = xat(-1, -1, 0); x.pm = xat(1, -1, 0); x.mp = xat(-1, 1, 0); x.pp = xat(1, 1, 0);
  x.m_im = xat(-1, 0, -1); x.m_ip = xat(1, 0, -1); x.m_jm = xat(0, -1, -1); x.m_jp = xat(0, 1, -1);
  x.p_im = xat(-1, 0, 1); x.p_ip = xat(1, 0, 1); x.p_jm = xat(0, -1, 1); x.p_jp = xat(0, 1, 1);
  gcptr pc = as_global(L.box_base[box]) + first + i + j * jS + (long long)k * kS;
  double v = x.c;
  const int par0 = (L.box_low[3 * box] ^ L.box_low[3 * box + 1] ^ L.box_low[3 * box + 2] ^ P.sweep) & 1;
  if (((i ^ j ^ k ^ par0) & 1) == 0) {
    B18 bt;
    beta18_global(bt, pc, vol, jS, kS);
    const double sum = fv4_sum(x, bt);
    const double Ax = kHelm ? (P.a * pc[(size_t)VECTOR_ALPHA * vol]) * x.c - (P.b * P.h2inv) * sum : ((-P.b) * P.h2inv) * sum;
    v = x.c + pc[(size_t)VECTOR_DINV * vol] * (pc[(size_t)P.rhs_id * vol] - Ax);
  }
  sel_origin(L, P.tg, box)[i + j * jS + (long long)k * kS] = v;
}
template <int V>
__global__ __launch_bounds__(256) void fv4_special_kernel(const hpgmg_hip_level L, const Fv4SpecialArgs P) {
  fv4_special_cell<V>(L, P, (int)blockIdx.x * 256 + (int)threadIdx.x);
}
// The pre-pass as one launch: the first sp_blocks workgroups take the special cells (long chains of dependent loads: they start first and
// finish under the others), the rest run the tiled half sweep on the planes next to the k walls (fv4_tile_body).  Both write the interior of
// P.tg; where they write the same cell they write the same value.
template <int V, int TJ, int TI>
__global__ __launch_bounds__(TI * TJ) void fv4_rb_prepass_kernel(const hpgmg_hip_level L, const Fv4TileArgs T, const Fv4SpecialArgs S, int sp_blocks) {
  if ((int)blockIdx.x >= sp_blocks) fv4_tile_body<V, FV4_GSRB, TJ, TI, false>(L, T, (int)blockIdx.x - sp_blocks);
  else fv4_special_cell<V>(L, S, (int)blockIdx.x * (TI * TJ) + (int)threadIdx.y * TI + (int)threadIdx.x);
}

}  // namespace hpgmg

// cheby_pair.hpp -- TWO Chebyshev sweeps of the 7-point operator in one pass over the fine level.
//
// One sweep of chebyshev.c:43-99 moves 9 streams per cell (x_n, x_nm1, rhs, Dinv, beta_i/j/k, alpha read,
// x_np1 written); two consecutive sweeps read the same six coefficient streams twice.  This kernel
// computes  x1 = S_a(x0, xm1)  and  x2 = S_b(x1, x0)  together: x0, xm1 and the coefficients are read once,
// x1 and x2 are written once -- 10 streams for two sweeps instead of 18.  The arithmetic per cell is the same
// expression tree as the single-sweep kernels (apply_op_7pt + the update of chebyshev.c:86-95), so the result
// is bit-identical; only the order in which cells are visited changes.
//
// Decomposition (global cell coordinates; boxes are numbered lexicographically, checked by the host):
//   workgroup = NW waves x 64 lanes;  wave w owns ONE 128-cell row  gj = R-1+w  of a slab of NR = NW-2 output
//   rows (waves 0 and NW-1 compute x1 on the two halo rows only), a lane owns 2 adjacent cells (16-byte loads,
//   1 KiB per wave-load); the workgroup marches in +k over KC output planes plus one halo plane at each end.
//   Per step p it computes x1 on plane p and then x2 on plane p-1.
//   * k neighbours and the plane-(p-1) coefficients needed again for x2: registers;
//   * j neighbours of x0 / x1 and the upper beta_j face: the neighbouring wave's row via LDS (x0 and beta_j are
//     deposited one step ahead; one __syncthreads per step, buffers double-buffered);
//   * i neighbours: adjacent lane (shuffle).  Across a 128-cell tile edge x0 is read from memory; x1 there was
//     written beforehand by cheby_pair_edge_kernel (the columns next to interior tile edges, 0.8 % of the cells);
//   * the domain boundary is the in-register Dirichlet rule ghost = -centre (apply_BCs_p1), applied to x0 for
//     the first sweep and to x1 for the second, as the reference does between sweeps.
// x1/x2 go to vectors distinct from x0/xm1 (plugin-private scratch for the first pair of a smooth() call), since
// neighbouring workgroups still read x0/xm1 on their halo.
#pragma once
#include "common.hpp"
#include "stencil_math.hpp"

namespace hpgmg {

struct VecRef { int scratch, id; };                  // vector `id` of the level, or of the plugin's scratch pair
struct PairArgs {
  VecRef x0, xm1, out1, out2;
  int rhs_id;
  double a, b, h2inv, c1a, c2a, c1b, c2b;           // Chebyshev coefficients of the two sweeps
  int sweep_a;                                      // GSRB: number of the first half sweep (its colour; the second is sweep_a + 1)
  int keep_x1;                                      // 0: x1 is not stored (only its tile-edge columns exist, from the pre-pass): the caller declared out1 scratch
  // INTERP variants: x0 is not read as stored but as  prescale * x0 + (coarse parent)  -- interpolation_vcycle
  // (interpolation_p0.c:43) folded into the first sweep pair of the smooth() that follows it in MGVCycle (mg.c:1160-1161)
  hpgmg_hip_level Lc; int coarse_id; double prescale;
  double *const *scr_base;                          // per box: base of 2 scratch vectors (same padded layout)
  const float *const *c32_base;                     // mixed-precision mode: per box, fp32 copies of Dinv, alpha, beta_i, beta_j, beta_k
                                                    // (5 x volume floats, same padded indexing); null in fp64 mode
  int nbi, nbj;                                     // boxes per dimension (lexicographic numbering)
  int Di, Dj, Dk;                                   // cells of the brick this rank owns (= the whole domain on one rank)
  // REMOTE variants (several ranks): faces -i,+i,-j,+j,-k,+k of the brick: 0 = domain boundary (Dirichlet), 1 = owned by another
  // rank.  Across a remote face x0 is known TWO cells deep -- the ghost zone plus `deep` -- and xm1, rhs and the coefficients one
  // cell deep (ghost zone; the normal beta of a high-side ghost cell's far face comes from `deep_beta`), so x1 can be formed on
  // the ghost layer here exactly as the owning rank forms it, and one halo exchange serves both sweeps.
  int rem[6];
  const double *deep;                               // [box][face 0..5][v][u]: x0 two cells outside face f (u, v = the in-face axes in i<j<k order)
  const double *deep_beta;                          // [box][face 1,3,5 -> 0..2][v][u]: beta_i / beta_j / beta_k at local index dim+1
  int tiles_i, slabs_j, chunks_k, KC, per_xcd, total_blocks;
  int edge_blocks, edge_per_xcd;                    // the pre-pass grid (cheby_pair_edge_kernel)
  // the eight coefficient values of every pre-pass cell (beta_i low / high face, beta_j, beta_k, alpha, Dinv), packed [column][k][j][8] once per
  // operator rebuild: the pre-pass walks COLUMNS of the boxes, where each value read in place costs a 64-byte sector of its own (null: read in place)
  double *edge_coef;
  // REMOTE, two-part launches (the halo exchange runs on another stream under part 1): dispatch slot -> workgroup, padded with total_blocks (null: identity)
  const int *order;
};

__device__ __forceinline__ double *pair_vec(const hpgmg_hip_level &L, const PairArgs &A, VecRef r, int box) {
  double *base = r.scratch ? A.scr_base[box] : L.box_base[box];
  return base + (size_t)r.id * (size_t)L.volume + (size_t)L.ghosts * (size_t)(1 + L.jStride + L.kStride);
}
// Boxes narrower than a 128-cell row: the row spans 128/dim boxes that are consecutive in the level's slab, so lane l
// reaches ITS box by adding (l / lanes_per_box) box strides to the address computed for the row's first box.
struct LaneShift { long long lvl, scr, c32; };   // element offsets for level vectors / scratch vectors / fp32 copies
__device__ __forceinline__ long long shift_of(const LaneShift &s, VecRef r) { return r.scratch ? s.scr : s.lvl; }

struct alignas(16) p2 { double x, y; };
// Device memory is addressed as such (address space 1: global_load / global_store), not through generic pointers: a FLAT access counts
// against lgkmcnt as well as vmcnt, and the box-base pointers these kernels look up per plane are scalar loads -- each wait for one of them
// (lgkmcnt) was a wait for every FLAT load already in flight, so the eight streams of a step were fetched one round trip after the other.
typedef double __attribute__((ext_vector_type(2))) d2v;
typedef const d2v __attribute__((address_space(1))) *gd2cptr;
typedef d2v __attribute__((address_space(1))) *gd2ptr;
__device__ __forceinline__ p2 pld(const double *p) { const d2v v = *(gd2cptr)as_global(p); return p2{v.x, v.y}; }
// (non-temporal stores for x1/x2 were measured: 4.15 vs 3.80 ms per F-cycle -- slower; plain stores stay)
__device__ __forceinline__ void pst(double *p, p2 v) { d2v w; w.x = v.x; w.y = v.y; *(gd2ptr)as_global(p) = w; }
__device__ __forceinline__ double gld1(const double *p) { return *as_global(p); }      // one double of device memory
__device__ __forceinline__ p2 pneg(p2 v) { return p2{-v.x, -v.y}; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Coefficient streams.  fp64 mode reads the level's own vectors; the mixed-precision smoother (BASELINE config 5) reads
// fp32 copies -- 4 instead of 8 bytes per value on 5 of the 8 input streams -- and widens them to double, so all
// arithmetic and the iterate stay fp64 (the rounding of a coefficient perturbs the operator by <= 6e-8 relative, and
// only inside the smoother: residual, restriction and interpolation keep the fp64 coefficients).
enum { C32_DINV = 0, C32_ALPHA, C32_BETA_I, C32_BETA_J, C32_BETA_K, C32_COUNT };
template <bool C32> struct CoefStream {
  const double *p64; const float *p32;
  __device__ __forceinline__ CoefStream(const hpgmg_hip_level &L, const PairArgs &A, int box, int vec_id, int slot, const LaneShift &sh = LaneShift{0, 0, 0}) {
    const size_t first = (size_t)L.ghosts * (size_t)(1 + L.jStride + L.kStride);
    if (C32) { p32 = A.c32_base[box] + (size_t)slot * (size_t)L.volume + first + sh.c32; p64 = nullptr; }
    else     { p64 = L.box_base[box] + (size_t)vec_id * (size_t)L.volume + first + sh.lvl; p32 = nullptr; }
  }
  __device__ __forceinline__ p2 pair(int off) const {
    if (C32) { typedef float __attribute__((ext_vector_type(2))) f2v; const f2v f = *(const f2v __attribute__((address_space(1))) *)(p32 + off); return p2{(double)f.x, (double)f.y}; }
    return pld(p64 + off);
  }
  __device__ __forceinline__ double one(int off) const { return C32 ? (double)*(const float __attribute__((address_space(1))) *)(p32 + off) : gld1(p64 + off); }
};

// coefficients of one plane for a lane's two cells
template <int V> struct PlaneCoef { p2 rhs, dinv, al, bi, bjlo, bjhi, bk0, bk1; double bir; };

enum { PAIR_CHEBY = 0, PAIR_GSRB = 1 };
// one smoother update of the pair (c.x, c.y).  PAIR_CHEBY: chebyshev.c:86-95 with `old` = x_{n-1}.  PAIR_GSRB: one
// coloured half sweep of gsrb.c:90-105 -- the cell of the pair whose colour is swept (`first_is_swept` says whether
// that is c.x) becomes c + Dinv (rhs - A c), the other one keeps its value.
template <int V, int SM>
__device__ __forceinline__ p2 pair_update(p2 c, double left, double right, p2 jm, p2 jp, p2 km, p2 kp, p2 old,
                                          const PlaneCoef<V> &q, double a, double b, double h2inv, double c1, double c2, bool first_is_swept) {
  const double Ax0 = apply_op_7pt<V>(c.x, left, c.y, jm.x, jp.x, km.x, kp.x, q.bi.x, q.bi.y, q.bjlo.x, q.bjhi.x, q.bk0.x, q.bk1.x, q.al.x, a, b, h2inv);
  const double Ax1 = apply_op_7pt<V>(c.y, c.x, right, jm.y, jp.y, km.y, kp.y, q.bi.y, q.bir, q.bjlo.y, q.bjhi.y, q.bk0.y, q.bk1.y, q.al.y, a, b, h2inv);
  p2 o;
  if (SM == PAIR_CHEBY) {
    o.x = c.x + c1 * (c.x - old.x) + c2 * q.dinv.x * (q.rhs.x - Ax0);
    o.y = c.y + c1 * (c.y - old.y) + c2 * q.dinv.y * (q.rhs.y - Ax1);
  } else {
    const double nx = c.x + q.dinv.x * (q.rhs.x - Ax0), ny = c.y + q.dinv.y * (q.rhs.y - Ax1);
    o.x = first_is_swept ? nx : c.x;
    o.y = first_is_swept ? c.y : ny;
  }
  return o;
}

constexpr int kPairMaxBoxes = 1024;       // boxes per rank the pair kernel takes (their base-pointer tables are copied into LDS, see below)
template <int V, int NW, bool C32, int SM, bool NARROW, bool INTERP, bool REMOTE = false>
__global__ __launch_bounds__(64 * NW) void cheby_pair_kernel(const hpgmg_hip_level L_in, const PairArgs A_in) {
  static_assert(!(REMOTE && C32) && !(REMOTE && NARROW && INTERP), "the multi-rank variant is built for fp64 coefficients; with narrow boxes the interpolation is not folded in");
  // REMOTE + INTERP: the parent is added to the cells of the brick only -- what arrives in ghost zones and deep planes was packed by its owner with ITS parents already added (pair_halo_kernel)
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  constexpr int NR = NW - 2;
  extern __shared__ p2 pair_lds[];                              // 3 x [2 buffers][NW rows][64 pairs] = NW x 6 KiB
  p2 (*slabX0)[NW][64] = reinterpret_cast<p2 (*)[NW][64]>(pair_lds);
  p2 (*slabX1)[NW][64] = reinterpret_cast<p2 (*)[NW][64]>(pair_lds + 2 * NW * 64);
  p2 (*slabBJ)[NW][64] = reinterpret_cast<p2 (*)[NW][64]>(pair_lds + 4 * NW * 64);
  // The box-base tables (level vectors, scratch vectors, fp32 copies, the coarse level of a folded interpolation) are looked up for every
  // stream of every plane.  From memory that is a load whose result the stream's own load has to wait for -- and the wait, counting
  // loads in flight, also waited for every stream issued before it: the eight streams of a step went out one round trip after the
  // other (timeline: 6 us per step in the load phase).  Copies of the tables in LDS make the lookup a ds_read, which nothing in flight
  // to memory delays, and the level / argument structs below point at them.
  __shared__ double *sTabLvl[kPairMaxBoxes];
  __shared__ double *sTabScr[kPairMaxBoxes];
  __shared__ const float *sTabC32[C32 ? kPairMaxBoxes : 1];
  __shared__ double *sTabCrs[INTERP ? kPairMaxBoxes : 1];
  hpgmg_hip_level L = L_in;
  PairArgs A = A_in;
  {
    const int nt = 64 * NW, tid0 = (int)threadIdx.y * 64 + (int)threadIdx.x;
    for (int bx = tid0; bx < L_in.num_boxes; bx += nt) {
      sTabLvl[bx] = L_in.box_base[bx];
      sTabScr[bx] = A_in.scr_base ? A_in.scr_base[bx] : nullptr;
      if (C32) sTabC32[bx] = A_in.c32_base[bx];
      if (INTERP) sTabCrs[bx] = A_in.Lc.box_base[bx];
    }
    L.box_base = sTabLvl; A.scr_base = sTabScr;
    if (C32) A.c32_base = sTabC32;
    if (INTERP) A.Lc.box_base = sTabCrs;
    __syncthreads();
  }

  int logical = xcd_logical_block((int)blockIdx.x, A.per_xcd);
  if (REMOTE && A.order) logical = A.order[logical];
  if (logical >= A.total_blocks) return;
  int t = logical;
  const int ti = t % A.tiles_i; t /= A.tiles_i;
  const int sj = t % A.slabs_j; t /= A.slabs_j;
  const int ck = t;

  const int lane = (int)threadIdx.x, w = uni((int)threadIdx.y);
  const int bd = L.dim, jS = L.jStride, kS = L.kStride;
  const int gi0 = ti * 128, R = sj * NR, K0 = ck * A.KC;
  const int NRs = (R + NR <= A.Dj) ? NR : A.Dj - R;          // output rows of this slab
  const int KCs = (K0 + A.KC <= A.Dk) ? A.KC : A.Dk - K0;    // output planes of this chunk
  const int gj = R - 1 + w;
  // rows / planes on which x1 is formed, and on which x0 is known: the brick, plus one (two) layers across a remote face
  const int jlo = (REMOTE && A.rem[2]) ? -1 : 0, jhi = (REMOTE && A.rem[3]) ? A.Dj + 1 : A.Dj;
  const int klo = (REMOTE && A.rem[4]) ? -1 : 0, khi = (REMOTE && A.rem[5]) ? A.Dk + 1 : A.Dk;
  auto x0row = [&](int r) { return r >= ((REMOTE && A.rem[2]) ? -2 : 0) && r < ((REMOTE && A.rem[3]) ? A.Dj + 2 : A.Dj); };
  auto x0plane = [&](int g) { return g >= ((REMOTE && A.rem[4]) ? -2 : 0) && g < ((REMOTE && A.rem[5]) ? A.Dk + 2 : A.Dk); };
  const bool row_x1 = (gj >= jlo && gj < jhi && w <= NRs + 1);  // this wave computes x1 on its row
  const bool row_out = (w >= 1 && w <= NRs);                   // ... and x2, and stores both
  // NARROW (boxes of 64, 32 or 16 cells): hop = which of the row's boxes this lane is in; otherwise the shifts fold to zero
  const int lanes_per_box = NARROW ? bd / 2 : 64, hop = NARROW ? lane / lanes_per_box : 0;
  const int bi_ = gi0 / bd, li = NARROW ? 2 * (lane - hop * lanes_per_box) : gi0 - bi_ * bd + 2 * lane;
  LaneShift sh = {0, 0, 0};
  if (NARROW) { sh.lvl = (long long)hop * L.box_stride; sh.scr = (long long)hop * 2 * L.volume; sh.c32 = (long long)hop * C32_COUNT * L.volume; }
  const int gjc = row_x1 ? gj : 0;
  // box row and local row; a ghost row (REMOTE: gj = -1 or Dj) is row -1 / bd of the brick's first / last box row
  const int nbj_ = A.Dj / bd, nbk_ = A.Dk / bd;
  auto jbox = [&](int r) { return !REMOTE ? r / bd : (r < 0 ? 0 : (r >= A.Dj ? nbj_ - 1 : r / bd)); };
  auto kbox = [&](int g) { return !REMOTE ? g / bd : (g < 0 ? 0 : (g >= A.Dk ? nbk_ - 1 : g / bd)); };
  const int bj_ = jbox(gjc), lj = gjc - bj_ * bd;
  const bool ghost_row = REMOTE && (gjc < 0 || gjc >= A.Dj);
  const int row_off = li + lj * jS;                            // offset of this lane's pair inside a plane of its box
  // neighbours in i across the tile edge (lane 0 / lane 63 only)
  const bool left_dom = (gi0 == 0), right_dom = (gi0 + 128 == A.Di);
  const int biL = left_dom ? bi_ : (gi0 - 1) / bd, liL = left_dom ? (REMOTE ? -1 : 0) : (gi0 - 1) - biL * bd;
  const int biR = right_dom ? (gi0 + 127) / bd : (gi0 + 128) / bd, liR = right_dom ? (REMOTE ? bd : 0) : (gi0 + 128) - biR * bd;      // (right_dom: the row's LAST box -- narrow boxes: not its first)
  const bool left_ghost = REMOTE && left_dom && A.rem[0], right_ghost = REMOTE && right_dom && A.rem[1];   // read the ghost column instead of -centre
  // far rows of x0 the two halo waves need from memory (REMOTE: two rows outside the brick = the deep halo)
  const bool far_lo = row_x1 && (w == 0) && x0row(gj - 1);
  const bool far_hi = row_x1 && (w == NRs + 1) && x0row(gj + 1);
  const int gjf = far_lo ? gj - 1 : (far_hi ? gj + 1 : gjc);
  const bool far_deep = REMOTE && (gjf < -1 || gjf > A.Dj);
  const int bjf = jbox(gjf), ljf = gjf - bjf * bd;

  // pointers of the current plane (wave-uniform: one box per row and plane)
  auto box_of = [&](int bi, int bj, int gk) { return uni(bi + A.nbi * (bj + A.nbj * kbox(gk))); };
  auto plane_off = [&](int gk) { return !REMOTE ? (gk % bd) * kS : (gk - kbox(gk) * bd) * kS; };
  // x0 for the pair starting at local (l_i, l_j) of `box` on global plane gk (INTERP: plus the coarse parent, which the two
  // cells of a pair share), and for a single cell
  auto x0_pair = [&](int box, int l_i, int l_j, int gk) -> p2 {
    p2 v = pld(pair_vec(L, A, A.x0, box) + shift_of(sh, A.x0) + l_i + l_j * jS + plane_off(gk));
    if (INTERP && (!REMOTE || (l_j >= 0 && l_j < bd && gk >= 0 && gk < A.Dk))) {
      const int lk = REMOTE ? gk - kbox(gk) * bd : gk % bd;
      // (narrow boxes: the lane's own box of the row -- the coarse boxes are numbered like the fine ones)
      const double c = gld1(vec_origin(A.Lc, NARROW ? box + hop : box, A.coarse_id) + ((l_i >> 1) + (l_j >> 1) * A.Lc.jStride + (lk >> 1) * A.Lc.kStride));
      v.x = A.prescale * v.x + c; v.y = A.prescale * v.y + c;
    }
    return v;
  };
  auto x0_one = [&](int box, int l_i, int l_j, int gk) -> double {
    double v = gld1(pair_vec(L, A, A.x0, box) + (l_i + l_j * jS + plane_off(gk)));
    if (INTERP && (!REMOTE || (l_i >= 0 && l_i < bd && l_j >= 0 && l_j < bd && gk >= 0 && gk < A.Dk))) {
      const int lk = REMOTE ? gk - kbox(gk) * bd : gk % bd;
      v = A.prescale * v + gld1(vec_origin(A.Lc, box, A.coarse_id) + ((l_i >> 1) + (l_j >> 1) * A.Lc.jStride + (lk >> 1) * A.Lc.kStride));
    }
    return v;
  };
  // REMOTE: the same pair on ANY plane x0 is known on -- inside, ghost (-1, Dk) or deep (-2, Dk+1).  A deep plane of a ghost row
  // would be a brick corner: nobody needs it, zeros stand in.
  auto x0_row = [&](int bj, int l_j, int gk) -> p2 {
    if (REMOTE && (gk < -1 || gk > A.Dk)) {
      if (l_j < 0 || l_j >= bd) return p2{0, 0};
      const int box = box_of(bi_, bj, gk) + hop;      // (the lane's own box: the deep planes are stored per box)
      return pld(A.deep + (((size_t)box * 6 + (gk < 0 ? 4 : 5)) * bd + l_j) * bd + li);
    }
    return x0_pair(box_of(bi_, bj, gk), li, l_j, gk);
  };
  // the far row of a halo wave on plane gk: memory, or (REMOTE, two rows outside the brick) the deep halo of the j face
  auto far_row = [&](int gk) -> p2 {
    if (REMOTE && far_deep) {
      if (gk < 0 || gk >= A.Dk) return p2{0, 0};               // corner again
      const int box = box_of(bi_, bjf, gk) + hop;
      return pld(A.deep + (((size_t)box * 6 + (gjf < 0 ? 2 : 3)) * bd + (gk - kbox(gk) * bd)) * bd + li);
    }
    return REMOTE ? x0_row(bjf, ljf, gk) : x0_pair(box_of(bi_, bjf, gk), li, ljf, gk);
  };

  const int P0 = K0 - 1, P1 = K0 + KCs;                         // x1 planes P0..P1 (those x1 is formed on)
  auto in_dom = [&](int gk) { return gk >= klo && gk < khi; };

  p2 x0m = {0, 0}, x0c = {0, 0}, x0p = {0, 0};                  // x0 on planes p-1, p, p+1
  p2 x1m2 = {0, 0}, x1m1 = {0, 0}, x1c = {0, 0};                // x1 on planes p-2, p-1, p
  p2 far_c = {0, 0}, far_n = {0, 0};                            // far x0 row on planes p, p+1 (halo waves)
  p2 bj_c = {0, 0}, bj_n = {0, 0};                              // own beta_j row on planes p, p+1
  PlaneCoef<V> qc = {}, qp = {};                                // coefficients of planes p and p-1

  // ---- prologue: planes pstart-1 and pstart of x0, beta_j of pstart, into registers / LDS
  const int pstart = in_dom(P0) ? P0 : P0 + 1;
  if (row_x1) {
    const int box = box_of(bi_, bj_, pstart), off = row_off + plane_off(pstart);
    x0c = REMOTE ? x0_row(bj_, lj, pstart) : x0_pair(box, li, lj, pstart);
    if (kVC) bj_c = CoefStream<C32>(L, A, box, VECTOR_BETA_J, C32_BETA_J, sh).pair(off);
    if (x0plane(pstart - 1)) x0m = REMOTE ? x0_row(bj_, lj, pstart - 1) : x0_pair(box_of(bi_, bj_, pstart - 1), li, lj, pstart - 1);
    if (far_lo || far_hi) far_c = far_row(pstart);
    slabX0[pstart & 1][w][lane] = x0c;
    slabBJ[pstart & 1][w][lane] = bj_c;
  }
  __syncthreads();

#ifdef HPGMG_EXP_TIMELINE
  // experiment build: wave 1 of the first workgroup records the 100 MHz clock at five points of every step into A.deep (as raw 64-bit counts)
  unsigned long long *tl = (unsigned long long *)A.deep;
  const bool probe = !REMOTE && tl && logical == 0 && w == 1 && lane == 0;
  int tl_n = 0;
#define TL_MARK() do { if (probe && tl_n < 4000) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TL_MARK() do { } while (0)
#endif
  for (int p = pstart; p <= P1 && p < khi; p++) {
    const int box = box_of(bi_, bj_, p), off = row_off + plane_off(p);
    TL_MARK();                                                   // 0: step begins
    const bool have_next = in_dom(p + 1) && (p + 1 <= P1);     // plane p+1 is needed as a centre later
    const bool above_in = x0plane(p + 1);
    const bool ghost_plane = REMOTE && (p < 0 || p >= A.Dk);
    // ---- loads: x0 / beta_j / far row one plane ahead, everything else for plane p
    if (row_x1) {
      if (above_in) {
        const int bn = box_of(bi_, bj_, p + 1), offn = row_off + plane_off(p + 1);
        x0p = REMOTE ? x0_row(bj_, lj, p + 1) : x0_pair(bn, li, lj, p + 1);
        if (have_next) {
          if (kVC) bj_n = CoefStream<C32>(L, A, bn, VECTOR_BETA_J, C32_BETA_J, sh).pair(offn);
          if (far_lo || far_hi) far_n = far_row(p + 1);
        }
      }
      qc.rhs = pld(vec_origin(L, box, A.rhs_id) + sh.lvl + off);
      qc.dinv = CoefStream<C32>(L, A, box, VECTOR_DINV, C32_DINV, sh).pair(off);
      if (kHelm) qc.al = CoefStream<C32>(L, A, box, VECTOR_ALPHA, C32_ALPHA, sh).pair(off);
      if (kVC) {
        const CoefStream<C32> bis(L, A, box, VECTOR_BETA_I, C32_BETA_I, sh), bks(L, A, box, VECTOR_BETA_K, C32_BETA_K, sh);
        qc.bi = bis.pair(off);
        qc.bir = __shfl_down(qc.bi.x, 1, 64);
        if (lane == 63) qc.bir = bis.one(off + 2);
        // the lower face is the previous plane's upper face (same address) unless this is the first plane of a box / of the march
        if (p == pstart || (p % bd) == 0) qc.bk0 = bks.pair(off); else qc.bk0 = qp.bk1;
        // the box's own upper face (ghost plane at the box top); REMOTE: the far face of a ghost cell above the brick is one
        // index beyond the ghost zone and comes from the deep coefficient halo
        if (REMOTE && p >= A.Dk) qc.bk1 = ghost_row ? p2{0, 0} : pld(A.deep_beta + (((size_t)(box + hop) * 3 + 2) * bd + lj) * bd + li);
        else qc.bk1 = bks.pair(off + kS);
        qc.bjlo = bj_c;
        if (REMOTE && gj >= A.Dj) qc.bjhi = ghost_plane ? p2{0, 0} : pld(A.deep_beta + (((size_t)(box + hop) * 3 + 1) * bd + (p - kbox(p) * bd)) * bd + li);
        else if (w == NRs + 1 || gj + 1 >= jhi || ((lj + 1) == bd)) qc.bjhi = CoefStream<C32>(L, A, box, VECTOR_BETA_J, C32_BETA_J, sh).pair(off + jS);
        else qc.bjhi = slabBJ[p & 1][w + 1][lane];
      }
      // x_{n-1} of the first sweep.  When its coefficient c1a is exactly 0 (the first Chebyshev sweep of every smooth(), chebyshev.c:30) the
      // stream is not read: x0 stands in, the term is c1a * 0 = +0, and x0 + (+0) = x0 as with any finite x_{n-1} (the one representable
      // difference: an x0 of exactly -0.0 could come out as +0.0)
      p2 xm1 = x0c;
      if (SM == PAIR_CHEBY && A.c1a != 0.0) xm1 = pld(pair_vec(L, A, A.xm1, box) + shift_of(sh, A.xm1) + off);
#ifdef HPGMG_EXP_TIMELINE
      if (probe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      TL_MARK();                                                 // 1: every load of the step has arrived
#endif

      // ---- x1 on plane p (first sweep): neighbours of x0
      p2 jm, jp;
      if (!x0row(gj - 1)) jm = pneg(x0c); else if (w == 0) jm = far_c; else jm = slabX0[p & 1][w - 1][lane];
      if (!x0row(gj + 1)) jp = pneg(x0c); else if (w == NRs + 1) jp = far_c; else jp = slabX0[p & 1][w + 1][lane];
      const p2 km = x0plane(p - 1) ? x0m : pneg(x0c);
      const p2 kp = above_in ? x0p : pneg(x0c);
      double left = __shfl_up(x0c.y, 1, 64), right = __shfl_down(x0c.x, 1, 64);
      if (lane == 0)  left  = (left_dom && !left_ghost)   ? -x0c.x : x0_one(box_of(biL, bj_, p), liL, lj, p);
      if (lane == 63) right = (right_dom && !right_ghost) ? -x0c.y : x0_one(box_of(biR, bj_, p), liR, lj, p);
      // GSRB: cell (gi, gj, gk) is swept in half sweep s when (gi ^ gj ^ gk ^ s) is even; a pair starts at an even gi
      // (brick origins are multiples of the box size, so brick-local and global parities agree)
      if (!(ghost_row && ghost_plane))                          // a ghost row on a ghost plane is a brick edge: x1 there is never read
        x1c = pair_update<V, SM>(x0c, left, right, jm, jp, km, kp, xm1, qc, A.a, A.b, A.h2inv, A.c1a, A.c2a, ((gj ^ p ^ A.sweep_a) & 1) == 0);
    }
#ifdef HPGMG_EXP_TIMELINE
    if (probe) { asm volatile("" :: "v"(x1c.x)); }
    TL_MARK();                                                   // 2: x1 formed
#endif

    // ---- x2 on plane q = p-1 (second sweep): neighbours of x1; x0 is the older iterate
    const int q = p - 1;
    if (row_out && q >= K0) {
      const int boxq = box_of(bi_, bj_, q), offq = row_off + plane_off(q);
      p2 jm, jp;
      if (gj - 1 < jlo) jm = pneg(x1m1); else jm = slabX1[q & 1][w - 1][lane];
      if (gj + 1 >= jhi) jp = pneg(x1m1); else jp = slabX1[q & 1][w + 1][lane];
      const p2 km = in_dom(q - 1) ? x1m2 : pneg(x1m1);
      const p2 kp = x1c;                                         // x1 was formed on plane p
      double left = __shfl_up(x1m1.y, 1, 64), right = __shfl_down(x1m1.x, 1, 64);
      // across a tile edge -- and, REMOTE, across a remote i face -- x1 was written beforehand by cheby_pair_edge_kernel
      if (lane == 0)  left  = (left_dom && !left_ghost)   ? -x1m1.x : gld1(pair_vec(L, A, A.out1, box_of(biL, bj_, q)) + (liL + lj * jS + plane_off(q)));
      if (lane == 63) right = (right_dom && !right_ghost) ? -x1m1.y : gld1(pair_vec(L, A, A.out1, box_of(biR, bj_, q)) + (liR + lj * jS + plane_off(q)));
      const p2 x2 = pair_update<V, SM>(x1m1, left, right, jm, jp, km, kp, x0m, qp, A.a, A.b, A.h2inv, A.c1b, A.c2b, ((gj ^ q ^ (A.sweep_a + 1)) & 1) == 0);
      pst(pair_vec(L, A, A.out2, boxq) + shift_of(sh, A.out2) + offq, x2);
    }
    if (SM == PAIR_CHEBY && A.keep_x1 && row_out && p >= K0 && p < K0 + KCs) pst(pair_vec(L, A, A.out1, box) + shift_of(sh, A.out1) + off, x1c);   // GSRB keeps no x1

    TL_MARK();                                                   // 3: x2 formed and stored
    // ---- hand this plane's x1 and the next plane's x0 / beta_j to the neighbouring waves
    if (row_x1) {
      slabX1[p & 1][w][lane] = x1c;
      if (have_next) { slabX0[(p + 1) & 1][w][lane] = x0p; slabBJ[(p + 1) & 1][w][lane] = bj_n; }
    }
    __syncthreads();
    TL_MARK();                                                   // 4: barrier passed
    x0m = x0c; x0c = x0p; x1m2 = x1m1; x1m1 = x1c; far_c = far_n; bj_c = bj_n; qp = qc;
  }
#ifdef HPGMG_EXP_TIMELINE
  if (probe) tl[4095] = (unsigned long long)tl_n;
#endif
#undef TL_MARK

  // ---- the last output plane when the chunk ends at the top of the domain: x1 above it is the Dirichlet ghost
  if (P1 >= khi && row_out) {
    const int q = A.Dk - 1;
    if (q >= K0) {
      const int boxq = box_of(bi_, bj_, q), offq = row_off + plane_off(q);
      p2 jm, jp;
      if (gj - 1 < jlo) jm = pneg(x1m1); else jm = slabX1[q & 1][w - 1][lane];
      if (gj + 1 >= jhi) jp = pneg(x1m1); else jp = slabX1[q & 1][w + 1][lane];
      const p2 km = in_dom(q - 1) ? x1m2 : pneg(x1m1);
      const p2 kp = pneg(x1m1);
      double left = __shfl_up(x1m1.y, 1, 64), right = __shfl_down(x1m1.x, 1, 64);
      if (lane == 0)  left  = (left_dom && !left_ghost)   ? -x1m1.x : gld1(pair_vec(L, A, A.out1, box_of(biL, bj_, q)) + (liL + lj * jS + plane_off(q)));
      if (lane == 63) right = (right_dom && !right_ghost) ? -x1m1.y : gld1(pair_vec(L, A, A.out1, box_of(biR, bj_, q)) + (liR + lj * jS + plane_off(q)));
      const p2 x2 = pair_update<V, SM>(x1m1, left, right, jm, jp, km, kp, x0m, qp, A.a, A.b, A.h2inv, A.c1b, A.c2b, ((gj ^ q ^ (A.sweep_a + 1)) & 1) == 0);
      pst(pair_vec(L, A, A.out2, boxq) + shift_of(sh, A.out2) + offq, x2);
    }
  }
}

// x1 on the cell columns next to interior 128-cell tile edges (gi = 128 t - 1 and 128 t), which the pair kernel
// reads as i neighbours of the second sweep.  One lane per cell of those columns (0.8 % of the level), lanes along j.
// REMOTE: also on the ghost columns gi = -1 / Di of remote i faces (columns 2 (tiles_i - 1) and + 1), formed from the ghost
// zone and the deep halo exactly as the owning rank forms its own cells; neighbours across remote j / k faces come from the ghost zone.
// PACK: instead of forming x1, store the cell's eight coefficient values to A.edge_coef (once per operator rebuild); later launches read them there.
template <int V, bool C32, int SM, bool INTERP, bool REMOTE = false, bool PACK = false>
__global__ __launch_bounds__(256) void cheby_pair_edge_kernel(const hpgmg_hip_level L, const PairArgs A) {
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  // logical workgroup order: column, then plane, then block of 64 rows -- dealt to the XCDs in contiguous ranges (common.hpp), so the
  // planes k-1, k, k+1 a cell needs were fetched into the SAME L2 by the neighbouring workgroups (dealt round robin, every plane was
  // fetched by three XCDs)
  const int logical = xcd_logical_block((int)blockIdx.x, A.edge_per_xcd);
  if (logical >= A.edge_blocks) return;
  const int jblocks = (A.Dj + 63) / 64;
  const int gj = (logical % jblocks) * 64 + (int)threadIdx.x, gk = (logical / jblocks) % A.Dk, col = logical / (jblocks * A.Dk);
  if (gj >= A.Dj) return;
  const int ncol_in = 2 * (A.tiles_i - 1);
  int gi;
  if (col < ncol_in) gi = 128 * (col / 2 + 1) - 1 + (col & 1);
  else if (REMOTE) { const int which = (col - ncol_in == 0 && A.rem[0]) ? 0 : 1; gi = which ? A.Di : -1; }
  else return;
  const int bd = L.dim, jS = L.jStride, kS = L.kStride;
  const int nbi_ = A.Di / bd, nbj_ = A.Dj / bd, nbk_ = A.Dk / bd;
  auto clampbox = [&](int c, int D, int nb) { return !REMOTE ? c / bd : (c < 0 ? 0 : (c >= D ? nb - 1 : c / bd)); };
  auto cell = [&](int ci, int cj, int ck, int &box) -> int {
    const int bi = clampbox(ci, A.Di, nbi_), bj = clampbox(cj, A.Dj, nbj_), bk = clampbox(ck, A.Dk, nbk_);
    box = bi + A.nbi * (bj + A.nbj * bk);
    return (ci - bi * bd) + (cj - bj * bd) * jS + (ck - bk * bd) * kS;
  };
  // x0 at a cell that may lie outside the brick: a Dirichlet face gives -centre (apply_BCs_p1); REMOTE: one cell outside a
  // remote face is the ghost zone, two cells outside (only along i, for the ghost columns) the deep halo
  auto x0_at = [&](int ci, int cj, int ck, double centre) -> double {
    if (!REMOTE) { if (ci < 0 || cj < 0 || ck < 0 || ci >= A.Di || cj >= A.Dj || ck >= A.Dk) return -centre; }
    else {
      if ((ci < 0 && !A.rem[0]) || (ci >= A.Di && !A.rem[1]) || (cj < 0 && !A.rem[2]) || (cj >= A.Dj && !A.rem[3]) || (ck < 0 && !A.rem[4]) || (ck >= A.Dk && !A.rem[5])) return -centre;
      if (ci < -1 || ci > A.Di) {
        const int bj = cj / bd, bk = ck / bd, box = ((ci < 0) ? 0 : nbi_ - 1) + A.nbi * (bj + A.nbj * bk);
        return A.deep[(((size_t)box * 6 + (ci < 0 ? 0 : 1)) * bd + (ck - bk * bd)) * bd + (cj - bj * bd)];
      }
    }
    int box; const int idx = cell(ci, cj, ck, box);
    double v = gld1(pair_vec(L, A, A.x0, box) + idx);
    // (REMOTE: a cell outside the brick was packed by its owner with its parent already added)
    if (INTERP && (!REMOTE || (ci >= 0 && ci < A.Di && cj >= 0 && cj < A.Dj && ck >= 0 && ck < A.Dk)))
      v = A.prescale * v + gld1(vec_origin(A.Lc, box, A.coarse_id) + (((ci % bd) >> 1) + ((cj % bd) >> 1) * A.Lc.jStride + ((ck % bd) >> 1) * A.Lc.kStride));
    return v;
  };
  int box; const int idx = cell(gi, gj, gk, box);
  double bi0 = 0, bi1 = 0, bj0 = 0, bj1 = 0, bk0 = 0, bk1 = 0, al = 0, dinv = 0;
  const bool packed = !PACK && !REMOTE && !C32 && kVC && A.edge_coef != nullptr;
  double *const pk = (!REMOTE && !C32 && kVC && A.edge_coef) ? A.edge_coef + ((size_t)(col * A.Dk + gk) * A.Dj + gj) * 8 : nullptr;
  if (packed) {                                               // one 64-byte line per cell, consecutive lanes consecutive lines
    const d2v v0 = *(gd2cptr)as_global(pk), v1 = *(gd2cptr)as_global(pk + 2), v2 = *(gd2cptr)as_global(pk + 4), v3 = *(gd2cptr)as_global(pk + 6);
    bi0 = v0.x; bi1 = v0.y; bj0 = v1.x; bj1 = v1.y; bk0 = v2.x; bk1 = v2.y; al = v3.x; dinv = v3.y;
  }
  const double xc = PACK ? 0.0 : x0_at(gi, gj, gk, 0.0);
  if (kVC && !packed) {
    const CoefStream<C32> bi(L, A, box, VECTOR_BETA_I, C32_BETA_I), bj(L, A, box, VECTOR_BETA_J, C32_BETA_J), bk(L, A, box, VECTOR_BETA_K, C32_BETA_K);
    bi0 = bi.one(idx); bj0 = bj.one(idx); bj1 = bj.one(idx + jS); bk0 = bk.one(idx); bk1 = bk.one(idx + kS);
    // the far face of a ghost cell beyond the high i face is one index past the ghost zone: deep coefficient halo
    if (REMOTE && gi >= A.Di) bi1 = A.deep_beta[(((size_t)box * 3 + 0) * bd + (gk % bd)) * bd + (gj % bd)];
    else bi1 = bi.one(idx + 1);
  }
  if (kHelm && !packed) al = CoefStream<C32>(L, A, box, VECTOR_ALPHA, C32_ALPHA).one(idx);
  if (!packed) dinv = CoefStream<C32>(L, A, box, VECTOR_DINV, C32_DINV).one(idx);
  if (PACK) {
    if (pk) { pk[0] = bi0; pk[1] = bi1; pk[2] = bj0; pk[3] = bj1; pk[4] = bk0; pk[5] = bk1; pk[6] = al; pk[7] = dinv; }
    return;
  }
  const double Ax = apply_op_7pt<V>(xc, x0_at(gi - 1, gj, gk, xc), x0_at(gi + 1, gj, gk, xc), x0_at(gi, gj - 1, gk, xc), x0_at(gi, gj + 1, gk, xc),
                                    x0_at(gi, gj, gk - 1, xc), x0_at(gi, gj, gk + 1, xc), bi0, bi1, bj0, bj1, bk0, bk1, al, A.a, A.b, A.h2inv);
  const double rhs = vec_origin(L, box, A.rhs_id)[idx];
  if (SM == PAIR_CHEBY) {
    const double xnm1 = (A.c1a != 0.0) ? pair_vec(L, A, A.xm1, box)[idx] : xc;     // as in the main kernel: not read when its coefficient is 0
    pair_vec(L, A, A.out1, box)[idx] = xc + A.c1a * (xc - xnm1) + A.c2a * dinv * (rhs - Ax);
  } else {
    pair_vec(L, A, A.out1, box)[idx] = (((gi ^ gj ^ gk ^ A.sweep_a) & 1) == 0) ? xc + dinv * (rhs - Ax) : xc;
  }
}

// fp32 copies of the five coefficient vectors (whole padded boxes: the kernels read ghost faces of the betas)
__global__ __launch_bounds__(256) void coef32_convert_kernel(const hpgmg_hip_level L, float *const *c32_base, int num_vectors) {
  const int box = blockIdx.y;
  const int ids[C32_COUNT] = { VECTOR_DINV, VECTOR_ALPHA, VECTOR_BETA_I, VECTOR_BETA_J, VECTOR_BETA_K };
  const long long n = (long long)C32_COUNT * L.volume;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int slot = (int)(t / L.volume), idx = (int)(t - (long long)slot * L.volume);
    if (ids[slot] < num_vectors)      // Poisson / constant-coefficient builds carry no alpha (no beta) vectors
      c32_base[box][(size_t)slot * (size_t)L.volume + idx] = (float)L.box_base[box][(size_t)ids[slot] * (size_t)L.volume + idx];
  }
}

}  // namespace hpgmg

// brick_wide.hip -- brick_visit.hip's scheme for the operators with WIDE stencils: the 27-point operator (operators.27pt.c:60-91: radius 1, faces + edges +
// corners, apply_BCs_p2, interpolation_p2) and the 4th-order one (operators.fv4.c:55-134: radius 2, faces + edges, apply_BCs_v4, interpolation_v2).  The visits of
// the launch-bound levels of MGVCycle (mg.c:1147-1163: 64^3, 32^3, 16^3 cells) are ONE launch per V-cycle leg: a level is cut into bricks of 8^3 cells, a
// workgroup of 256 lanes each, two cells per lane; the iterate and VECTOR_TEMP live in LDS as images of the brick with a halo of the stencil's radius.  (The one-box
// levels under them ride along: 8^3 cells as ONE brick -- no neighbour, every halo cell a boundary condition --, 4^3 cells as one brick of 4^3, a wave.)
//   DOWN  per level: smooth; residual -> TEMP; restriction(next.R <- TEMP); zero_vector(next.e)
//   UP    per level, coarsest first: interpolation_vcycle (e += P next.e: interpolation_p2.c / interpolation_v2.c, the tensor rule over the 3^3 coarse
//         cells around the parent; the coarse ghost cells of apply_BCs_p2 / _v2 formed on an LDS image of the brick's 4^3 parents + a ring of one); smooth
// What differs from the 7-point bricks:
//  * the halo.  Per exchange a brick publishes every cell of its outer shell (depth = radius) ONCE, as a record indexed by (brick, cell); a neighbour -- face,
//    edge or corner -- fetches the cells of its halo from the owners' records.  Out-of-place GSRB (gsrb.c:94-104 with GSRB_OOP: operators.27pt.c:126,
//    operators.fv4.c:178) changes one colour per half sweep: only that colour travels, the other colour of the halo is copied from the source image.  (Corner
//    cells always travel: a brick must read from EVERY neighbour at every exchange, or a neighbour could overwrite a record slot two exchanges later before it
//    has been read -- the double buffering of brick_visit.hip rests on that.)
//  * the boundary conditions.  A halo cell outside the domain is not exchanged but formed on the image, after the exchange, from the image's in-domain cells
//    (own cells and the halo just received) with the cell routines of block_ops.hpp -- apply_BCs reads the ghost zone of the same box the same way
//    (boundary_fd.c:93-205, boundary_fv.c:262-569: exchange first, conditions second).
//  * the coefficients of the 4th-order operator.  A cell takes 30 values of beta_i/j/k (operators.fv4.c:87-108): kept in registers for the length of a visit they are 36 registers per cell next
//    to the 25 values of the iterate a stencil has in flight.  The three arrays wait in LDS instead, as images of the brick with the one-cell ring the differences reach into (read from the brick's own box, ghost zone included:
//    extrapolate_betas filled it, boundary_fv.c:573-681): 3 x 900 doubles; LDS per workgroup 51 KB, three per CU.
// Records, tags, epochs, the error words and the co-residency guard: brick_records.hpp.  Arithmetic: the expression trees of the per-operator kernels
// (fv4_math.hpp fv4_sum, stencil_direct.hpp apply_op_27pt, block_ops.hpp bc_*_cell / interp_rule, restriction.c:54-57), so every vector is bit-identical to
// the per-operator path; tests/test_gpu_operators.py runs both against the oracle.
#include "stencil_direct.hpp"
#include "dense_levels.hpp"
#include "brick_records.hpp"
#include "fv4_math.hpp"

namespace hpgmg {

typedef double __attribute__((address_space(3))) *wlds;
typedef const double __attribute__((address_space(3))) *wldsc;

enum { BW_CHEBY = 0, BW_GSRB = 1 };                 // (GSRB: out of place, the form these two operators are built with)
enum { BW_DOWN = 0, BW_UP = 1 };
constexpr int kRecordsPerBrick = 512;               // record areas are laid out for bricks of 8^3 cells; a brick of 4^3 uses the first 64 of its 512

// bricks of B^3 cells: B = 8 (256 lanes, two cells each); B = 4 / 2 for a level of 4^3 / 2^3 cells (one brick, one wave, 32 / 4 of its lanes own two cells each)
template <int V, int B_> struct WideGeom {
  static constexpr bool k27 = (V == HPGMG_HIP_27PT_CC);
  static constexpr int B = B_, LB = (B == 8) ? 3 : (B == 4 ? 2 : 1), R = k27 ? 1 : 2, W = B + 2 * R, Plane = W * W, Cells = W * W * W;
  static constexpr int FaceHalo = 6 * B * B * R, EdgeHalo = 12 * B, CornerHalo = k27 ? 8 : 0, Halo = FaceHalo + EdgeHalo + CornerHalo;
  static constexpr int Shell1 = 6 * B * B + 12 * B + (k27 ? 8 : 0);      // cells of the one-cell shell the boundary pass looks at (fv4: not the corners)
  // the three coefficient images (fv4): beta_i holds i 0 .. B, j and k -1 .. B; beta_j and beta_k likewise with their own axis short
  static constexpr int BI_J = B + 1, BI_K = (B + 1) * (B + 2), BJ_J = B + 2, BJ_K = (B + 2) * (B + 1), BK_J = B + 2, BK_K = (B + 2) * (B + 2);
  static constexpr int BetaOne = (B + 1) * (B + 2) * (B + 2), BetaDoubles = k27 ? 0 : 3 * BetaOne;
  static constexpr int CW = B / 2 + 2, CoarseCells = CW * CW * CW;      // UP: the brick's (B/2)^3 parents with a ring of one
  static constexpr int Threads = (B == 8) ? 256 : 64, Owners = B * B * B / 2;
  __device__ __forceinline__ static constexpr int hpos(int li, int lj, int lk) { return (li + R) + W * (lj + R) + Plane * (lk + R); }
};

// the idx-th cell of a halo: faces (depth 1 .. r, B^2 cells each), the 12 edges (the diagonal next to the brick, B cells each), the 8 corners
template <int RR, int B>
__device__ __forceinline__ void halo_cell(int idx, int &li, int &lj, int &lk) {
  constexpr int faces = 6 * B * B * RR, LB = (B == 8) ? 3 : (B == 4 ? 2 : 1);
  if (idx < faces) {
    const int f = idx / (B * B * RR), r = idx - f * (B * B * RR), d = r >> (2 * LB), u = r & (B - 1), v = (r >> LB) & (B - 1), w = (f & 1) ? B + d : -1 - d;
    if (f < 2) { li = w; lj = u; lk = v; } else if (f < 4) { li = u; lj = w; lk = v; } else { li = u; lj = v; lk = w; }
  } else if (idx < faces + 12 * B) {
    const int e = (idx - faces) >> LB, w = (idx - faces) & (B - 1), ax = e >> 2, c1 = (e & 1) ? B : -1, c2 = (e & 2) ? B : -1;
    if (ax == 0) { li = w; lj = c1; lk = c2; } else if (ax == 1) { li = c1; lj = w; lk = c2; } else { li = c1; lj = c2; lk = w; }
  } else {
    const int c = idx - faces - 12 * B;
    li = (c & 1) ? B : -1; lj = (c & 2) ? B : -1; lk = (c & 4) ? B : -1;
  }
}

struct WideLevel {
  hpgmg_hip_level L;
  double h2inv, c1[kBrickMaxSweeps], c2[kBrickMaxSweeps];
  int side, nwg;
};
struct WideArgs {
  WideLevel lv[kBrickMaxLevels];    // the levels of the chain, finest first: level j is worked on by workgroups 0 .. lv[j].nwg - 1
  hpgmg_hip_level C;                // the level below the last one
  int n;
  double a, b;
  int sweeps, e_id, R_id;
  int top_e_zero, below_zero;       // as brick_visit.hip
  BrickRecords Rc;
  int absent_wg;
};

// the ghost cell at image position `pos` (NN axes leave the domain there, all at depth one; steps lead back inside, in axis order) from the image's in-domain
// cells: apply_BCs_p2 (27-point), apply_BCs_v4 (fv4: the near AND the far cell behind every leaving axis), apply_BCs_v2 (the coarse image of the fv4 plugin)
template <int KIND>      // 2: p2, 4: v4, 3: v2, 1: p1 (boundary_fd.c:35-65: -, +, - the cell diagonally inside for a face, an edge, a corner)
__device__ __forceinline__ void ghost_cell(wlds img, int pos, int nn, int s0, int s1, int s2) {
  if (KIND == 1) { img[pos] = ((nn == 2) ? 1.0 : -1.0) * img[pos + s0 + (nn >= 2 ? s1 : 0) + (nn == 3 ? s2 : 0)]; return; }
  if (KIND == 2) { if (nn == 1) bc_p2_cell<1>((wldsc)img, img, pos, s0, 0, 0); else if (nn == 2) bc_p2_cell<2>((wldsc)img, img, pos, s0, s1, 0); else bc_p2_cell<3>((wldsc)img, img, pos, s0, s1, s2); }
  else if (KIND == 4) { if (nn == 1) bc_v4_cell<1>((wldsc)img, img, pos, s0, 0, 0); else if (nn == 2) bc_v4_cell<2>((wldsc)img, img, pos, s0, s1, 0); }      // (no corner cells: the stencil does not read them)
  else { if (nn == 1) bc_v2_cell<1>((wldsc)img, img, pos, s0, 0, 0); else if (nn == 2) bc_v2_cell<2>((wldsc)img, img, pos, s0, s1, 0); else bc_v2_cell<3>((wldsc)img, img, pos, s0, s1, s2); }
}

// A x at the brick cell whose image position is p (27-point: operators.27pt.c:60-91 through apply_op_27pt; fv4: operators.fv4.c:55-134 through fv4_sum, the
// coefficients read from the three LDS images at I, J, K = the cell's position in each)
template <int V, int B>
__device__ __forceinline__ double wide_apply(wldsc x, int p, wldsc I, wldsc J, wldsc K, double alpha, double a, double b, double h2inv) {
  using G = WideGeom<V, B>;
  constexpr int W = G::W, P = G::Plane;
  if constexpr (G::k27) {
    const plane9 m = load_plane(x + (p - P), W), c = load_plane(x + p, W), q = load_plane(x + (p + P), W);
    return apply_op_27pt(m, c, q, a, b, h2inv);
  } else {
    fv4rb::X25 s;
    wldsc c = x + p;
    s.c = c[0]; s.im1 = c[-1]; s.ip1 = c[1]; s.im2 = c[-2]; s.ip2 = c[2];
    s.jm1 = c[-W]; s.jp1 = c[W]; s.jm2 = c[-2 * W]; s.jp2 = c[2 * W];
    s.km1 = c[-P]; s.kp1 = c[P]; s.km2 = c[-2 * P]; s.kp2 = c[2 * P];
    s.mm = c[-1 - W]; s.pm = c[1 - W]; s.mp = c[-1 + W]; s.pp = c[1 + W];
    s.m_im = c[-P - 1]; s.m_ip = c[-P + 1]; s.m_jm = c[-P - W]; s.m_jp = c[-P + W];
    s.p_im = c[P - 1]; s.p_ip = c[P + 1]; s.p_jm = c[P - W]; s.p_jp = c[P + W];
    fv4rb::Br18 r;
    fv4rb::fv4_brackets(r, s);
    constexpr int IJ = G::BI_J, IK = G::BI_K, JJ = G::BJ_J, JK = G::BJ_K, KJ = G::BK_J, KK = G::BK_K;      // image strides (B = 8: 9, 90; 10, 90; 10, 100)
    fv4rb::B18 q;      // operators.fv4.c:87-108 (fv4_math.hpp beta18_global)
    q.f[0] = I[0]; q.f[1] = I[1]; q.f[2] = J[0]; q.f[3] = J[JJ]; q.f[4] = K[0]; q.f[5] = K[KK];
    q.d[0] = I[IJ] - I[-IJ];          q.d[1] = I[IK] - I[-IK];
    q.d[2] = J[1] - J[-1];            q.d[3] = J[JK] - J[-JK];
    q.d[4] = K[1] - K[-1];            q.d[5] = K[KJ] - K[-KJ];
    q.d[6] = I[1 + IJ] - I[1 - IJ];   q.d[7] = I[1 + IK] - I[1 - IK];
    q.d[8] = J[JJ + 1] - J[JJ - 1];   q.d[9] = J[JJ + JK] - J[JJ - JK];
    q.d[10] = K[KK + 1] - K[KK - 1];  q.d[11] = K[KK + KJ] - K[KK - KJ];
    const double sum = fv4rb::fv4_combine(r, q);
    if (V == HPGMG_HIP_FV4_VC_HELMHOLTZ) return (a * alpha) * s.c - (b * h2inv) * sum;
    return ((-b) * h2inv) * sum;
  }
}

// B = 8: 256 lanes, TWO cells each (cells 2t and 2t + 1: neighbours in i, so one red and one black): three workgroups on a CU (what the LDS allows) are then 3 waves
// per SIMD with 168 registers each -- with 512 lanes the 80 registers of 6 waves per SIMD spilled (fv4: 25 values of the iterate in flight; 55-78 registers to
// scratch).  A GSRB half sweep updates exactly one of a lane's two cells: no lane idles.
template <int V, int B> constexpr bool wide_ok() { return B == 8 || B == 4 || (B == 2 && V == HPGMG_HIP_27PT_CC); }
template <int V, int SM, int DIR, int B>
__global__ __launch_bounds__((WideGeom<V, B>::Threads), (B == 8 ? 3 : 1)) void brick_wide_kernel(const WideArgs A) {
  using G = WideGeom<V, B>;
  constexpr bool k27 = G::k27, kHelm = (V == HPGMG_HIP_FV4_VC_HELMHOLTZ);
  constexpr int R = G::R, W = G::W, P = G::Plane, LB = G::LB, kThreads = G::Threads, kBcKind = k27 ? 2 : 4, kCoarseBc = k27 ? (B == 2 ? 1 : 2) : 3, kInterp = k27 ? 2 : 3;      // (the coarse level of a 2^3 level is ONE cell: apply_BCs_p1, as interpolation_p2 takes it there)
  constexpr int kSlots = (G::Halo + kThreads - 1) / kThreads, kShellRounds = (G::Shell1 + kThreads - 1) / kThreads, kCellUnroll = k27 ? 2 : 1;
  constexpr int CW = G::CW, kCoarseCells = G::CoarseCells, H = B / 2;
  constexpr bool kUp = (DIR == BW_UP), kDown = !kUp, kAllOwn = (G::Owners == kThreads);
  static_assert(kCoarseCells <= kThreads && H * H * H <= kThreads, "lane roles");
  extern __shared__ double wide_lds[];
  const wlds sx = (wlds)wide_lds, st = sx + G::Cells;
  const wlds sbi = st + G::Cells, sbj = sbi + G::BetaOne, sbk = sbj + G::BetaOne;      // (fv4 only)
  const wlds sc = st + G::Cells + G::BetaDoubles;                                      // (UP only)
  const int t = (int)threadIdx.x, wg = (int)blockIdx.x, e_id = A.e_id, R_id = A.R_id, n = A.n;
  const unsigned epoch = A.Rc.epoch;
  const unsigned *const err_dev = A.Rc.error_dev;
  const RecordWindow RW = record_window(A.Rc);      // the record areas as a buffer (brick_records.hpp)
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool gave_up = false;
  if (wg == A.absent_wg) return;
  const bool owner = kAllOwn || t < G::Owners;      // the lane owns two cells (B = 4: the first 32 lanes of the wave)
  const int to = owner ? t : 0;
  const int li0 = 2 * (to & (H - 1)), lj0 = (to >> (LB - 1)) & (B - 1), lk0 = to >> (2 * LB - 1), pos0 = G::hpos(li0, lj0, lk0);      // the lane's cells: (li0, lj0, lk0) and (li0 + 1, lj0, lk0)
  const int cell0 = li0 + B * (lj0 + B * lk0);      // ... and the first one's number in the brick (= 2 t)
  const bool jk_shell = (lj0 < R || lj0 >= B - R || lk0 < R || lk0 >= B - R), jk_corner = (lj0 == 0 || lj0 == B - 1) && (lk0 == 0 || lk0 == B - 1);
  bool own_shell[2], own_corner[2];
#pragma unroll
  for (int m = 0; m < 2; m++) { const int li = li0 + m; own_shell[m] = jk_shell || li < R || li >= B - R; own_corner[m] = k27 && jk_corner && (li == 0 || li == B - 1); }

  for (int step = 0; step < n; step++) {
    const int j = kUp ? n - 1 - step : step;
    const WideLevel &T = A.lv[j];
    const int nwg = T.nwg;
    if (wg >= nwg) { if (kUp) continue; else break; }
    const bool first = (j == 0), last = (j == n - 1);
    const hpgmg_hip_level &L = T.L;
    const hpgmg_hip_level &C = last ? A.C : A.lv[last ? j : j + 1].L;
    const int side = T.side, bx = wg % side, by = (wg / side) % side, bz = wg / (side * side);
    const int D = L.dim_i, o_i = bx * B, o_j = by * B, o_k = bz * B;
    const LevelGeom GL = geom_of(L), GC = geom_of(C);
    const int gi = o_i + li0, gj = o_j + lj0, gk = o_k + lk0;
    const int colour0 = (gi ^ gj ^ gk) & 1;                 // of the lane's first cell; the second one has the other colour
    const double a = A.a, b = A.b, h2inv = T.h2inv;
    FaceCell *const faces = A.Rc.faces + (size_t)j * kFaceRecords;
    const bool e_zero = kDown && (first ? (A.top_e_zero != 0) : true);
    const bool rhs_by_record = kDown && !first, parent_by_record = kUp && !last;
    const int side_c = last ? 1 : A.lv[last ? j : j + 1].side;
    const bool at_wall = (bx == 0 || by == 0 || bz == 0 || bx == side - 1 || by == side - 1 || bz == side - 1);
    auto below_record = [&](int ci, int cj, int ck) -> size_t {      // the record of cell (ci, cj, ck) of the level below (bricks of that level are numbered like ours)
      const int qx = ci >> LB, qy = cj >> LB, qz = ck >> LB;
      return (size_t)(qx + side_c * (qy + side_c * qz)) * kRecordsPerBrick + (size_t)((ci & (B - 1)) + B * ((cj & (B - 1)) + B * (ck & (B - 1))));
    };

    // ---- what this lane does for the halo: kSlots cells of it.  h_state bit 0: a neighbour's cell (else: none, or outside the domain -- the boundary pass forms
    // it), bit 1: its colour, bit 2: a corner cell
    int h_pos[kSlots], h_rec[kSlots], h_state[kSlots];
#pragma unroll
    for (int m = 0; m < kSlots; m++) {
      const int idx = t + m * kThreads;
      h_pos[m] = 0; h_rec[m] = 0; h_state[m] = 0;
      if (idx >= G::Halo) continue;
      int li, lj, lk;
      halo_cell<R, B>(idx, li, lj, lk);
      const int hi = o_i + li, hj = o_j + lj, hk = o_k + lk;
      h_pos[m] = G::hpos(li, lj, lk);
      if (hi < 0 || hi >= D || hj < 0 || hj >= D || hk < 0 || hk >= D) continue;
      h_rec[m] = ((hi >> LB) + side * ((hj >> LB) + side * (hk >> LB))) * kRecordsPerBrick + ((hi & (B - 1)) + B * ((hj & (B - 1)) + B * (hk & (B - 1))));
      h_state[m] = 1 | (((hi ^ hj ^ hk) & 1) << 1) | ((idx >= G::FaceHalo + G::EdgeHalo) ? 4 : 0);
    }
    int exchange_n = 0;
    // one exchange: the shell of img goes out, the neighbours' cells come into its halo.  colour >= 0 (a GSRB half sweep): only cells of that colour have
    // changed -- the others are taken from the image the half sweep read (corners always travel, see the head of the file).  A level of ONE brick: nothing to trade.
    auto exchange = [&](wlds img, wldsc from, int colour) {
      if (side == 1) return;
      const int par = exchange_n & 1;
      const unsigned seq = epoch + SEQ_FACES + (unsigned)(12 * j + exchange_n);
      exchange_n++;
#pragma unroll
      for (int m = 0; m < 2; m++)
        if (owner && own_shell[m] && (colour < 0 || (colour0 ^ m) == colour || own_corner[m])) face_store(RW, faces + ((size_t)par * nwg + wg) * kRecordsPerBrick + (cell0 + m), img[pos0 + m], seq);
      const FaceCell *want[kSlots];
      double got[kSlots];
      unsigned pending = 0;
#pragma unroll
      for (int m = 0; m < kSlots; m++) {
        want[m] = faces + (size_t)par * nwg * kRecordsPerBrick + h_rec[m];
        got[m] = 0.0;
        if (!(h_state[m] & 1)) continue;
        if (colour >= 0 && ((h_state[m] >> 1) & 1) != colour && !(h_state[m] & 4)) img[h_pos[m]] = from[h_pos[m]];
        else pending |= 1u << m;
      }
      const unsigned polled = pending;
      record_wait_many<kSlots>(RW, want, pending, seq, got, t0, gave_up, err_dev);      // all of the lane's records in flight together
#pragma unroll
      for (int m = 0; m < kSlots; m++) if ((polled >> m) & 1u) img[h_pos[m]] = got[m];
      __syncthreads();
    };
    // apply_BCs on the image: the cells of the one-cell shell around the brick, dealt to the lanes; where such a cell lies outside the domain (all its leaving axes
    // at depth one by construction) it is formed from the in-domain cells behind it -- fv4: not at the corner positions (NO_CORNERS, operators.fv4.c:132)
    auto boundary = [&](wlds img) {
      if (!at_wall) return;
#pragma unroll
      for (int q = 0; q < kShellRounds; q++) {
        const int idx = t + q * kThreads;
        if (idx >= 6 * B * B + 12 * B + (k27 ? 8 : 0)) continue;
        int li, lj, lk;
        halo_cell<1, B>(idx, li, lj, lk);
        const bool oi = (li < 0 && bx == 0) || (li >= B && bx == side - 1), oj = (lj < 0 && by == 0) || (lj >= B && by == side - 1), ok = (lk < 0 && bz == 0) || (lk >= B && bz == side - 1);
        const int nn = (int)oi + (int)oj + (int)ok;
        if (nn == 0) continue;
        const int si = oi ? (li < 0 ? 1 : -1) : 0, sj = oj ? (lj < 0 ? W : -W) : 0, sk = ok ? (lk < 0 ? P : -P) : 0;
        const int s0 = oi ? si : (oj ? sj : sk), s1 = oi ? (oj ? sj : sk) : sk, s2 = sk;      // the leaving axes in axis order
        ghost_cell<kBcKind>(img, G::hpos(li, lj, lk), nn, s0, s1, s2);
      }
      __syncthreads();
    };

    // ---- the brick: iterate, VECTOR_TEMP, coefficients (every load is in flight before anything is waited for)
    const CellRef own = locate(GL, gi, gj, gk);      // (the second cell: own.ijk + 1 -- a brick lies inside one box)
    double e_st[2] = {0.0, 0.0}, rhs[2] = {0.0, 0.0}, dinv[2] = {0.0, 0.0}, alpha[2] = {0.0, 0.0};
    if (owner) {
#pragma unroll
      for (int m = 0; m < 2; m++) {
        e_st[m] = e_zero ? 0.0 : vec_origin(L, own.box, e_id)[own.ijk + m];
        st[pos0 + m] = vec_origin(L, own.box, VECTOR_TEMP)[own.ijk + m];
        rhs[m] = rhs_by_record ? 0.0 : vec_origin(L, own.box, R_id)[own.ijk + m];
        dinv[m] = vec_origin(L, own.box, VECTOR_DINV)[own.ijk + m];
        alpha[m] = kHelm ? vec_origin(L, own.box, VECTOR_ALPHA)[own.ijk + m] : 0.0;
      }
    }
    if constexpr (!k27) {
      // the three coefficient images: the brick and the one-cell ring the differences reach into, as the brick's OWN box holds them
      const CellRef org = locate(GL, o_i, o_j, o_k);
      const int jS = L.jStride, kS = L.kStride;
      const double *bi = vec_origin(L, org.box, VECTOR_BETA_I) + org.ijk, *bj = vec_origin(L, org.box, VECTOR_BETA_J) + org.ijk, *bk = vec_origin(L, org.box, VECTOR_BETA_K) + org.ijk;
      constexpr int one = G::BetaOne;
      // four loads of a lane in flight at a time (all eleven at once cost 60 registers' worth of scratch traffic; one at a time is a round trip each)
#pragma unroll 1
      for (int base = 0; base < 3 * one; base += 4 * kThreads) {
        double v[4];
        int slot[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int idx = base + u * kThreads + t;
          slot[u] = -1; v[u] = 0.0;
          if (idx >= 3 * one) continue;
          const int arr = idx / one, r = idx - arr * one;
          slot[u] = idx;
          if (arr == 0)      { const int i = r % (B + 1), jj = (r / (B + 1)) % (B + 2) - 1, k = r / G::BI_K - 1;  v[u] = bi[i + jj * jS + k * kS]; }
          else if (arr == 1) { const int i = r % (B + 2) - 1, jj = (r / (B + 2)) % (B + 1), k = r / G::BJ_K - 1;  v[u] = bj[i + jj * jS + k * kS]; }
          else               { const int i = r % (B + 2) - 1, jj = (r / (B + 2)) % (B + 2) - 1, k = r / G::BK_K; v[u] = bk[i + jj * jS + k * kS]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (slot[u] >= 0) sbi[slot[u]] = v[u];      // (the three images are contiguous: sbi, sbj, sbk)
      }
    }
    const wldsc cI = (wldsc)sbi + (li0 + G::BI_J * (lj0 + 1) + G::BI_K * (lk0 + 1)), cJ = (wldsc)sbj + ((li0 + 1) + G::BJ_J * lj0 + G::BJ_K * (lk0 + 1)),
                cK = (wldsc)sbk + ((li0 + 1) + G::BK_J * (lj0 + 1) + G::BK_K * lk0);

    double start[2] = { e_st[0], e_st[1] };
    if (kUp) {
      // interpolation_vcycle: e = 1.0 * e + (the tensor rule over the 3^3 coarse cells around the parent).  The brick's (B/2)^3 parents and a ring of one wait in
      // sc: in-domain cells from the level below (its bricks' records, or memory below the last level of the chain), the others by apply_BCs_p2 / _v2
      const int Dc = C.dim_i;
      if (parent_by_record) {      // one lane watches the gate of the brick that holds this brick's parents: the long wait
        if (t == 0) (void)record_wait(RW, A.Rc.gate + (size_t)(j + 1) * kBrickMaxWgs + ((bx >> 1) + side_c * ((by >> 1) + side_c * (bz >> 1))), epoch + SEQ_GATE + (unsigned)(j + 1), t0, gave_up, err_dev, 8);
        __syncthreads();
      }
      const int cq_i = t % CW, cq_j = (t / CW) % CW, cq_k = t / (CW * CW);
      const int ci = (o_i >> 1) - 1 + cq_i, cj = (o_j >> 1) - 1 + cq_j, ck = (o_k >> 1) - 1 + cq_k;
      const bool oi = (ci < 0 || ci >= Dc), oj = (cj < 0 || cj >= Dc), ok = (ck < 0 || ck >= Dc);
      if (t < kCoarseCells && !(oi || oj || ok)) {
        double v;
        if (parent_by_record) v = record_wait(RW, A.Rc.up + (size_t)(j + 1) * kCellRecords + below_record(ci, cj, ck), epoch + SEQ_UP + (unsigned)(j + 1), t0, gave_up, err_dev);
        else { const CellRef c = locate(GC, ci, cj, ck); v = vec_origin(C, c.box, e_id)[c.ijk]; }
        sc[t] = v;
      }
      __syncthreads();
      if (at_wall) {
        const int nn = (int)oi + (int)oj + (int)ok;
        if (t < kCoarseCells && nn > 0) {
          const int si = oi ? (ci < 0 ? 1 : -1) : 0, sj = oj ? (cj < 0 ? CW : -CW) : 0, sk = ok ? (ck < 0 ? CW * CW : -CW * CW) : 0;
          const int s0 = oi ? si : (oj ? sj : sk), s1 = oi ? (oj ? sj : sk) : sk, s2 = sk;
          ghost_cell<kCoarseBc>(sc, t, nn, s0, s1, s2);
        }
        __syncthreads();
      }
      {
        const wldsc c = (wldsc)sc + (((li0 >> 1) + 1) + CW * (((lj0 >> 1) + 1) + CW * ((lk0 >> 1) + 1)));      // the parent (the same for both cells)
        double tk[2][3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
          double tj[2][3];
#pragma unroll
          for (int jj = 0; jj < 3; jj++) {
            double line[3];
#pragma unroll
            for (int ii = 0; ii < 3; ii++) line[ii] = c[(ii - 1) + (jj - 1) * CW + (kk - 1) * CW * CW];
            tj[0][jj] = interp_rule<kInterp>(false, line);      // the first cell has an even i, the second an odd one
            tj[1][jj] = interp_rule<kInterp>(true, line);
          }
          tk[0][kk] = interp_rule<kInterp>((lj0 & 1) != 0, tj[0]);
          tk[1][kk] = interp_rule<kInterp>((lj0 & 1) != 0, tj[1]);
        }
#pragma unroll
        for (int m = 0; m < 2; m++) start[m] = 1.0 * e_st[m] + interp_rule<kInterp>((lk0 & 1) != 0, tk[m]);
      }
    }
    if (rhs_by_record && owner) {
#pragma unroll
      for (int m = 0; m < 2; m++) rhs[m] = record_wait(RW, A.Rc.down + (size_t)j * kCellRecords + (size_t)wg * kRecordsPerBrick + (size_t)(cell0 + m), epoch + SEQ_DOWN + (unsigned)j, t0, gave_up, err_dev);
    }
    if (e_zero) {
      // zero_vector came before (mg.c:1153): the whole image is +0.0, ghost cells included (the conditions of a zero field are zeros)
      for (int z = t; z < G::Cells; z += kThreads) sx[z] = 0.0;
      __syncthreads();
    } else {
      if (owner) { sx[pos0] = start[0]; sx[pos0 + 1] = start[1]; }
      __syncthreads();
      exchange(sx, (wldsc)sx, -1);
      boundary(sx);
    }

    // ---- smooth(): chebyshev.c:43-99 / gsrb.c:24-132 out of place (an even number of sweeps: the result ends in sx, its predecessor in st)
    for (int s = 0; s < A.sweeps; s++) {
      const wldsc src = (s & 1) ? (wldsc)st : (wldsc)sx;
      const wlds dst = (s & 1) ? sx : st;
      if (owner) {
        if (SM == BW_GSRB) {
          // the cell whose global parity equals the half sweep's is updated (box.low folded in, gsrb.c:55), the other one copied (gsrb.c:96-99)
          const int mu = (colour0 == (s & 1)) ? 0 : 1;
          const double xu = src[pos0 + mu], xo = src[pos0 + (mu ^ 1)];
          const double Ax = wide_apply<V, B>(src, pos0 + mu, cI + mu, cJ + mu, cK + mu, mu ? alpha[1] : alpha[0], a, b, h2inv);
          dst[pos0 + mu] = xu + (mu ? dinv[1] : dinv[0]) * ((mu ? rhs[1] : rhs[0]) - Ax);
          dst[pos0 + (mu ^ 1)] = xo;
        } else {
          const double c1 = T.c1[s], c2 = T.c2[s];
          // (fv4: one cell after the other -- two stencils' worth of operands in flight, 2 x 55 values, is more than the registers hold)
#pragma unroll kCellUnroll
          for (int m = 0; m < 2; m++) {
            const double xc = src[pos0 + m];
            const double Ax = wide_apply<V, B>(src, pos0 + m, cI + m, cJ + m, cK + m, m ? alpha[1] : alpha[0], a, b, h2inv);
            const double xnm1 = dst[pos0 + m];
            dst[pos0 + m] = xc + c1 * (xc - xnm1) + c2 * (m ? dinv[1] : dinv[0]) * ((m ? rhs[1] : rhs[0]) - Ax);
          }
        }
      }
      __syncthreads();
      if (kDown || s + 1 < A.sweeps) {      // (the way down goes on to the residual of the result)
        exchange(dst, src, SM == BW_GSRB ? (s & 1) : -1);
        boundary(dst);
      }
    }

    if (kDown) {                                        // residual -> TEMP (residual.c:42-48)
      if (owner) {
#pragma unroll kCellUnroll
        for (int m = 0; m < 2; m++) {
          const double Ax = wide_apply<V, B>((wldsc)sx, pos0 + m, cI + m, cJ + m, cK + m, m ? alpha[1] : alpha[0], a, b, h2inv);
          st[pos0 + m] = (m ? rhs[1] : rhs[0]) - Ax;      // (each lane overwrites only its own TEMP cells: no hazard with the reads of sx)
        }
      }
      __syncthreads();
    }

    // ---- leave e and TEMP in memory as the per-operator sequence would; UP: hand the correction to the finer level of the chain
    if (owner) {
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const double xv = sx[pos0 + m];
        vec_origin(L, own.box, e_id)[own.ijk + m] = xv;
        vec_origin(L, own.box, VECTOR_TEMP)[own.ijk + m] = st[pos0 + m];
        if (kUp && !first) face_store(RW, A.Rc.up + (size_t)j * kCellRecords + (size_t)wg * kRecordsPerBrick + (size_t)(cell0 + m), xv, epoch + SEQ_UP + (unsigned)j);
      }
    }
    if (kUp && !first && t == 0) face_store(RW, A.Rc.gate + (size_t)j * kBrickMaxWgs + wg, 0.0, epoch + SEQ_GATE + (unsigned)j);

    if (kDown) {
      // restriction(coarse.R <- TEMP): 0.125 * the 8 children in the reference's order (restriction.c:54-57); this brick's (B/2)^3 coarse cells
      if (t < H * H * H) {
        const int ci = t & (H - 1), cj = (t >> (LB - 1)) & (H - 1), ck = t >> (2 * (LB - 1));
        const wldsc f = (wldsc)st + G::hpos(2 * ci, 2 * cj, 2 * ck);
        double v = f[0] + f[1]; v = v + f[W]; v = v + f[1 + W]; v = v + f[P]; v = v + f[1 + P]; v = v + f[W + P];
        v = v + f[1 + W + P];
        v = v * 0.125;
        const int qi = (o_i >> 1) + ci, qj = (o_j >> 1) + cj, qk = (o_k >> 1) + ck;
        const CellRef c = locate(GC, qi, qj, qk);
        vec_origin(C, c.box, R_id)[c.ijk] = v;
        if (!last) face_store(RW, A.Rc.down + (size_t)(j + 1) * kCellRecords + below_record(qi, qj, qk), v, epoch + SEQ_DOWN + (unsigned)(j + 1));
      }
      // zero_vector(C.e): whole padded boxes (misc.c:6-44), each workgroup a slice -- not when C is the next level of this chain (its visit clears what it does not store)
      if (last && A.below_zero) {
        const int vol = C.volume, total = C.num_boxes * vol, per = (total + nwg - 1) / nwg;
        const int lo = wg * per, hi = (lo + per < total) ? lo + per : total;
        for (int z = lo + t; z < hi; z += kThreads) {
          const int box = z / vol;
          (C.box_base[box] + (size_t)e_id * (size_t)vol)[z - box * vol] = 0.0;
        }
      }
      if (e_zero) {      // the part of zero_vector(this level's e) the stores above do not overwrite: every cell of the padded boxes that is not an interior cell
        const int vol = L.volume, total = L.num_boxes * vol, per = (total + nwg - 1) / nwg, jS = L.jStride, kS = L.kStride, g = L.ghosts, d = L.dim;
        const int lo = wg * per, hi = (lo + per < total) ? lo + per : total;
        for (int z = lo + t; z < hi; z += kThreads) {
          const int box = z / vol, off = z - box * vol, k = off / kS, r = off - k * kS, jj = r / jS, i = r - jj * jS;
          if (i >= g && i < g + d && jj >= g && jj < g + d && k >= g && k < g + d) continue;
          (L.box_base[box] + (size_t)e_id * (size_t)vol)[off] = 0.0;
        }
      }
    }
    __syncthreads();                                    // the next level of the chain reuses the LDS arrays
  }
  if (gave_up) brick_raise_error(A.Rc);
}

template <int V, int DIR, int B> constexpr size_t wide_lds_bytes() { return ((size_t)2 * WideGeom<V, B>::Cells + WideGeom<V, B>::BetaDoubles + (DIR == BW_UP ? WideGeom<V, B>::CoarseCells : 0)) * sizeof(double); }
template <int V, int SM, int DIR, int B>
static int wide_launch(const WideArgs &A) {
  static bool once = false;
  const size_t lds = wide_lds_bytes<V, DIR, B>();
  if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)brick_wide_kernel<V, SM, DIR, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
  hipLaunchKernelGGL((brick_wide_kernel<V, SM, DIR, B>), dim3(A.lv[0].nwg), dim3(WideGeom<V, B>::Threads), lds, g_stream, A);
  return 0;
}
// (bricks of 2^3: the 27-point operator only)
template <int V, int SM, int DIR>
static int wide_launch2(const WideArgs &A) {
  if constexpr (V == HPGMG_HIP_27PT_CC) return wide_launch<V, SM, DIR, 2>(A);
  else return record_error(hipErrorInvalidValue, "brick_wide_chain: a level of 2^3 cells of this operator");
}
template <int V, int SM>
static int wide_capacity() {
  static int cap = -1;
  if (cap < 0) {
    const int c0 = brick_workgroups_resident((const void *)brick_wide_kernel<V, SM, BW_DOWN, 8>, 256, wide_lds_bytes<V, BW_DOWN, 8>());
    const int c1 = brick_workgroups_resident((const void *)brick_wide_kernel<V, SM, BW_UP, 8>, 256, wide_lds_bytes<V, BW_UP, 8>());
    cap = c0 < c1 ? c0 : c1;
  }
  return cap;
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

// which bricks a level of dim^3 cells of this operator can be visited as: 8 (1 .. 8^3 bricks of 8^3 cells; a level of ONE brick has no neighbours: every halo cell
// is a boundary condition), 4 / 2 (a level of 4^3 / 2^3 cells: one brick of that size; 2^3: 27-point only), 0 (not at all).  Every brick lies inside one box; Dirichlet (the caller checks), every box here.
int hpgmg_hip_brick_wide_supported(const hpgmg_hip_level *L, int variant) {
  if (variant != HPGMG_HIP_27PT_CC && variant != HPGMG_HIP_FV4_VC_HELMHOLTZ && variant != HPGMG_HIP_FV4_VC_POISSON) return 0;
  const int r = (variant == HPGMG_HIP_27PT_CC) ? 1 : 2;
  if (!(L->dim_i == L->dim_j && L->dim_i == L->dim_k && L->dim > 0 && L->ghosts >= r && !L->periodic)) return 0;
  if (L->dim_i == 4 && L->dim == 4 && L->num_boxes == 1) return 4;
  if (L->dim_i == 2 && L->dim == 2 && L->num_boxes == 1 && variant == HPGMG_HIP_27PT_CC) return 2;      /* (the 4th-order operator's 2^3 level is its bottom level: the solver's) */
  const int side = L->dim_i / 8;
  return (L->dim_i % 8 == 0 && side >= 1 && side <= 8 && L->dim % 8 == 0) ? 8 : 0;
}
// workgroups of the (variant, smoother) kernels this device holds at once, less an eighth (hpgmg_hip_brick_chain_capacity)
int hpgmg_hip_brick_wide_capacity(int variant, int smoother) {
  int cap;
  switch (variant * 2 + smoother) {
    case HPGMG_HIP_27PT_CC * 2 + BW_CHEBY:           cap = wide_capacity<HPGMG_HIP_27PT_CC, BW_CHEBY>(); break;
    case HPGMG_HIP_27PT_CC * 2 + BW_GSRB:            cap = wide_capacity<HPGMG_HIP_27PT_CC, BW_GSRB>(); break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ * 2 + BW_CHEBY:  cap = wide_capacity<HPGMG_HIP_FV4_VC_HELMHOLTZ, BW_CHEBY>(); break;
    case HPGMG_HIP_FV4_VC_HELMHOLTZ * 2 + BW_GSRB:   cap = wide_capacity<HPGMG_HIP_FV4_VC_HELMHOLTZ, BW_GSRB>(); break;
    case HPGMG_HIP_FV4_VC_POISSON * 2 + BW_CHEBY:    cap = wide_capacity<HPGMG_HIP_FV4_VC_POISSON, BW_CHEBY>(); break;
    case HPGMG_HIP_FV4_VC_POISSON * 2 + BW_GSRB:     cap = wide_capacity<HPGMG_HIP_FV4_VC_POISSON, BW_GSRB>(); break;
    default: return 0;
  }
  return cap - cap / 8;
}

// (every level of one launch is cut into bricks of the same size: a chain of 8^3-brick levels, or the level of 4^3 cells on its own)
int hpgmg_hip_brick_wide_chain(int n, const hpgmg_hip_brick_level *levels, const hpgmg_hip_level *below, int sweeps, int variant, int smoother,
                               int e_id, int R_id, double a, double b, int dir, int top_e_zero, int below_zero) {
  if (g_skip_launches) return record_error(hipErrorInvalidValue, "brick_wide_chain: not replayable (the launch number is a kernel argument)");
  if (brick_error_pending()) return 0;      // (as hpgmg_hip_brick_chain: the host learns of the failed launch at its next scalar)
  if (n < 1 || n > kBrickMaxLevels || sweeps < 1 || sweeps > kBrickMaxSweeps || (sweeps & 1) || dir < 0 || dir > 1 || smoother < 0 || smoother > 1)
    return record_error(hipErrorInvalidValue, "brick_wide_chain: levels / sweeps / direction / smoother");
  const int brick = hpgmg_hip_brick_wide_supported(&levels[0].L, variant);
  if (!brick || (brick < 8 && n != 1)) return record_error(hipErrorInvalidValue, "brick_wide_chain: level");
  for (int j = 0; j < n; j++) {
    const hpgmg_hip_level *next = (j + 1 < n) ? &levels[j + 1].L : below;
    if (hpgmg_hip_brick_wide_supported(&levels[j].L, variant) != brick || 2 * next->dim_i != levels[j].L.dim_i) return record_error(hipErrorInvalidValue, "brick_wide_chain: level");
  }
  WideArgs A = {};
  for (int j = 0; j < n; j++) {
    A.lv[j].L = levels[j].L; A.lv[j].h2inv = levels[j].h2inv;
    for (int s = 0; s < sweeps; s++) { A.lv[j].c1[s] = levels[j].c1[s]; A.lv[j].c2[s] = levels[j].c2[s]; }
    A.lv[j].side = levels[j].L.dim_i / brick; A.lv[j].nwg = A.lv[j].side * A.lv[j].side * A.lv[j].side;
  }
  A.C = *below; A.n = n; A.a = a; A.b = b; A.sweeps = sweeps; A.e_id = e_id; A.R_id = R_id;
  A.top_e_zero = (dir == 0 && top_e_zero) ? 1 : 0; A.below_zero = below_zero ? 1 : 0;
  { const int rc = brick_records_for_launch(&A.Rc); if (rc) return rc; }
  A.absent_wg = brick_test_absent_wg();
  int rc;
#define WIDE_CASE(VV, SS) case VV * 2 + SS: \
    if (brick == 8)      rc = (dir == 0) ? wide_launch<VV, SS, BW_DOWN, 8>(A) : wide_launch<VV, SS, BW_UP, 8>(A); \
    else if (brick == 4) rc = (dir == 0) ? wide_launch<VV, SS, BW_DOWN, 4>(A) : wide_launch<VV, SS, BW_UP, 4>(A); \
    else                 rc = (dir == 0) ? wide_launch2<VV, SS, BW_DOWN>(A) : wide_launch2<VV, SS, BW_UP>(A); \
    break
  switch (variant * 2 + smoother) {
    WIDE_CASE(HPGMG_HIP_27PT_CC, BW_CHEBY);
    WIDE_CASE(HPGMG_HIP_27PT_CC, BW_GSRB);
    WIDE_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ, BW_CHEBY);
    WIDE_CASE(HPGMG_HIP_FV4_VC_HELMHOLTZ, BW_GSRB);
    WIDE_CASE(HPGMG_HIP_FV4_VC_POISSON, BW_CHEBY);
    WIDE_CASE(HPGMG_HIP_FV4_VC_POISSON, BW_GSRB);
    default: return record_error(hipErrorInvalidValue, "brick_wide_chain: variant / smoother");
  }
#undef WIDE_CASE
  if (rc) return rc;
  HPGMG_LAUNCH_CHECK("brick_wide_kernel");
  brick_count_visits(n);
  return 0;
}

}  // extern "C"

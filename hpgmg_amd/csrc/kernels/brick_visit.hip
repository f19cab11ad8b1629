// brick_visit.hip -- the visits of the LAUNCH-BOUND levels of MGVCycle (mg.c:1147-1163: 64^3, 32^3, 16^3 cells) as ONE launch per V-cycle leg.
// A level is cut into bricks of 8^3 cells, one workgroup of 512 lanes per brick (64^3: 512 workgroups, 32^3: 64, 16^3: 8; level j of a chain is worked on
// by workgroups 0 .. n_j - 1), the brick's iterate and VECTOR_TEMP in LDS with a one-cell halo, its coefficients in registers -- tail.hip's scheme on more
// than one CU.  (Bricks of 16^3 cells, 1024 lanes, four cells per lane also exist: a sweep of 4096 cells is ~3 us of fp64 issue on ONE CU, so they lose.)
//   DOWN   per level:  smooth; residual -> TEMP; restriction(next.R <- TEMP); zero_vector(next.e)
//   UP     per level, coarsest first:  interpolation_vcycle (e += P next.e, piecewise constant); smooth
//   FDOWN  (one level) interpolation_fcycle (e = 0.0 e + P1 next.e, piecewise linear, the coarse ghost cells of apply_BCs_p1 formed on the fly), then DOWN --
//          the step of FMGSolve (mg.c:1289-1293) that opens a V-cycle: its exchange + boundary + interpolation launches ride in the load of the visit
// i.e. per level visit the 5 + 4 launches of ~5 us each that the per-operator path issues (4 sweeps of 1.3 MB each are not what they cost: a launch
// boundary and one memory round trip per sweep are).
//
// Everything that crosses a workgroup boundary INSIDE the launch -- faces between sweeps, the restricted residuals a level hands to the one below it, the
// corrections it hands to the one above -- goes THROUGH MEMORY WITHOUT LEAVING THE KERNEL.  The XCDs' L2 caches are not coherent with each other inside a
// kernel, so a cell travels as a 16-byte record of two 8-byte words {tag, value low} {value high, tag} (brick_records.hpp), each written through with one
// atomic store by the lane that owns the cell and polled with atomic loads by the lane that needs it, accepted when BOTH words carry the expected tag: the
// record is its own flag -- one memory hop, no counter everybody adds to, no fence.  tools/microbench/p2p_flags.hip: 1.6 us per
// exchange for 8 .. 64 workgroups, 2.4 for 256, against 3.1-3.7 us for a kernel boundary around the same traffic, 2.4-8.3 for data + flag, 5.7-76 for a
// central counter (what grid.sync() is).  Face records are double-buffered by exchange parity: a brick can publish exchange n only after it has read all
// its neighbours' exchange n-1, which they published after reading n-2 -- the slot of parity n is free.  Tags never repeat (launch number x 64 + a code
// for the record's role; the areas are cleared before the 32-bit number wraps), so records of earlier launches never match.  A gate record per brick ends the LONG wait of the finer level's workgroups
// on the way up: one lane watches it, the others poll their own records only afterwards.
// Every other global access is of the ordinary kind and obeys one rule: within a launch an address is written by ONE workgroup only and never read by
// another (the halo of a level's first sweep comes from what EARLIER launches stored).  Hence zero_vector of a level below the first of a chain is done
// by that level's own visit (it then does not read the vector at all), and FDOWN -- every brick of which reads the level below -- is a launch of its own.
// All workgroups of the launch must be resident at once: at least six waves per SIMD are forced (80 registers), so three workgroups fit a CU -- 768 slots
// on 256 CUs for the 512 bricks of a 64^3 level.  The host ASKS (hpgmg_hip_brick_chain_capacity: occupancy x CUs of this device, less an eighth) and takes the
// launch-by-launch path when a level has more bricks than that (a partitioned or smaller device).  What no query can see -- other processes' launches of this
// kind filling the slots -- ends a poll after 2 s: it raises the error words, every launch behind it gives up within 100 us, the host learns of it at its next
// scalar and redoes the solve launch by launch (host/mg.c FMGSolve) or stops with a message (the reference's own driver, multi-rank jobs): never a hung GPU.
// Arithmetic: the expression trees of tail.hip / the streaming kernels (stencil_math.hpp, chebyshev.c:86-95, gsrb.c:100-104, jacobi.c:50-56,
// residual.c:42-48, restriction.c:54-57, interpolation_p0.c:43, interpolation_p1.c:40-70): bit-identical to the per-operator path;
// tests/test_gpu_operators.py runs both, tools/stress_bricks.py repeats a cycle hundreds of times and compares the bytes.
#include "common.hpp"
#include "stencil_math.hpp"
#include "dense_levels.hpp"
#include "brick_records.hpp"

namespace hpgmg {

// brick geometry: B^3 cells per workgroup.  B = 16: 1024 lanes, 4 cells each; B = 8: 512 lanes, one cell each (eight times the workgroups: a sweep of a
// 16^3 brick is ~3 us of fp64 issue on ONE CU, which is most of what a launch boundary costs)
template <int B_> struct BrickGeom {
  static constexpr int B = B_, Cells = B * B * B, Threads = (B == 16) ? 1024 : 512, PerLane = Cells / Threads;
  static constexpr int W = B + 2, Plane = W * W, Halo = W * W * W;            // LDS: (B + 2)^3 doubles per array
  static constexpr int Face = B * B, MaxSide = (B == 16) ? 4 : 8;
  static constexpr int StepK = Threads / Face, StepPos = StepK * Plane;       // from a lane's cell m to its cell m + 1
  __device__ __forceinline__ static constexpr int hpos(int li, int lj, int lk) { return (li + 1) + W * (lj + 1) + Plane * (lk + 1); }   // li, lj, lk in -1 .. B
  // face f (0,1: -i,+i; 2,3: -j,+j; 4,5: -k,+k), in-face cell (u, v): the brick cell on the face (depth 0) or the halo cell beyond it (depth 1)
  __device__ __forceinline__ static void face_cell(int f, int u, int v, int depth, int &li, int &lj, int &lk) {
    const int w = (f & 1) ? (B - 1 + depth) : -depth;
    if (f < 2) { li = w; lj = u; lk = v; } else if (f < 4) { li = u; lj = w; lk = v; } else { li = u; lj = v; lk = w; }
  }
};
struct BrickLevel {
  hpgmg_hip_level L;
  double h2inv, c1[kBrickMaxSweeps], c2[kBrickMaxSweeps];
  int side, nwg;                    // bricks per dimension, per level
};
struct BrickArgs {
  BrickLevel lv[kBrickMaxLevels];   // the levels of the chain, finest first: level j is worked on by workgroups 0 .. lv[j].nwg - 1
  hpgmg_hip_level C;                // the level below the last one (restriction target / interpolation source): other launches' business
  int n;
  double a, b;
  int sweeps, e_id, R_id;
  int top_e_zero;                   // DOWN: the correction of lv[0] counts as +0.0 (zero_vector came before, mg.c:1153) and is not read; the cells of its padded
                                    // boxes that no brick stores (ghost zone, padding) are cleared here.  (Levels below the first: always.)
  int below_zero;                   // DOWN: zero_vector(C.e) at the end (0: somebody else's)
  BrickRecords Rc;                  // record areas, launch epoch, error words (brick_records.hpp)
  int absent_wg;                    // tests (HPGMG_TEST_BRICK_ABSENT): this workgroup leaves at once, as if it had never been given a CU; -1: none
};

template <int V, int kHaloW, int kHaloPlane>
__device__ __forceinline__ double brick_apply(const double *src, int p, int gi, int gj, int gk, int D, const CellCoef<V> &q, double a, double b, double h2inv) {
  const double xc = src[p];
  const int last = D - 1;
  const double xim = (gi == 0)    ? -xc : src[p - 1];
  const double xip = (gi == last) ? -xc : src[p + 1];
  const double xjm = (gj == 0)    ? -xc : src[p - kHaloW];
  const double xjp = (gj == last) ? -xc : src[p + kHaloW];
  const double xkm = (gk == 0)    ? -xc : src[p - kHaloPlane];
  const double xkp = (gk == last) ? -xc : src[p + kHaloPlane];
  return apply_op_7pt<V>(xc, xim, xip, xjm, xjp, xkm, xkp, q.bi0, q.bi1, q.bj0, q.bj1, q.bk0, q.bk1, q.al, a, b, h2inv);
}

enum { BV_CHEBY = 0, BV_GSRB = 1, BV_JACOBI = 2 };
enum { DIR_DOWN = 0, DIR_UP = 1, DIR_FDOWN = 2 };      // FDOWN: interpolation_fcycle, then DOWN (one level)

// One launch = the visits of a CHAIN of launch-bound levels in one direction.  DOWN: level 0, then (workgroups < lv[1].nwg) level 1 with the residuals
// level 0 restricted, ...; UP: the last level first, every finer level waiting for the corrections of the one below it.  What passes between two levels
// of the chain passes as records, like the faces (a level's bricks are not the workgroups that held the cells above / below them).
template <int V, int SM, int DIR, int B>
// (B = 8: at least six waves per SIMD = three workgroups per CU = room for 768 -- a level of 64^3 cells needs 512 of them running at once, and not on the last free slot)
__global__ __launch_bounds__(BrickGeom<B>::Threads, (B == 8) ? 6 : 4) void brick_chain_kernel(const BrickArgs A) {
  using BG = BrickGeom<B>;
  constexpr int kBrick = B, kBrickCells = BG::Cells, kBrickThreads = BG::Threads, kBrickPerLane = BG::PerLane, kHaloW = BG::W, kHaloPlane = BG::Plane, kHaloCells = BG::Halo;
  constexpr int kFaceCells = BG::Face, kStepPos = BG::StepPos, kStepK = BG::StepK;
  auto hpos = [](int li, int lj, int lk) { return BG::hpos(li, lj, lk); };
  auto face_cell = [](int f, int u, int v, int depth, int &li, int &lj, int &lk) { BG::face_cell(f, u, v, depth, li, lj, lk); };
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  constexpr bool kUp = (DIR == DIR_UP), kDown = (DIR != DIR_UP), kFInterp = (DIR == DIR_FDOWN);
  extern __shared__ double brick_lds[];
  double *const sx = brick_lds, *const st = brick_lds + kHaloCells;
  constexpr int kCW = kBrick / 2 + 2;                 // interpolation_fcycle: the brick's coarse cells with a ring of one, (B/2 + 2)^3 doubles behind the two arrays
  double *const sc = brick_lds + 2 * kHaloCells;
  const int t = (int)threadIdx.x, wg = (int)blockIdx.x, e_id = A.e_id, R_id = A.R_id, n = A.n;
  const unsigned epoch = A.Rc.epoch;
  const unsigned *const err_dev = A.Rc.error_dev;
  const RecordWindow RW = record_window(A.Rc);      // the record areas as a buffer (brick_records.hpp)
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool gave_up = false;
  if (wg == A.absent_wg) return;
  const int li0 = t % kBrick, lj0 = (t / kBrick) % kBrick, lk0 = t / kFaceCells, pos0 = hpos(li0, lj0, lk0);

  for (int step = 0; step < n; step++) {
    const int j = kUp ? n - 1 - step : step;
    const BrickLevel &T = A.lv[j];
    const int nwg = T.nwg;
    if (wg >= nwg) { if (kUp) continue; else break; }
    const bool first = (j == 0), last = (j == n - 1);
    const hpgmg_hip_level &L = T.L;
    const hpgmg_hip_level &C = last ? A.C : A.lv[last ? j : j + 1].L;      // the level below this one
    const int side = T.side, bx = wg % side, by = (wg / side) % side, bz = wg / (side * side);
    const int D = L.dim_i, o_i = bx * kBrick, o_j = by * kBrick, o_k = bz * kBrick;
    const LevelGeom G = geom_of(L), GC = geom_of(C);
    const int gi = o_i + li0, gj = o_j + lj0, gk0 = o_k + lk0;
    const double a = A.a, b = A.b, h2inv = T.h2inv;
    FaceCell *const faces = A.Rc.faces + (size_t)j * kFaceRecords;
    const bool e_zero = kDown && !kFInterp && (first ? (A.top_e_zero != 0) : true);
    const bool rhs_by_record = kDown && !first, parent_by_record = kUp && !last;
    // the record of cell (ci, cj, ck) of the level BELOW this one (where its owner publishes / expects it): bricks of that level are numbered like ours
    const int side_c = last ? 1 : A.lv[last ? j : j + 1].side;
    auto below_record = [&](int ci, int cj, int ck) -> size_t {
      const int qx = ci / kBrick, qy = cj / kBrick, qz = ck / kBrick;
      const int owner = qx + side_c * (qy + side_c * qz);
      return (size_t)owner * kBrickCells + (size_t)((ci - qx * kBrick) + kBrick * ((cj - qy * kBrick) + kBrick * (ck - qz * kBrick)));
    };
    const FaceCell *const up_from_below = A.Rc.up + (size_t)(last ? j : j + 1) * kCellRecords;      // corrections of the level below (UP, not the last level)
    const unsigned seq_parent = epoch + SEQ_UP + (unsigned)(j + 1);

    // what the first sweep reads at global cell (ci, cj, ck) of this level, given what is stored there
    auto start_value = [&](double stored, int ci, int cj, int ck) -> double {
      if (kUp) {               // interpolation_vcycle: e = 1.0*e + (coarse parent), interpolation_p0.c:43
        double parent;
        if (parent_by_record) parent = record_wait(RW, up_from_below + below_record(ci >> 1, cj >> 1, ck >> 1), seq_parent, t0, gave_up, err_dev);
        else { const CellRef c = locate(GC, ci >> 1, cj >> 1, ck >> 1); parent = vec_origin(C, c.box, e_id)[c.ijk]; }
        return 1.0 * stored + parent;
      }
      if (kFInterp) {          // interpolation_fcycle, piecewise linear (interpolation_p1.c:40-70): f = 0.0 f + 27/64 c + 9/64 (3 face neighbours) + 3/64 (3 edge
        // neighbours) + 1/64 corner, an even fine cell leaning on the coarse neighbour behind it, an odd one on the one ahead; the coarse values wait in LDS (sc)
        const int qi = (ci >> 1) - (o_i >> 1) + 1, qj = (cj >> 1) - (o_j >> 1) + 1, qk = (ck >> 1) - (o_k >> 1) + 1;      // position in sc
        const int di = (ci & 1) ? 1 : -1, dj = (cj & 1) ? kCW : -kCW, dk = (ck & 1) ? kCW * kCW : -kCW * kCW;
        const double *c0 = sc + qi + kCW * (qj + kCW * qk);
        double v = 0.0 * stored;
        v = v + 0.421875 * c0[0];
        v = v + 0.140625 * c0[dk];
        v = v + 0.140625 * c0[dj];
        v = v + 0.046875 * c0[dj + dk];
        v = v + 0.140625 * c0[di];
        v = v + 0.046875 * c0[di + dk];
        v = v + 0.046875 * c0[di + dj];
        v = v + 0.015625 * c0[di + dj + dk];
        return v;
      }
      return stored;
    };

    // ---- the brick: what is stored of the iterate, VECTOR_TEMP, coefficients (every load is in flight before anything is waited for)
    CellCoef<V> q[kBrickPerLane];
    double e_st[kBrickPerLane];
    auto load_coefficients = [&]() {
#pragma unroll
      for (int m = 0; m < kBrickPerLane; m++) {
        const CellRef w = locate(G, gi, gj, gk0 + m * kStepK);
        const int box = w.box, ijk = w.ijk, jS = L.jStride, kS = L.kStride;
        q[m].rhs = rhs_by_record ? 0.0 : vec_origin(L, box, R_id)[ijk];
        q[m].dinv = vec_origin(L, box, VECTOR_DINV)[ijk];
        q[m].bi0 = q[m].bi1 = q[m].bj0 = q[m].bj1 = q[m].bk0 = q[m].bk1 = q[m].al = 0.0;
        if (kVC) {
          const double *bi = vec_origin(L, box, VECTOR_BETA_I), *bj = vec_origin(L, box, VECTOR_BETA_J), *bk = vec_origin(L, box, VECTOR_BETA_K);
          q[m].bi0 = bi[ijk]; q[m].bi1 = bi[ijk + 1]; q[m].bj0 = bj[ijk]; q[m].bj1 = bj[ijk + jS]; q[m].bk0 = bk[ijk]; q[m].bk1 = bk[ijk + kS];
        }
        if (kHelm) q[m].al = vec_origin(L, box, VECTOR_ALPHA)[ijk];
      }
    };
#pragma unroll
    for (int m = 0; m < kBrickPerLane; m++) {
      const int p = pos0 + m * kStepPos;
      const CellRef w = locate(G, gi, gj, gk0 + m * kStepK);
      e_st[m] = e_zero ? 0.0 : vec_origin(L, w.box, e_id)[w.ijk];
      st[p] = vec_origin(L, w.box, VECTOR_TEMP)[w.ijk];
    }
    if (kFInterp) {
      // the coarse cells this brick's interpolation reads: its (B/2)^3 parents and a ring of one.  A coarse ghost cell is -, +, - its mirror image for 1, 2, 3
      // directions leaving the domain (exchange_boundary + apply_BCs_p1, BOX shape, boundary_fd.c:35-38), formed here
      const int Dc = C.dim_i;
      for (int idx = t; idx < kCW * kCW * kCW; idx += kBrickThreads) {
        int qi = (o_i >> 1) - 1 + idx % kCW, qj = (o_j >> 1) - 1 + (idx / kCW) % kCW, qk = (o_k >> 1) - 1 + idx / (kCW * kCW);
        double sg = 1.0;
        if (qi < 0) { qi = 0; sg = -sg; } else if (qi >= Dc) { qi = Dc - 1; sg = -sg; }
        if (qj < 0) { qj = 0; sg = -sg; } else if (qj >= Dc) { qj = Dc - 1; sg = -sg; }
        if (qk < 0) { qk = 0; sg = -sg; } else if (qk >= Dc) { qk = Dc - 1; sg = -sg; }
        const CellRef r = locate(GC, qi, qj, qk);
        sc[idx] = sg * vec_origin(C, r.box, e_id)[r.ijk];
      }
      __syncthreads();
    }
    if (!kFInterp) load_coefficients();      // (interpolation_fcycle: after the start values -- its eight coarse loads per cell and the coefficients do not fit the registers together)
    // the stored iterate beyond the faces (the neighbouring bricks' cells as EARLIER launches left them); lane roles for everything on faces:
    // cell fc = (f B + v) B + u of the 6 B^2 face cells
    constexpr int kHaloPerLane = (6 * kFaceCells + kBrickThreads - 1) / kBrickThreads;
    double h_st[kHaloPerLane];
#pragma unroll
    for (int hm = 0; hm < kHaloPerLane; hm++) {
      const int fc = t + hm * kBrickThreads;
      h_st[hm] = 0.0;
      if (fc >= 6 * kFaceCells || e_zero) continue;
      const int f = fc / kFaceCells, u = fc % kBrick, v = (fc / kBrick) % kBrick;
      int li, lj, lk;
      face_cell(f, u, v, 1, li, lj, lk);
      const int hi = o_i + li, hj = o_j + lj, hk = o_k + lk;
      if (hi < 0 || hi >= D || hj < 0 || hj >= D || hk < 0 || hk >= D) continue;      // beyond the domain: the Dirichlet rule, never read
      const CellRef r = locate(G, hi, hj, hk);
      h_st[hm] = vec_origin(L, r.box, e_id)[r.ijk];
    }
    // UP, not the last level: the corrections of the level below are on their way.  One lane watches the gate of the brick that holds this brick's parents
    // (hundreds of workgroups polling a record per lane for the length of a visit would be in the way of the bricks that work)
    if (parent_by_record) {
      if (t == 0) (void)record_wait(RW, A.Rc.gate + (size_t)(j + 1) * kBrickMaxWgs + ((bx >> 1) + side_c * ((by >> 1) + side_c * (bz >> 1))), epoch + SEQ_GATE + (unsigned)(j + 1), t0, gave_up, err_dev, 8);
      __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < kBrickPerLane; m++) {
      const int p = pos0 + m * kStepPos, gk = gk0 + m * kStepK;
      sx[p] = start_value(e_st[m], gi, gj, gk);
      if (rhs_by_record) q[m].rhs = record_wait(RW, A.Rc.down + (size_t)j * kCellRecords + (size_t)wg * kBrickCells + (size_t)(t + m * kBrickThreads), epoch + SEQ_DOWN + (unsigned)j, t0, gave_up, err_dev);
    }
    if (kFInterp) __builtin_amdgcn_sched_barrier(0);      // (the eight coarse loads of a cell's interpolation, twice over, are more than the register budget holds at once)
#pragma unroll
    for (int hm = 0; hm < kHaloPerLane; hm++) {
      const int fc = t + hm * kBrickThreads;
      if (fc >= 6 * kFaceCells) continue;
      const int f = fc / kFaceCells, u = fc % kBrick, v = (fc / kBrick) % kBrick;
      int li, lj, lk;
      face_cell(f, u, v, 1, li, lj, lk);
      const int hi = o_i + li, hj = o_j + lj, hk = o_k + lk;
      if (hi < 0 || hi >= D || hj < 0 || hj >= D || hk < 0 || hk >= D) continue;
      sx[hpos(li, lj, lk)] = e_zero ? 0.0 : start_value(h_st[hm], hi, hj, hk);
    }
    if (kFInterp) { __builtin_amdgcn_sched_barrier(0); load_coefficients(); }
    __syncthreads();

    // ---- smooth(): chebyshev.c:43-99 / gsrb.c:24-132 / jacobi.c:17-62 (an even number of sweeps: the result ends in sx)
    int exchange_n = 0;
    for (int s = 0; s < A.sweeps; s++) {
      const double *src = (SM != BV_GSRB && (s & 1)) ? st : sx;
      double *dst = (SM == BV_GSRB) ? sx : ((s & 1) ? sx : st);
      const double c1 = T.c1[s], c2 = T.c2[s];
#pragma unroll
      for (int m = 0; m < kBrickPerLane; m++) {
        const int p = pos0 + m * kStepPos, gk = gk0 + m * kStepK;
        if (SM == BV_GSRB && ((gi ^ gj ^ gk ^ s) & 1) != 0) continue;      // global parity: box.low folded in (gsrb.c:55); red cells read black neighbours only
        const double xc = src[p];
        const double Ax = brick_apply<V, kHaloW, kHaloPlane>(src, p, gi, gj, gk, D, q[m], a, b, h2inv);
        if (SM == BV_CHEBY)     { const double xnm1 = dst[p]; dst[p] = xc + c1 * (xc - xnm1) + c2 * q[m].dinv * (q[m].rhs - Ax); }
        else if (SM == BV_GSRB) { dst[p] = xc + q[m].dinv * (q[m].rhs - Ax); }
        else                    { dst[p] = xc + (2.0 / 3.0) * q[m].dinv * (q[m].rhs - Ax); }
      }
      __syncthreads();
      if (kDown || s + 1 < A.sweeps) {      // (the way down goes on to the residual of the result)
        // one exchange: the faces of dst go out, the neighbours' faces come into its halo
        const int par = exchange_n & 1;
        const unsigned seq = epoch + SEQ_FACES + (unsigned)(12 * j + exchange_n);
        exchange_n++;
        FaceCell *mine = faces + ((size_t)par * nwg + wg) * 6 * kFaceCells;
        for (int fc = t; fc < 6 * kFaceCells; fc += kBrickThreads) {
          const int f = fc / kFaceCells, u = fc % kBrick, v = (fc / kBrick) % kBrick;
          const int bc = (f < 2) ? bx : ((f < 4) ? by : bz);
          if ((f & 1) ? (bc == side - 1) : (bc == 0)) continue;                          // no brick beyond this face
          int li, lj, lk;
          face_cell(f, u, v, 0, li, lj, lk);
          face_store(RW, mine + fc, dst[hpos(li, lj, lk)], seq);
        }
        for (int fc = t; fc < 6 * kFaceCells; fc += kBrickThreads) {
          const int f = fc / kFaceCells, u = fc % kBrick, v = (fc / kBrick) % kBrick;
          const int bc = (f < 2) ? bx : ((f < 4) ? by : bz);
          if ((f & 1) ? (bc == side - 1) : (bc == 0)) continue;
          const int stp = (f < 2) ? 1 : ((f < 4) ? side : side * side), nb = wg + ((f & 1) ? stp : -stp);
          const double xv = record_wait(RW, faces + (((size_t)par * nwg + nb) * 6 + (f ^ 1)) * kFaceCells + (fc % kFaceCells), seq, t0, gave_up, err_dev);
          int li, lj, lk;
          face_cell(f, u, v, 1, li, lj, lk);
          dst[hpos(li, lj, lk)] = xv;
        }
        __syncthreads();
      }
    }

    if (kDown) {                                        // residual -> TEMP (residual.c:42-48)
#pragma unroll
      for (int m = 0; m < kBrickPerLane; m++) {
        const int p = pos0 + m * kStepPos;
        const double Ax = brick_apply<V, kHaloW, kHaloPlane>(sx, p, gi, gj, gk0 + m * kStepK, D, q[m], a, b, h2inv);
        st[p] = q[m].rhs - Ax;                          // each lane overwrites only its own TEMP cells: no hazard with the reads of sx
      }
      __syncthreads();
    }

    // ---- leave e and TEMP in global memory as the per-operator sequence would; UP: hand the correction to the finer level of the chain
#pragma unroll
    for (int m = 0; m < kBrickPerLane; m++) {
      const int p = pos0 + m * kStepPos;
      const CellRef w = locate(G, gi, gj, gk0 + m * kStepK);
      vec_origin(L, w.box, e_id)[w.ijk] = sx[p];
      vec_origin(L, w.box, VECTOR_TEMP)[w.ijk] = st[p];
      if (kUp && !first) face_store(RW, A.Rc.up + (size_t)j * kCellRecords + (size_t)wg * kBrickCells + (size_t)(t + m * kBrickThreads), sx[p], epoch + SEQ_UP + (unsigned)j);
    }
    if (kUp && !first && t == 0) face_store(RW, A.Rc.gate + (size_t)j * kBrickMaxWgs + wg, 0.0, epoch + SEQ_GATE + (unsigned)j);      // (issued after lane 0's records; the others' may still be on their way: the gate only ends the long wait)

    if (kDown) {
      // restriction(coarse.R <- TEMP): 0.125 * sum of the 8 children in the reference's order (restriction.c:54-57); this brick's (B/2)^3 coarse cells
      if (t < kBrickCells / 8) {
        constexpr int H = kBrick / 2;
        const int ci = t % H, cj = (t / H) % H, ck = t / (H * H);
        const double *f = st + hpos(2 * ci, 2 * cj, 2 * ck);
        double v = f[0] + f[1]; v = v + f[kHaloW]; v = v + f[1 + kHaloW]; v = v + f[kHaloPlane]; v = v + f[1 + kHaloPlane]; v = v + f[kHaloW + kHaloPlane];
        v = v + f[1 + kHaloW + kHaloPlane];
        v = v * 0.125;
        const int qi = (o_i >> 1) + ci, qj = (o_j >> 1) + cj, qk = (o_k >> 1) + ck;
        const CellRef c = locate(GC, qi, qj, qk);
        vec_origin(C, c.box, R_id)[c.ijk] = v;
        if (!last) face_store(RW, A.Rc.down + (size_t)(j + 1) * kCellRecords + below_record(qi, qj, qk), v, epoch + SEQ_DOWN + (unsigned)(j + 1));
      }
      // zero_vector(C.e): the whole padded boxes, ghosts included (misc.c:6-44), each workgroup a slice of the flat range.  Not when C is the next level of
      // this chain (its visit does not read the vector and clears what it does not store itself, below)
      if (last && A.below_zero) {
        const int vol = C.volume, total = C.num_boxes * vol, per = (total + nwg - 1) / nwg;
        const int lo = wg * per, hi = (lo + per < total) ? lo + per : total;
        for (int z = lo + t; z < hi; z += kBrickThreads) {
          const int box = z / vol;
          (C.box_base[box] + (size_t)e_id * (size_t)vol)[z - box * vol] = 0.0;
        }
      }
      if (e_zero) {
        // the part of zero_vector(this level's e) the stores above do not overwrite: every cell of the padded boxes that is not an interior cell
        const int vol = L.volume, total = L.num_boxes * vol, per = (total + nwg - 1) / nwg, jS = L.jStride, kS = L.kStride, g = L.ghosts, d = L.dim;
        const int lo = wg * per, hi = (lo + per < total) ? lo + per : total;
        for (int z = lo + t; z < hi; z += kBrickThreads) {
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

static FaceCell *g_records = nullptr;         // faces | down | up | gate, kBrickMaxLevels of each, then the device copy of the error word
static unsigned *g_error = nullptr;          // pinned host word
static unsigned *g_error_dev = nullptr;
static unsigned g_epoch = 0;
static long long g_visits = 0;

int brick_records_for_launch(BrickRecords *R) {
  if (!g_records) {
    HPGMG_CHECK(hipMalloc((void **)&g_records, (kRecordsTotal + 4) * sizeof(FaceCell)));
    HPGMG_CHECK(hipMemset(g_records, 0, (kRecordsTotal + 4) * sizeof(FaceCell)));
    HPGMG_CHECK(hipHostMalloc((void **)&g_error, 64, hipHostMallocDefault));
    *g_error = 0;
    g_error_dev = (unsigned *)(g_records + kRecordsTotal);
    HPGMG_CHECK(hipDeviceSynchronize());
  }
  if (g_epoch > 0xFFFFFFFFu - 256u) {            // the 32-bit tags would repeat: clear every record first (in stream order, behind the launches that use them)
    HPGMG_CHECK(hipMemsetAsync(g_records, 0, kRecordsTotal * sizeof(FaceCell), g_stream));
    g_epoch = 0;
  }
  g_epoch += 64;
  R->faces = g_records; R->down = R->faces + (size_t)kBrickMaxLevels * kFaceRecords; R->up = R->down + (size_t)kBrickMaxLevels * kCellRecords;
  R->gate = R->up + (size_t)kBrickMaxLevels * kCellRecords;
  R->epoch = g_epoch; R->error = g_error; R->error_dev = g_error_dev;
  return 0;
}
// Every workgroup of a brick launch waits for records of others: all of them must be RESIDENT at once.  What the device holds of a kernel = its occupancy
// per CU (registers, LDS, waves) x the CUs of this device (a partition in CPX / DPX mode reports its own count).
int brick_workgroups_resident(const void *kernel, int threads, size_t lds_bytes) {
  static const int forced = [] { const char *e = getenv("HPGMG_TEST_BRICK_CAPACITY"); return (e && *e) ? atoi(e) : -1; }();
  if (forced >= 0) return forced;
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return per_cu * cus;
}
int brick_test_absent_wg(void) {
  static const int absent = [] { const char *e = getenv("HPGMG_TEST_BRICK_ABSENT"); return (e && *e) ? atoi(e) : -1; }();
  return absent;
}
void brick_count_visits(int n) { g_visits += n; }
bool brick_error_pending(void) { return g_error && *(volatile unsigned *)g_error; }

template <int DIR, int B> constexpr size_t brick_lds_bytes() { return ((size_t)2 * BrickGeom<B>::Halo + (DIR == DIR_FDOWN ? (size_t)(B / 2 + 2) * (B / 2 + 2) * (B / 2 + 2) : 0)) * sizeof(double); }
template <int V, int SM, int DIR, int B>
static int brick_launch(const BrickArgs &A) {
  static bool once = false;
  const size_t lds = brick_lds_bytes<DIR, B>();
  if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)brick_chain_kernel<V, SM, DIR, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
  hipLaunchKernelGGL((brick_chain_kernel<V, SM, DIR, B>), dim3(A.lv[0].nwg), dim3(BrickGeom<B>::Threads), lds, g_stream, A);
  return 0;
}
// workgroups of the (variant, smoother) kernels the device holds at once: the smallest figure of the three directions
template <int V, int SM, int B>
static int brick_capacity() {
  static int cap = -1;
  if (cap < 0) {
    const int c0 = brick_workgroups_resident((const void *)brick_chain_kernel<V, SM, 0, B>, BrickGeom<B>::Threads, brick_lds_bytes<0, B>());
    const int c1 = brick_workgroups_resident((const void *)brick_chain_kernel<V, SM, 1, B>, BrickGeom<B>::Threads, brick_lds_bytes<1, B>());
    const int c2 = brick_workgroups_resident((const void *)brick_chain_kernel<V, SM, 2, B>, BrickGeom<B>::Threads, brick_lds_bytes<2, B>());
    cap = c0 < c1 ? (c0 < c2 ? c0 : c2) : (c1 < c2 ? c1 : c2);
  }
  return cap;
}
template <int V, int SM> static int brick_capacity_of(int brick) { return brick == 16 ? brick_capacity<V, SM, 16>() : brick_capacity<V, SM, 8>(); }
template <int V, int SM>
static int brick_launch_dir(const BrickArgs &A, int dir, int brick) {
  if (brick == 16) return dir == 0 ? brick_launch<V, SM, 0, 16>(A) : (dir == 1 ? brick_launch<V, SM, 1, 16>(A) : brick_launch<V, SM, 2, 16>(A));
  return dir == 0 ? brick_launch<V, SM, 0, 8>(A) : (dir == 1 ? brick_launch<V, SM, 1, 8>(A) : brick_launch<V, SM, 2, 8>(A));
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

int hpgmg_hip_brick_visit_max_sweeps(void) { return kBrickMaxSweeps; }
int hpgmg_hip_brick_chain_max_levels(void) { return kBrickMaxLevels; }
long long hpgmg_hip_brick_visits(void) { return g_visits; }      // level visits so far (tests)
// 1: a level of dim^3 cells can be visited as bricks of brick^3 cells (brick = 16: 2^3 or 4^3 of them; brick = 8: 2^3 .. 8^3)
int hpgmg_hip_brick_visit_supported(const hpgmg_hip_level *L, int brick) {
  if (brick != 8 && brick != 16) return 0;
  const int side = L->dim_i / brick, max_side = (brick == 16) ? BrickGeom<16>::MaxSide : BrickGeom<8>::MaxSide;
  return L->dim_i == L->dim_j && L->dim_i == L->dim_k && L->dim_i % brick == 0 && side >= 2 && side <= max_side && L->dim > 0 && L->dim_i % L->dim == 0;
}
// 0: fine; 1: a poll of an earlier launch gave up (the results since then are not to be used)
int hpgmg_hip_brick_visit_error(void) { return brick_error_pending() ? 1 : 0; }
// after the host has dealt with it (every launch behind the failed one has ended: the caller synchronised to learn of it)
int hpgmg_hip_brick_visit_error_clear(void) {
  if (!g_error) return 0;
  HPGMG_CHECK(hipStreamSynchronize(g_stream));
  HPGMG_CHECK(hipMemsetAsync(g_error_dev, 0, sizeof(unsigned), g_stream));
  HPGMG_CHECK(hipStreamSynchronize(g_stream));
  *(volatile unsigned *)g_error = 0;
  return 0;
}
// How many workgroups of the (variant, smoother) brick kernels this device holds at once, less a margin of one eighth for whatever else is running: a launch
// of more than that many bricks would wait for workgroups that cannot start (the caller then takes the launch-by-launch path).
int hpgmg_hip_brick_chain_capacity(int variant, int smoother, int brick) {
  int cap;
  switch (variant * 3 + smoother) {
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_CHEBY:  cap = brick_capacity_of<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_CHEBY>(brick); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_GSRB:   cap = brick_capacity_of<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_GSRB>(brick); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_JACOBI: cap = brick_capacity_of<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_JACOBI>(brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_CHEBY:    cap = brick_capacity_of<HPGMG_HIP_7PT_VC_POISSON, BV_CHEBY>(brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_GSRB:     cap = brick_capacity_of<HPGMG_HIP_7PT_VC_POISSON, BV_GSRB>(brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_JACOBI:   cap = brick_capacity_of<HPGMG_HIP_7PT_VC_POISSON, BV_JACOBI>(brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_CHEBY:            cap = brick_capacity_of<HPGMG_HIP_7PT_CC, BV_CHEBY>(brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_GSRB:             cap = brick_capacity_of<HPGMG_HIP_7PT_CC, BV_GSRB>(brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_JACOBI:           cap = brick_capacity_of<HPGMG_HIP_7PT_CC, BV_JACOBI>(brick); break;
    default: return 0;
  }
  return cap - cap / 8;
}

int hpgmg_hip_brick_chain(int n, const hpgmg_hip_brick_level *levels, const hpgmg_hip_level *below, int sweeps, int variant, int smoother,
                          int e_id, int R_id, double a, double b, int dir, int brick, int top_e_zero, int below_zero) {
  if (g_skip_launches) return record_error(hipErrorInvalidValue, "brick_chain: not replayable (the launch number is a kernel argument)");
  // a poll inside an earlier launch gave up: the results since then are void whatever is launched now; the host learns of it at its next scalar
  // (hpgmg_hip_brick_visit_error) and redoes the solve launch by launch or stops
  if (hpgmg_hip_brick_visit_error()) return 0;
  if (n < 1 || n > kBrickMaxLevels || sweeps < 1 || sweeps > kBrickMaxSweeps || (sweeps & 1) || dir < 0 || dir > 2 || (dir == 2 && n != 1))
    return record_error(hipErrorInvalidValue, "brick_chain: levels / sweeps / direction");
  for (int j = 0; j < n; j++) {
    const hpgmg_hip_level *next = (j + 1 < n) ? &levels[j + 1].L : below;
    if (!hpgmg_hip_brick_visit_supported(&levels[j].L, brick) || 2 * next->dim_i != levels[j].L.dim_i) return record_error(hipErrorInvalidValue, "brick_chain: level");
  }
  BrickArgs A = {};
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
  const int key = variant * 3 + smoother;
  switch (key) {
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_CHEBY:  rc = brick_launch_dir<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_CHEBY>(A, dir, brick); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_GSRB:   rc = brick_launch_dir<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_GSRB>(A, dir, brick); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_JACOBI: rc = brick_launch_dir<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_JACOBI>(A, dir, brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_CHEBY:    rc = brick_launch_dir<HPGMG_HIP_7PT_VC_POISSON, BV_CHEBY>(A, dir, brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_GSRB:     rc = brick_launch_dir<HPGMG_HIP_7PT_VC_POISSON, BV_GSRB>(A, dir, brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_JACOBI:   rc = brick_launch_dir<HPGMG_HIP_7PT_VC_POISSON, BV_JACOBI>(A, dir, brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_CHEBY:            rc = brick_launch_dir<HPGMG_HIP_7PT_CC, BV_CHEBY>(A, dir, brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_GSRB:             rc = brick_launch_dir<HPGMG_HIP_7PT_CC, BV_GSRB>(A, dir, brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_JACOBI:           rc = brick_launch_dir<HPGMG_HIP_7PT_CC, BV_JACOBI>(A, dir, brick); break;
    default: return record_error(hipErrorInvalidValue, "brick_chain: variant / smoother");
  }
  if (rc) return rc;
  HPGMG_LAUNCH_CHECK("brick_chain_kernel");
  brick_count_visits(n);
  return 0;
}

}  // extern "C"

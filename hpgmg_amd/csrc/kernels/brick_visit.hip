// brick_visit.hip -- one visit of a LAUNCH-BOUND level of MGVCycle (mg.c:1147-1163) as ONE launch: the level cut into bricks of 16^3 cells, one
// workgroup of 1024 lanes per brick (64^3: 64 workgroups, 32^3: 8), the brick's iterate and VECTOR_TEMP in LDS with a one-cell halo, its
// coefficients in registers (4 cells per lane) -- tail.hip's scheme, on more than one CU.
//   leg 0 (down):  smooth; residual -> TEMP; restriction(coarse.R <- TEMP); zero_vector(coarse.e)
//   leg 1 (up):    interpolation_vcycle (e += P coarse.e, piecewise constant); smooth
//   leg 2:         interpolation_fcycle (e = 0.0 e + P1 coarse.e, piecewise linear, the coarse ghost cells of apply_BCs_p1 formed on the fly), then leg 0 -- the
//                  step of FMGSolve (mg.c:1289-1293) that opens a V-cycle: its exchange + boundary + interpolation launches ride in the load of the visit
// i.e. the 5 + 4 launches of ~5 us each the per-operator path issues for the visit (4 sweeps of 1.3 MB each are not what they cost: a launch
// boundary and one memory round trip per sweep are).
//
// Between sweeps the bricks exchange their faces THROUGH MEMORY WITHOUT LEAVING THE KERNEL.  The XCDs' L2 caches are not coherent with each other
// inside a kernel, so a face cell travels as a 16-byte record {value, sequence number} written through (sc1) by the lane that owns the cell and
// polled (sc1 loads) by the lane that needs it: the record is its own flag -- one memory hop per exchange, no counter everybody adds to, no
// fence.  tools/microbench/p2p_flags.hip: 1.6 us per exchange for 8 .. 64 workgroups, against 3.1-3.3 us for a kernel boundary around the
// same traffic, 2.4-3.2 for data + flag, 5.7-18 for a central counter (what grid.sync() is).  Records are double-buffered by exchange parity: a
// brick can publish exchange n only after it has read all its neighbours' exchange n-1, which they published after reading n-2 -- the slot of
// parity n is free.  Sequence numbers never repeat (launch epoch x 64 + exchange), so records of earlier launches never match.
// Every other global access is of the ordinary kind and obeys one rule: within a launch an address is written by ONE workgroup only and never
// read by another (the halo of the first sweep comes from what EARLIER launches stored).
// All workgroups of the launch must be resident at once (<= 64 of 1024 lanes on 256 CUs: they are, unless other processes' launches of this
// kind fill the GPU -- hpgmg_hip_brick_visit_supported / HPGMG_BRICK_VISITS=0); a poll gives up after 2 s and raises the error flag the host
// checks at its next synchronisation, so a mistake here ends as an abort with a message, not as a hung GPU.
// Arithmetic: the expression trees of tail.hip / the streaming kernels (stencil_math.hpp, chebyshev.c:86-95, gsrb.c:100-104, jacobi.c:50-56,
// residual.c:42-48, restriction.c:54-57, interpolation_p0.c:43): bit-identical to the per-operator path; tests/test_gpu_operators.py runs both.
#include "common.hpp"
#include "stencil_math.hpp"
#include "dense_levels.hpp"

namespace hpgmg {

typedef unsigned long long u64;
struct alignas(16) FaceCell { double v; u64 seq; };
typedef unsigned __attribute__((ext_vector_type(4))) u4v;

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
constexpr int kBrickMaxSweeps = 8;
constexpr size_t kFaceRecords = (size_t)2 * 512 * 6 * 64;      // both parities of 8^3 bricks of 8^3 cells (= 4^3 bricks of 16^3 take half of it)
constexpr u64 kPollTicks = 200000000ull;            // 2 s of the 100 MHz clock

struct BrickArgs {
  hpgmg_hip_level L, C;             // the level visited; the next coarser one (restriction target / interpolation source)
  double h2inv, a, b;
  double c1[kBrickMaxSweeps], c2[kBrickMaxSweeps];
  int sweeps, e_id, R_id;
  int side;                         // bricks per dimension
  int e_zero;                       // legs 0: the correction counts as +0.0 (zero_vector came before, mg.c:1153): it is not read, and the cells of its padded boxes that
                                    // no brick stores at the end (ghost zone, padding) are cleared HERE -- the launch that visited the finer level left it alone
  int coarse_zero;                  // legs 0, 2: zero_vector(C.e) at the end (0: the launch that visits C does it, see e_zero)
  FaceCell *faces;                  // [2][workgroup][6][B^2]
  u64 epoch;                        // launch number x 64: the first sequence number of this launch is epoch + 1
  unsigned *error;                  // pinned host word: set when a poll gave up
};

__device__ __forceinline__ void face_store(FaceCell *p, double v, u64 seq) {
  const long long b = __double_as_longlong(v);
  u4v w; w.x = (unsigned)b; w.y = (unsigned)(b >> 32); w.z = (unsigned)seq; w.w = (unsigned)(seq >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ FaceCell face_load(const FaceCell *p) {
  u4v w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  FaceCell c; c.v = __longlong_as_double((long long)(((u64)w.y << 32) | w.x)); c.seq = ((u64)w.w << 32) | w.z;
  return c;
}

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

template <int V, int SM, int LEG, int B>
__global__ __launch_bounds__(BrickGeom<B>::Threads) void brick_visit_kernel(const BrickArgs A) {
  using BG = BrickGeom<B>;
  constexpr int kBrick = B, kBrickCells = BG::Cells, kBrickThreads = BG::Threads, kBrickPerLane = BG::PerLane, kHaloW = BG::W, kHaloPlane = BG::Plane, kHaloCells = BG::Halo;
  constexpr int kFaceCells = BG::Face, kStepPos = BG::StepPos, kStepK = BG::StepK;
  auto hpos = [](int li, int lj, int lk) { return BG::hpos(li, lj, lk); };
  auto face_cell = [](int f, int u, int v, int depth, int &li, int &lj, int &lk) { BG::face_cell(f, u, v, depth, li, lj, lk); };
  constexpr bool kVC = (V != HPGMG_HIP_7PT_CC);
  constexpr bool kHelm = (V == HPGMG_HIP_7PT_VC_HELMHOLTZ);
  constexpr bool kUp = (LEG == 1), kDown = (LEG != 1), kFInterp = (LEG == 2);
  extern __shared__ double brick_lds[];
  double *const sx = brick_lds, *const st = brick_lds + kHaloCells;
  const int t = (int)threadIdx.x, wg = (int)blockIdx.x, side = A.side, nwg = side * side * side;
  const int bx = wg % side, by = (wg / side) % side, bz = wg / (side * side);
  const int D = A.L.dim_i, o_i = bx * kBrick, o_j = by * kBrick, o_k = bz * kBrick;
  const LevelGeom G = geom_of(A.L), GC = geom_of(A.C);
  FaceCell *const faces = A.faces;
  const u64 epoch = A.epoch;
  const int e_id = A.e_id;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool gave_up = false;

  // what the first sweep reads at global cell (ci, cj, ck) of this level, given what is stored there
  auto start_value = [&](double stored, int ci, int cj, int ck) -> double {
    if (kUp) {               // interpolation_vcycle: e = 1.0*e + (coarse parent), interpolation_p0.c:43
      const CellRef c = locate(GC, ci >> 1, cj >> 1, ck >> 1);
      return 1.0 * stored + vec_origin(A.C, c.box, e_id)[c.ijk];
    }
    if (kFInterp) {          // interpolation_fcycle, piecewise linear (interpolation_p1.c:40-70): f = 0.0 f + 27/64 c + 9/64 (3 face neighbours) + 3/64 (3 edge
      // neighbours) + 1/64 corner, an even fine cell leaning on the coarse neighbour behind it, an odd one on the one ahead.  A coarse ghost cell is -, +, -
      // its mirror image for 1, 2, 3 directions leaving the domain (exchange_boundary + apply_BCs_p1, BOX shape, boundary_fd.c:35-38), formed here.
      const int Dc = A.C.dim_i;
      auto coarse = [&](int qi, int qj, int qk) -> double {
        double sg = 1.0;
        if (qi < 0) { qi = 0; sg = -sg; } else if (qi >= Dc) { qi = Dc - 1; sg = -sg; }
        if (qj < 0) { qj = 0; sg = -sg; } else if (qj >= Dc) { qj = Dc - 1; sg = -sg; }
        if (qk < 0) { qk = 0; sg = -sg; } else if (qk >= Dc) { qk = Dc - 1; sg = -sg; }
        const CellRef r = locate(GC, qi, qj, qk);
        return sg * vec_origin(A.C, r.box, e_id)[r.ijk];
      };
      const int qi = ci >> 1, qj = cj >> 1, qk = ck >> 1, di = (ci & 1) ? 1 : -1, dj = (cj & 1) ? 1 : -1, dk = (ck & 1) ? 1 : -1;
      double v = 0.0 * stored;
      v = v + 0.421875 * coarse(qi, qj, qk);
      v = v + 0.140625 * coarse(qi, qj, qk + dk);
      v = v + 0.140625 * coarse(qi, qj + dj, qk);
      v = v + 0.046875 * coarse(qi, qj + dj, qk + dk);
      v = v + 0.140625 * coarse(qi + di, qj, qk);
      v = v + 0.046875 * coarse(qi + di, qj, qk + dk);
      v = v + 0.046875 * coarse(qi + di, qj + dj, qk);
      v = v + 0.015625 * coarse(qi + di, qj + dj, qk + dk);
      return v;
    }
    return stored;
  };
  // a lane's cells: (li, lj) fixed, lk = lk0 + kStepK m -- one LDS position and one global coordinate triple describe all of them
  const int li0 = t % kBrick, lj0 = (t / kBrick) % kBrick, lk0 = t / kFaceCells;
  const int pos0 = hpos(li0, lj0, lk0), gi = o_i + li0, gj = o_j + lj0, gk0 = o_k + lk0;
  CellCoef<V> q[kBrickPerLane];

  // ---- the brick: iterate (+ the coarse parent on the way up), VECTOR_TEMP, coefficients
#pragma unroll
  for (int m = 0; m < kBrickPerLane; m++) {
    const int p = pos0 + m * kStepPos, gk = gk0 + m * kStepK;
    const CellRef w = locate(G, gi, gj, gk);
    const int box = w.box, ijk = w.ijk, jS = A.L.jStride, kS = A.L.kStride;
    sx[p] = start_value(A.e_zero ? 0.0 : vec_origin(A.L, box, e_id)[ijk], gi, gj, gk);
    st[p] = vec_origin(A.L, box, VECTOR_TEMP)[ijk];
    q[m].rhs = vec_origin(A.L, box, A.R_id)[ijk];
    q[m].dinv = vec_origin(A.L, box, VECTOR_DINV)[ijk];
    q[m].bi0 = q[m].bi1 = q[m].bj0 = q[m].bj1 = q[m].bk0 = q[m].bk1 = q[m].al = 0.0;
    if (kVC) {
      const double *bi = vec_origin(A.L, box, VECTOR_BETA_I), *bj = vec_origin(A.L, box, VECTOR_BETA_J), *bk = vec_origin(A.L, box, VECTOR_BETA_K);
      q[m].bi0 = bi[ijk]; q[m].bi1 = bi[ijk + 1]; q[m].bj0 = bj[ijk]; q[m].bj1 = bj[ijk + jS]; q[m].bk0 = bk[ijk]; q[m].bk1 = bk[ijk + kS];
    }
    if (kHelm) q[m].al = vec_origin(A.L, box, VECTOR_ALPHA)[ijk];
  }
  // ---- the halo of the FIRST sweep's input: the neighbouring bricks' cells as earlier launches left them (+ their coarse parents on the way up: the same
  // expression the owning brick forms).  Lane roles for everything on faces: cell fc = (f B + v) B + u of the 6 B^2 face cells.
  for (int fc = t; fc < 6 * kFaceCells; fc += kBrickThreads) {
    const int f = fc / kFaceCells, u = fc % kBrick, v = (fc / kBrick) % kBrick;
    int li, lj, lk;
    face_cell(f, u, v, 1, li, lj, lk);
    const int hi = o_i + li, hj = o_j + lj, hk = o_k + lk;
    if (hi < 0 || hi >= D || hj < 0 || hj >= D || hk < 0 || hk >= D) continue;      // beyond the domain: the Dirichlet rule, never read
    const CellRef r = locate(G, hi, hj, hk);
    sx[hpos(li, lj, lk)] = start_value(A.e_zero ? 0.0 : vec_origin(A.L, r.box, e_id)[r.ijk], hi, hj, hk);
  }
  __syncthreads();

  // ---- smooth(): chebyshev.c:43-99 / gsrb.c:24-132 / jacobi.c:17-62 (an even number of sweeps: the result ends in sx)
  int exchange_n = 0;
  for (int s = 0; s < A.sweeps; s++) {
    const double *src = (SM != BV_GSRB && (s & 1)) ? st : sx;
    double *dst = (SM == BV_GSRB) ? sx : ((s & 1) ? sx : st);
    const double c1 = A.c1[s], c2 = A.c2[s];
#pragma unroll
    for (int m = 0; m < kBrickPerLane; m++) {
      const int p = pos0 + m * kStepPos, gk = gk0 + m * kStepK;
      if (SM == BV_GSRB && ((gi ^ gj ^ gk ^ s) & 1) != 0) continue;      // global parity: box.low folded in (gsrb.c:55); red cells read black neighbours only
      const double xc = src[p];
      const double Ax = brick_apply<V, kHaloW, kHaloPlane>(src, p, gi, gj, gk, D, q[m], A.a, A.b, A.h2inv);
      if (SM == BV_CHEBY)     { const double xnm1 = dst[p]; dst[p] = xc + c1 * (xc - xnm1) + c2 * q[m].dinv * (q[m].rhs - Ax); }
      else if (SM == BV_GSRB) { dst[p] = xc + q[m].dinv * (q[m].rhs - Ax); }
      else                    { dst[p] = xc + (2.0 / 3.0) * q[m].dinv * (q[m].rhs - Ax); }
    }
    __syncthreads();
    if (kDown || s + 1 < A.sweeps) {      // (the way down goes on to the residual of the result)
      // one exchange: the faces of dst go out, the neighbours' faces come into its halo
      const int par = exchange_n & 1;
      const u64 seq = epoch + 1 + (u64)exchange_n;
      exchange_n++;
      FaceCell *mine = faces + ((size_t)par * nwg + wg) * 6 * kFaceCells;
      for (int fc = t; fc < 6 * kFaceCells; fc += kBrickThreads) {
        const int f = fc / kFaceCells, u = fc % kBrick, v = (fc / kBrick) % kBrick;
        const int bc = (f < 2) ? bx : ((f < 4) ? by : bz);
        if ((f & 1) ? (bc == side - 1) : (bc == 0)) continue;                          // no brick beyond this face
        int li, lj, lk;
        face_cell(f, u, v, 0, li, lj, lk);
        face_store(mine + fc, dst[hpos(li, lj, lk)], seq);
      }
      for (int fc = t; fc < 6 * kFaceCells; fc += kBrickThreads) {
        const int f = fc / kFaceCells, u = fc % kBrick, v = (fc / kBrick) % kBrick;
        const int bc = (f < 2) ? bx : ((f < 4) ? by : bz);
        if ((f & 1) ? (bc == side - 1) : (bc == 0)) continue;
        const int step = (f < 2) ? 1 : ((f < 4) ? side : side * side), n = wg + ((f & 1) ? step : -step);
        const FaceCell *theirs = faces + (((size_t)par * nwg + n) * 6 + (f ^ 1)) * kFaceCells + (fc % kFaceCells);
        FaceCell x = face_load(theirs);
        while (x.seq != seq) {
          if (__builtin_amdgcn_s_memrealtime() - t0 > kPollTicks) { gave_up = true; break; }
          __builtin_amdgcn_s_sleep(1);
          x = face_load(theirs);
        }
        int li, lj, lk;
        face_cell(f, u, v, 1, li, lj, lk);
        dst[hpos(li, lj, lk)] = x.v;
      }
      __syncthreads();
    }
  }

  if (kDown) {                                        // residual -> TEMP (residual.c:42-48)
#pragma unroll
    for (int m = 0; m < kBrickPerLane; m++) {
      const int p = pos0 + m * kStepPos;
      const double Ax = brick_apply<V, kHaloW, kHaloPlane>(sx, p, gi, gj, gk0 + m * kStepK, D, q[m], A.a, A.b, A.h2inv);
      st[p] = q[m].rhs - Ax;                          // each lane overwrites only its own TEMP cells: no hazard with the reads of sx
    }
    __syncthreads();
  }

  // ---- leave e and TEMP in global memory as the per-operator sequence would
#pragma unroll
  for (int m = 0; m < kBrickPerLane; m++) {
    const int p = pos0 + m * kStepPos;
    const CellRef w = locate(G, gi, gj, gk0 + m * kStepK);
    vec_origin(A.L, w.box, e_id)[w.ijk] = sx[p];
    vec_origin(A.L, w.box, VECTOR_TEMP)[w.ijk] = st[p];
  }

  if (kDown) {
    // restriction(coarse.R <- TEMP): 0.125 * sum of the 8 children in the reference's order (restriction.c:54-57); this brick's 8^3 coarse cells
    if (t < kBrickCells / 8) {
      constexpr int H = kBrick / 2;
      const int ci = t % H, cj = (t / H) % H, ck = t / (H * H);
      const double *f = st + hpos(2 * ci, 2 * cj, 2 * ck);
      double v = f[0] + f[1]; v = v + f[kHaloW]; v = v + f[1 + kHaloW]; v = v + f[kHaloPlane]; v = v + f[1 + kHaloPlane]; v = v + f[kHaloW + kHaloPlane];
      v = v + f[1 + kHaloW + kHaloPlane];
      const CellRef c = locate(GC, (o_i >> 1) + ci, (o_j >> 1) + cj, (o_k >> 1) + ck);
      vec_origin(A.C, c.box, A.R_id)[c.ijk] = v * 0.125;
    }
    // zero_vector(coarse.e): the whole padded boxes, ghosts included (misc.c:6-44), each workgroup a slice of the flat range
    if (A.coarse_zero) {
      const int vol = A.C.volume, total = A.C.num_boxes * vol, per = (total + nwg - 1) / nwg;
      const int lo = wg * per, hi = (lo + per < total) ? lo + per : total;
      for (int z = lo + t; z < hi; z += kBrickThreads) {
        const int box = z / vol;
        (A.C.box_base[box] + (size_t)e_id * (size_t)vol)[z - box * vol] = 0.0;
      }
    }
  }
  if (A.e_zero) {
    // the part of zero_vector(this level's e) the stores above do not overwrite: every cell of the padded boxes that is not an interior cell
    const int vol = A.L.volume, total = A.L.num_boxes * vol, per = (total + nwg - 1) / nwg, jS = A.L.jStride, kS = A.L.kStride, g = A.L.ghosts, d = A.L.dim;
    const int lo = wg * per, hi = (lo + per < total) ? lo + per : total;
    for (int z = lo + t; z < hi; z += kBrickThreads) {
      const int box = z / vol, off = z - box * vol, k = off / kS, r = off - k * kS, j = r / jS, i = r - j * jS;
      if (i >= g && i < g + d && j >= g && j < g + d && k >= g && k < g + d) continue;
      (A.L.box_base[box] + (size_t)e_id * (size_t)vol)[off] = 0.0;
    }
  }
  if (gave_up && A.error) __hip_atomic_store(A.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static FaceCell *g_faces = nullptr;
static unsigned *g_error = nullptr;          // pinned host word
static u64 g_epoch = 0;
static long long g_visits = 0;

template <int V, int SM, int LEG, int B>
static int brick_launch(const BrickArgs &A) {
  static bool once = false;
  const size_t lds = (size_t)2 * BrickGeom<B>::Halo * sizeof(double);
  if (!once) { HPGMG_CHECK(hipFuncSetAttribute((const void *)brick_visit_kernel<V, SM, LEG, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
  hipLaunchKernelGGL((brick_visit_kernel<V, SM, LEG, B>), dim3(A.side * A.side * A.side), dim3(BrickGeom<B>::Threads), lds, g_stream, A);
  return 0;
}
template <int V, int SM>
static int brick_launch_leg(const BrickArgs &A, int leg, int brick) {
  if (brick == 16) return leg == 0 ? brick_launch<V, SM, 0, 16>(A) : (leg == 1 ? brick_launch<V, SM, 1, 16>(A) : brick_launch<V, SM, 2, 16>(A));
  return leg == 0 ? brick_launch<V, SM, 0, 8>(A) : (leg == 1 ? brick_launch<V, SM, 1, 8>(A) : brick_launch<V, SM, 2, 8>(A));
}

}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

int hpgmg_hip_brick_visit_max_sweeps(void) { return kBrickMaxSweeps; }
long long hpgmg_hip_brick_visits(void) { return g_visits; }      // launches so far (tests)
// 1: a level of dim^3 cells can be visited as bricks of brick^3 cells (brick = 16: 2^3 or 4^3 of them; brick = 8: 2^3 .. 8^3)
int hpgmg_hip_brick_visit_supported(const hpgmg_hip_level *L, int brick) {
  if (brick != 8 && brick != 16) return 0;
  const int side = L->dim_i / brick, max_side = (brick == 16) ? BrickGeom<16>::MaxSide : BrickGeom<8>::MaxSide;
  return L->dim_i == L->dim_j && L->dim_i == L->dim_k && L->dim_i % brick == 0 && side >= 2 && side <= max_side && L->dim > 0 && L->dim_i % L->dim == 0;
}
// 0: fine; 1: a poll of an earlier visit gave up (the results since then are not to be used)
int hpgmg_hip_brick_visit_error(void) { return (g_error && *(volatile unsigned *)g_error) ? 1 : 0; }

// leg 0: smooth + residual + restriction + zero_vector(coarse e); leg 1: interpolation_vcycle + smooth; leg 2: interpolation_fcycle, then leg 0.
// c1 / c2: the level's Chebyshev coefficients.
int hpgmg_hip_brick_visit(const hpgmg_hip_level *L, const hpgmg_hip_level *C, double h2inv, const double *c1, const double *c2, int sweeps,
                          int variant, int smoother, int e_id, int R_id, double a, double b, int leg, int brick, int e_zero, int coarse_zero) {
  if (g_skip_launches) return record_error(hipErrorInvalidValue, "brick_visit: not replayable (the launch number is a kernel argument)");
  if (!hpgmg_hip_brick_visit_supported(L, brick) || sweeps < 1 || sweeps > kBrickMaxSweeps || (sweeps & 1) || leg < 0 || leg > 2 || 2 * C->dim_i != L->dim_i)
    return record_error(hipErrorInvalidValue, "brick_visit: level / sweeps / leg");
  if (!g_faces) {
    HPGMG_CHECK(hipMalloc((void **)&g_faces, kFaceRecords * sizeof(FaceCell)));
    HPGMG_CHECK(hipMemset(g_faces, 0, kFaceRecords * sizeof(FaceCell)));
    HPGMG_CHECK(hipHostMalloc((void **)&g_error, 64, hipHostMallocDefault));
    *g_error = 0;
    HPGMG_CHECK(hipDeviceSynchronize());
  }
  BrickArgs A = {};
  A.L = *L; A.C = *C; A.h2inv = h2inv; A.a = a; A.b = b; A.sweeps = sweeps; A.e_id = e_id; A.R_id = R_id;
  for (int s = 0; s < sweeps; s++) { A.c1[s] = c1 ? c1[s] : 0.0; A.c2[s] = c2 ? c2[s] : 0.0; }
  A.side = L->dim_i / brick;
  A.e_zero = (leg == 0 && e_zero) ? 1 : 0; A.coarse_zero = coarse_zero ? 1 : 0;
  A.faces = g_faces; A.error = g_error;
  g_epoch += 64; A.epoch = g_epoch;
  int rc;
  const int key = variant * 3 + smoother;
  switch (key) {
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_CHEBY:  rc = brick_launch_leg<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_CHEBY>(A, leg, brick); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_GSRB:   rc = brick_launch_leg<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_GSRB>(A, leg, brick); break;
    case HPGMG_HIP_7PT_VC_HELMHOLTZ * 3 + BV_JACOBI: rc = brick_launch_leg<HPGMG_HIP_7PT_VC_HELMHOLTZ, BV_JACOBI>(A, leg, brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_CHEBY:    rc = brick_launch_leg<HPGMG_HIP_7PT_VC_POISSON, BV_CHEBY>(A, leg, brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_GSRB:     rc = brick_launch_leg<HPGMG_HIP_7PT_VC_POISSON, BV_GSRB>(A, leg, brick); break;
    case HPGMG_HIP_7PT_VC_POISSON * 3 + BV_JACOBI:   rc = brick_launch_leg<HPGMG_HIP_7PT_VC_POISSON, BV_JACOBI>(A, leg, brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_CHEBY:            rc = brick_launch_leg<HPGMG_HIP_7PT_CC, BV_CHEBY>(A, leg, brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_GSRB:             rc = brick_launch_leg<HPGMG_HIP_7PT_CC, BV_GSRB>(A, leg, brick); break;
    case HPGMG_HIP_7PT_CC * 3 + BV_JACOBI:           rc = brick_launch_leg<HPGMG_HIP_7PT_CC, BV_JACOBI>(A, leg, brick); break;
    default: return record_error(hipErrorInvalidValue, "brick_visit: variant / smoother");
  }
  if (rc) return rc;
  HPGMG_LAUNCH_CHECK("brick_visit_kernel");
  g_visits++;
  return 0;
}

}  // extern "C"

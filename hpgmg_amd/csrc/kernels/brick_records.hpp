// brick_records.hpp -- what the brick launches (brick_visit.hip: 7-point; brick_wide.hip: 27-point / fv4) share: the 16-byte self-flagging record a cell
// travels in between workgroups of ONE launch, the record areas, the launch epoch, the error word and the co-residency guard.
//
// A record is 16 bytes, {tag, value low word, value high word, tag}, written with ONE 16-byte store and read with ONE 16-byte load -- buffer_store / buffer_load
// ... sc1: written through, read past this XCD's L2 (the L2s of the eight XCDs are not coherent inside a kernel) -- by the lane that owns the cell and the lane that
// needs it, and accepted only when BOTH tags carry the expected number: should the 16 bytes ever be observed in two pieces, one end still shows the old tag and the
// record reads as "not yet".  The accesses are the compiler's own buffer intrinsics on a descriptor of the record areas, not inline assembly: the compiler pads the
// store against the hazard of its data registers and keeps several polls of a lane in flight under one wait.  (Two 8-byte atomics per record were tried: twice the
// memory transactions, +10 us per launch of a 64^3 level.)  Tags never repeat inside the life of the record areas: tag = launch epoch (launch number x 64, 32 bits)
// + a code for the record's role; when the 32-bit epoch would wrap, the areas are cleared first (brick_records_for_launch).
#pragma once
#include "common.hpp"

namespace hpgmg {

typedef unsigned long long u64;
typedef unsigned __attribute__((ext_vector_type(4))) u4v;
struct alignas(16) FaceCell { unsigned tag0, lo, hi, tag1; };

constexpr int kBrickMaxSweeps = 8, kBrickMaxLevels = 4, kBrickMaxWgs = 512;
// record areas, each per level of a chain: faces [2 parities][workgroup][6][B^2] (7-point: 8^3 bricks of 8^3 cells fill it; 4^3 bricks of 16^3 take half;
// 27-point / fv4: [2 parities][workgroup][512 cells], the same number), one record per cell for what goes down (restricted residuals) and up (corrections),
// one gate per brick
constexpr size_t kFaceRecords = (size_t)2 * kBrickMaxWgs * 512, kCellRecords = (size_t)kBrickMaxWgs * 512;
constexpr size_t kRecordsTotal = (size_t)kBrickMaxLevels * (kFaceRecords + 2 * kCellRecords + kBrickMaxWgs);      // faces | down | up | gate, kBrickMaxLevels of each
constexpr u64 kPollTicks = 200000000ull;            // 2 s of the 100 MHz clock
constexpr u64 kPollLookTicks = 10000ull;            // a poll that has waited 100 us looks at the error word: a launch behind a failed one gives up at once
// tags inside a launch (added to its epoch, a multiple of 64): 1 + 12 j + n = exchange n of level j (j < 4, n < 12: <= 48); 50 + j, 54 + j, 58 + j = what level j
// receives from the finer level / hands to the finer level / its gate
enum { SEQ_FACES = 1, SEQ_DOWN = 50, SEQ_UP = 54, SEQ_GATE = 58 };
static_assert(SEQ_FACES + 12 * kBrickMaxLevels <= SEQ_DOWN && SEQ_GATE + kBrickMaxLevels <= 64, "tag codes");

struct BrickRecords {
  FaceCell *faces, *down, *up, *gate;
  unsigned epoch;                   // launch number x 64
  unsigned *error;                  // pinned host word: set when a poll gave up
  unsigned *error_dev;              // its mirror in device memory (what a long poll looks at)
};

// the record areas as a buffer (raw, byte-addressed; 0x00020000: the descriptor's format word for gfx90a and later); kAuxSc1: the sc1 bit of the cache policy
constexpr int kAuxSc1 = 16;
struct RecordWindow { __amdgpu_buffer_rsrc_t rsrc; const FaceCell *base; };
__device__ __forceinline__ RecordWindow record_window(const BrickRecords &R) {
  RecordWindow w;
  w.base = R.faces;
  w.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)R.faces, 0, (int)(kRecordsTotal * sizeof(FaceCell)), 0x00020000);
  return w;
}
__device__ __forceinline__ unsigned record_offset(const RecordWindow &W, const FaceCell *p) { return (unsigned)((const char *)p - (const char *)W.base); }
__device__ __forceinline__ void face_store(const RecordWindow &W, FaceCell *p, double v, unsigned tag) {
  const u64 bits = (u64)__double_as_longlong(v);
  u4v w; w.x = tag; w.y = (unsigned)bits; w.z = (unsigned)(bits >> 32); w.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(w, W.rsrc, (int)record_offset(W, p), 0, kAuxSc1);
}
typedef u4v FaceWords;
__device__ __forceinline__ FaceWords face_load(const RecordWindow &W, const FaceCell *p) { return __builtin_amdgcn_raw_buffer_load_b128(W.rsrc, (int)record_offset(W, p), 0, kAuxSc1); }
__device__ __forceinline__ bool face_ready(const FaceWords &w, unsigned tag) { return w.x == tag && w.w == tag; }
__device__ __forceinline__ double face_value(const FaceWords &w) { return __longlong_as_double((long long)(((u64)w.z << 32) | w.y)); }
// a poll that is not answered yet: give up after 2 s; after 100 us look (once) at the error word -- a launch behind a failed one ends at once
__device__ __forceinline__ bool poll_expired(u64 t0, bool &looked, const unsigned *error_dev) {
  const u64 waited = __builtin_amdgcn_s_memrealtime() - t0;
  if (waited > kPollTicks) return true;
  if (waited > kPollLookTicks && !looked) {
    looked = true;
    if (error_dev && __hip_atomic_load(error_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
  }
  return false;
}
// the value of a record once both its words carry `tag` (nap: s_sleep units between polls)
__device__ __forceinline__ double record_wait(const RecordWindow &W, const FaceCell *p, unsigned tag, u64 t0, bool &gave_up, const unsigned *error_dev, int nap = 1) {
  FaceWords x = face_load(W, p);
  bool looked = false;
  while (!face_ready(x, tag)) {
    if (poll_expired(t0, looked, error_dev)) { gave_up = true; break; }
    if (nap > 1) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
    x = face_load(W, p);
  }
  return face_value(x);
}
// N records at once: every poll round has all the still-missing ones in flight together (one memory round trip per round, not one per record).
// pending: bit m set = record p[m] is wanted; out[m] is written for those
template <int N>
__device__ __forceinline__ void record_wait_many(const RecordWindow &W, const FaceCell *const (&p)[N], unsigned pending, unsigned tag, double (&out)[N], u64 t0, bool &gave_up, const unsigned *error_dev) {
  bool looked = false;
  while (pending) {
    FaceWords x[N];
#pragma unroll
    for (int m = 0; m < N; m++) if ((pending >> m) & 1u) x[m] = face_load(W, p[m]);
#pragma unroll
    for (int m = 0; m < N; m++) if (((pending >> m) & 1u) && face_ready(x[m], tag)) { out[m] = face_value(x[m]); pending &= ~(1u << m); }
    if (!pending) break;
    if (poll_expired(t0, looked, error_dev)) { gave_up = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void brick_raise_error(const BrickRecords &R) {
  if (R.error) __hip_atomic_store(R.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (R.error_dev) __hip_atomic_store(R.error_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// host side (brick_visit.hip owns the state)
int brick_records_for_launch(BrickRecords *out);       // the areas (allocated at first use) and the next epoch; 0 or a recorded error
int brick_workgroups_resident(const void *kernel, int threads, size_t lds_bytes);      // how many workgroups of this kernel the device holds at once (occupancy x CUs; HPGMG_TEST_BRICK_CAPACITY overrides)
int brick_test_absent_wg(void);                        // HPGMG_TEST_BRICK_ABSENT: this workgroup leaves at once (tests), -1: none
void brick_count_visits(int n);
bool brick_error_pending(void);

}  // namespace hpgmg

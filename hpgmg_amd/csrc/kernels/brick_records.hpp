// brick_records.hpp -- what the brick launches (brick_visit.hip: 7-point; brick_wide.hip: 27-point / fv4) share: the 16-byte self-flagging record a cell
// travels in between workgroups of ONE launch, the record areas, the launch epoch, the error word and the co-residency guard.
//
// A record is {tag, value low word, value high word, tag}: written through (sc1) with one 16-byte store by the lane that owns the cell, polled (sc1 loads) by
// the lane that needs it, accepted only when BOTH tags carry the expected number.  The architecture does not promise that a 16-byte store is observed
// whole; a store that lands in two pieces (of 4 or 8 bytes, in either order) shows a reader one old tag until the second piece is there, i.e. reads as "not
// yet".  Tags never repeat inside the life of the record areas: tag = launch epoch (launch number x 64, 32 bits) + a code for the record's role; when the
// 32-bit epoch would wrap, the areas are cleared first (brick_next_epoch).
#pragma once
#include "common.hpp"

namespace hpgmg {

typedef unsigned long long u64;
typedef unsigned __attribute__((ext_vector_type(4))) u4v;
struct alignas(16) FaceCell { unsigned tag0, lo, hi, tag1; };

constexpr int kBrickMaxSweeps = 8, kBrickMaxLevels = 3, kBrickMaxWgs = 512;
// record areas, each per level of a chain: faces [2 parities][workgroup][6][B^2] (7-point: 8^3 bricks of 8^3 cells fill it; 4^3 bricks of 16^3 take half;
// 27-point / fv4: [2 parities][workgroup][512 cells], the same number), one record per cell for what goes down (restricted residuals) and up (corrections),
// one gate per brick
constexpr size_t kFaceRecords = (size_t)2 * kBrickMaxWgs * 512, kCellRecords = (size_t)kBrickMaxWgs * 512;
constexpr u64 kPollTicks = 200000000ull;            // 2 s of the 100 MHz clock
constexpr u64 kPollLookTicks = 10000ull;            // a poll that has waited 100 us looks at the error word: a launch behind a failed one gives up at once
// tags inside a launch (added to its epoch, a multiple of 64): 1 + 12 j + n = exchange n of level j (n < 12); 40 + j, 48 + j, 56 + j = what level j
// receives from the finer level / hands to the finer level / its gate
enum { SEQ_FACES = 1, SEQ_DOWN = 40, SEQ_UP = 48, SEQ_GATE = 56 };

struct BrickRecords {
  FaceCell *faces, *down, *up, *gate;
  unsigned epoch;                   // launch number x 64
  unsigned *error;                  // pinned host word: set when a poll gave up
  unsigned *error_dev;              // its mirror in device memory (what a long poll looks at)
};

__device__ __forceinline__ void face_store(FaceCell *p, double v, unsigned tag) {
  const long long b = __double_as_longlong(v);
  u4v w; w.x = tag; w.y = (unsigned)b; w.z = (unsigned)(b >> 32); w.w = tag;
  // the s_nop: the data registers of a VMEM store of more than 8 bytes are read for some cycles after issue, and a VALU write to them in that window
  // corrupts the store (the hazard LLVM's GCNHazardRecognizer pads its own stores against: 1 wait state, 2 on gfx940 and later; it cannot see into an asm
  // statement).  s_nop 3 = 4 wait states inside the same statement, so nothing can be scheduled between the store and the padding.
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" :: "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ u4v face_load(const FaceCell *p) {
  u4v w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  return w;
}
// the value of a record once both its tags carry `tag` (nap: s_sleep units between polls)
__device__ __forceinline__ double record_wait(const FaceCell *p, unsigned tag, u64 t0, bool &gave_up, const unsigned *error_dev, int nap = 1) {
  u4v x = face_load(p);
  bool looked = false;
  while (x.x != tag || x.w != tag) {
    const u64 waited = __builtin_amdgcn_s_memrealtime() - t0;
    if (waited > kPollTicks) { gave_up = true; break; }
    if (waited > kPollLookTicks && !looked) {
      looked = true;
      if (error_dev && __hip_atomic_load(error_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { gave_up = true; break; }
    }
    if (nap > 1) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
    x = face_load(p);
  }
  return __longlong_as_double((long long)(((u64)x.z << 32) | x.y));
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

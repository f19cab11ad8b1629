// common.hpp -- shared device/host helpers of libhpgmg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hpgmg_hip.h"
#include "hpgmg_operators.h"   // VECTOR_* ids

namespace hpgmg {

extern hipStream_t g_stream;          // stream every launcher enqueues on
extern int g_skip_launches;           // 1 while a hipGraph replay segment is open (graph.hip): launchers do nothing
int  record_error(hipError_t e, const char *where);
#define HPGMG_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hpgmg::record_error(e_, #call); } while (0)
#define HPGMG_SKIP_IF_REPLAY() do { if (hpgmg::g_skip_launches) return 0; } while (0)
#define HPGMG_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return hpgmg::record_error(e_, name); } while (0)

double *reduction_scratch(int n);                       // blas1.hip: device scratch of >= n doubles for per-workgroup partial results
int finish_max_reduction(int n, double init, double *out);   // blas1.hip: fold n partial maxima, publish to the host, wait for the value
// Scalar results go straight to a pinned host slot {value, sequence}: the last lane stores the value, fences at system scope, then stores the
// launch's sequence number; the host polls the sequence instead of paying a full stream synchronisation (blas1.hip owns the slot).
struct ResultSlot { double value; unsigned long long seq; };
ResultSlot *reduction_slot_next(unsigned long long *seq_out);   // blas1.hip: the slot and the sequence number the next publishing launch must use
int reduction_fetch(double *out);                               // blas1.hip: wait for that launch's value
double reduction_second_value(void);                            // blas1.hip: the word behind the slot's sequence number (a launch that publishes two values), after reduction_fetch()

constexpr int kXcds = 8;              // MI355X: 8 XCDs, workgroup b is placed on XCD b % 8

// Physical block id -> logical work id such that each XCD (hence each private L2)
// owns one CONTIGUOUS range of logical ids.  Logical ids are laid out box-major,
// then k, j, i, so an XCD works on a compact slab and the planes shared by
// neighbouring tiles are served by its own L2 instead of being refetched by another.
__device__ __forceinline__ int xcd_logical_block(int physical, int per_xcd) {
  return (physical % kXcds) * per_xcd + physical / kXcds;
}
inline int grid_for(int logical_blocks, int *per_xcd) {
  *per_xcd = (logical_blocks + kXcds - 1) / kXcds;
  return *per_xcd * kXcds;
}

// Launching a tiled kernel in two parts (multi-rank: the halo exchange runs on another stream under part 1).  A tile = (box, k chunk, tj, ti)
// numbered ti + tiles_i * (tj + tiles_j * (ck + chunks_k * box)).  Part 1: the tiles whose halo reaches no IMAGE of another rank's box (a
// neighbour index >= L->num_boxes in box_nbr; include/hpgmg_hip.h) -- and, with walls_to_part2, no domain wall either; part 2: the others.
// Returns the dispatch list (slot -> tile, padded with `total`) for a grid of *grid workgroups with *per_xcd per XCD; *grid = 0: nothing to launch.
extern int g_tile_part;               // 0: whole launches; 1 / 2: the next tiled launches run that part only
const int *tile_part_order(const hpgmg_hip_level *L, int tiles_i, int tiles_j, int chunks_k, int part, bool walls_to_part2, int *grid, int *per_xcd, int *count);

// A box pointer read from a table in memory has no known address space, so every access through it is a FLAT instruction, which counts
// against lgkmcnt as well as vmcnt: a wave waiting for its LDS reads then also waits for every prefetch still in flight from HBM.
// Kernels that depend on that overlap address device memory through these types (global_load / global_store).
typedef double __attribute__((address_space(1))) gdouble;
typedef const gdouble *gcptr;
typedef gdouble *gptr;
__device__ __forceinline__ gcptr as_global(const double *p) { return (gcptr)p; }
__device__ __forceinline__ gptr as_global(double *p) { return (gptr)p; }
// load / store at a base that is the same for every lane (a scalar register pair) plus an unsigned 32-bit BYTE offset per lane: the
// "scalar base + vector offset" form of global_load / global_store, no 64-bit address arithmetic per lane
typedef const char __attribute__((address_space(1))) *gcbytes;
typedef char __attribute__((address_space(1))) *gbytes;
__device__ __forceinline__ double gld(gcptr base, unsigned byte_off) { return *(gcptr)((gcbytes)base + byte_off); }
__device__ __forceinline__ void gst(gptr base, unsigned byte_off, double v) { *(gptr)((gbytes)base + byte_off) = v; }
__device__ __forceinline__ void publish(ResultSlot *slot, double v, unsigned long long seq) {
  slot->value = v;
  __threadfence_system();
  __hip_atomic_store(&slot->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// pointer to cell (0,0,0) (first interior cell) of vector `id` in box `box`
__device__ __forceinline__ double *vec_origin(const hpgmg_hip_level &L, int box, int id) {
  return L.box_base[box] + (size_t)id * (size_t)L.volume + (size_t)L.ghosts * (size_t)(1 + L.jStride + L.kStride);
}


__device__ __forceinline__ gptr gvec_origin(const hpgmg_hip_level &L, int box, int id) { return as_global(vec_origin(L, box, id)); }   // the same, typed as device memory

// Ghost-free reading on levels whose boxes are all local (L.box_nbr codes >= 0 or -1): a cell up to `ghosts` cells outside box
// `box` is read where it LIVES -- in the interior of the neighbouring box -- instead of from this box's ghost zone, so no
// exchange_boundary copy is needed before a stencil launch.  Only directions that leave the DOMAIN stay in a ghost zone: that of the
// box reached through the in-domain directions, which apply_BCs fills from that box's interior with the same formula and inputs as the
// reference fills this box's copy of the cell.
struct GfColumn { int box, off; };   // a column (all k) of a box: off = i + j * jStride relative to its first interior cell
__device__ __forceinline__ GfColumn gf_column(const hpgmg_hip_level &L, int box, int gi, int gj) {
  if (gi < 0)           { const int n = L.box_nbr[6 * box + 0]; if (n >= 0) { box = n; gi += L.dim; } }
  else if (gi >= L.dim) { const int n = L.box_nbr[6 * box + 1]; if (n >= 0) { box = n; gi -= L.dim; } }
  if (gj < 0)           { const int n = L.box_nbr[6 * box + 2]; if (n >= 0) { box = n; gj += L.dim; } }
  else if (gj >= L.dim) { const int n = L.box_nbr[6 * box + 3]; if (n >= 0) { box = n; gj -= L.dim; } }
  return GfColumn{box, gi + gj * L.jStride};
}
// plane p (p < 0 or p >= dim) of that column
__device__ __forceinline__ double gf_load_outside(const hpgmg_hip_level &L, int id, GfColumn c, int p) {
  int box = c.box;
  if (p < 0) { const int n = L.box_nbr[6 * box + 4]; if (n >= 0) { box = n; p += L.dim; } }
  else       { const int n = L.box_nbr[6 * box + 5]; if (n >= 0) { box = n; p -= L.dim; } }
  return gvec_origin(L, box, id)[c.off + p * L.kStride];
}


// Fused forms of residual() for the tiled kernels (one cell per lane, TI x TJ lanes, marching in +k): the residual is not stored but
//   kind 1: reduced to its max-abs (norm(), misc.c:287-329) -- one partial per workgroup;
//   kind 2: restricted into the coarse level (restriction.c:54-57: the eight children in the order (i, i+1) of rows j, j+1 of plane k,
//           then of plane k+1; times 0.125).  A plane of residuals passes through an LDS tile; the lane of a child (even i, even j)
//           gathers its 2 x 2 patch one step later and carries the sum over the plane pair.
struct TileFused {
  int kind;                    // 0 plain store, 1 norm, 2 restrict
  double *partials;            // kind 1: [logical workgroup]
  hpgmg_hip_level Lc; int coarse_id; const int *map;   // kind 2: map[4 b .. 4 b + 3] = coarse box and coarse (i, j, k) under fine box b's first cell
};
template <int TI, int TJ>
struct TileFusedState {
  double lane_max = 0.0, racc = 0.0;
  // called once per plane step AFTER the step's first barrier with the residual of the PREVIOUS plane in sR[(k-1)&1] (k > k0)
  __device__ __forceinline__ void gather(const TileFused &F, const double *sR, int li, int lj, int kprev, int k0, gptr coarse) {
    if ((li | lj) & 1) return;
    const double *r = sR + ((kprev & 1) * TJ + lj) * TI + li;
    if (((kprev - k0) & 1) == 0) { racc = r[0] + r[1]; racc = racc + r[TI]; racc = racc + r[TI + 1]; }
    else { racc = racc + r[0]; racc = racc + r[1]; racc = racc + r[TI]; racc = racc + r[TI + 1]; coarse[((kprev - k0) >> 1) * F.Lc.kStride] = racc * 0.125; }
  }
};

}  // namespace hpgmg

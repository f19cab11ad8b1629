// brick_exchange.hip -- the exchange step of kernels/brick_visit.hip in isolation: G workgroups of 512 lanes (bricks of 8^3 cells), per iteration every
// workgroup publishes its six faces (6 x 64 records of 16 bytes, {tag, lo, hi, tag}, buffer_store ... sc1) and every lane t < 384 waits for ONE record of a
// neighbour.  What is varied is HOW the lane waits:
//   0  one poll in flight: load, wait, check, repeat (what record_wait does)
//   1  two polls in flight: the second load is issued before the first one is looked at
//   2  four polls in flight
//   3  as 0, without the s_sleep between polls
// and WHEN the record is published:  +0 after a barrier, from the LDS image (the kernel as it is); +4 by the lane that owns the cell, before the barrier.
// Time per exchange = launch time / iterations (a "sweep" of two barriers and a little arithmetic stands between exchanges).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned long long u64;
typedef unsigned __attribute__((ext_vector_type(4))) u4v;
struct alignas(16) Rec { unsigned tag0, lo, hi, tag1; };
constexpr int kAuxSc1 = 16;

struct Args { Rec *recs; u64 *fail; double *out; int side, iters, variant; size_t bytes; };

__device__ __forceinline__ void rec_store(__amdgpu_buffer_rsrc_t r, unsigned off, double v, unsigned tag) {
  const u64 bits = (u64)__double_as_longlong(v);
  u4v w; w.x = tag; w.y = (unsigned)bits; w.z = (unsigned)(bits >> 32); w.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(w, r, (int)off, 0, kAuxSc1);
}
__device__ __forceinline__ u4v rec_load(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, kAuxSc1); }
__device__ __forceinline__ bool ready(const u4v &w, unsigned tag) { return w.x == tag && w.w == tag; }
__device__ __forceinline__ double value(const u4v &w) { return __longlong_as_double((long long)(((u64)w.z << 32) | w.y)); }

template <int NAP>      // s_sleep units (64 clocks each) between polls; FIRST: also before the first poll
__device__ __forceinline__ double wait_for(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, u64 t0, bool &bad, bool first_nap) {
  if (first_nap) __builtin_amdgcn_s_sleep(NAP);
  u4v x = rec_load(r, off);
  while (!ready(x, tag)) {
    if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { bad = true; return 0.0; }
    __builtin_amdgcn_s_sleep(NAP);
    x = rec_load(r, off);
  }
  return value(x);
}

__global__ __launch_bounds__(512, 6) void bricks(const Args A) {
  __shared__ double imgs[2][1000];
  const int t = threadIdx.x, wg = blockIdx.x, side = A.side, G = side * side * side;
  const int bx = wg % side, by = (wg / side) % side, bz = wg / (side * side);
  const int li = t & 7, lj = (t >> 3) & 7, lk = t >> 6, pos = (li + 1) + 10 * (lj + 1) + 100 * (lk + 1);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)A.recs, 0, (int)A.bytes, 0x00020000);
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool bad = false;
  const bool own_publish = false, first_nap = (A.variant & 8) != 0;
  const int how = A.variant & 7;
  // the lane's face record (t < 384): face f, in-face cell (u, v); periodic grid of bricks: always six neighbours
  const int f = t >> 6, u = t & 7, v = (t >> 3) & 7;
  int fl = 0, fj = 0, fk = 0, hl = 0, hj = 0, hk = 0, nb = 0;
  if (t < 384) {
    const int w0 = (f & 1) ? 7 : 0, w1 = (f & 1) ? 8 : -1;
    if (f < 2) { fl = w0; fj = u; fk = v; hl = w1; hj = u; hk = v; } else if (f < 4) { fl = u; fj = w0; fk = v; hl = u; hj = w1; hk = v; } else { fl = u; fj = v; fk = w0; hl = u; hj = v; hk = w1; }
    int x = bx, y = by, z = bz; const int d = (f & 1) ? 1 : -1;
    if (f < 2) x = (x + d + side) % side; else if (f < 4) y = (y + d + side) % side; else z = (z + d + side) % side;
    nb = x + side * (y + side * z);
  }
  const int fpos = (fl + 1) + 10 * (fj + 1) + 100 * (fk + 1), hpos = (hl + 1) + 10 * (hj + 1) + 100 * (hk + 1);
  double acc = (double)(wg + 1) * 1e-3 + t * 1e-6;
  for (int z = t; z < 2000; z += 512) (&imgs[0][0])[z] = 0.0;
  __syncthreads();
  for (int it = 1; it <= A.iters; it++) {
    const int par = it & 1;
    const unsigned tag = (unsigned)it;
    const double *img = imgs[par ^ 1];
    double *dst = imgs[par];
    // the "sweep": a value per cell from the image
    acc = acc * 0.5 + 0.125 * (img[pos - 1] + img[pos + 1] + img[pos - 10] + img[pos + 10] + img[pos - 100] + img[pos + 100]);
    dst[pos] = acc;
    if (own_publish) {
      // the owner of a face cell stores it itself (up to three faces per cell), before the barrier
      const unsigned base = (unsigned)((((size_t)par * G + wg) * 6) * 64 * sizeof(Rec));
      if (li == 0) rec_store(rs, base + (unsigned)((0 * 64 + lk * 8 + lj) * sizeof(Rec)), acc, tag);
      if (li == 7) rec_store(rs, base + (unsigned)((1 * 64 + lk * 8 + lj) * sizeof(Rec)), acc, tag);
      if (lj == 0) rec_store(rs, base + (unsigned)((2 * 64 + lk * 8 + li) * sizeof(Rec)), acc, tag);
      if (lj == 7) rec_store(rs, base + (unsigned)((3 * 64 + lk * 8 + li) * sizeof(Rec)), acc, tag);
      if (lk == 0) rec_store(rs, base + (unsigned)((4 * 64 + lj * 8 + li) * sizeof(Rec)), acc, tag);
      if (lk == 7) rec_store(rs, base + (unsigned)((5 * 64 + lj * 8 + li) * sizeof(Rec)), acc, tag);
    } else {
      __syncthreads();
      if (t < 384) rec_store(rs, (unsigned)(((((size_t)par * G + wg) * 6 + f) * 64 + (t & 63)) * sizeof(Rec)), dst[fpos], tag);
    }
    if (t < 384) {
      const unsigned off = (unsigned)(((((size_t)par * G + nb) * 6 + (f ^ 1)) * 64 + (t & 63)) * sizeof(Rec));
      double got;
      switch (how) {
        case 0: got = wait_for<1>(rs, off, tag, t0, bad, first_nap); break;
        case 1: got = wait_for<2>(rs, off, tag, t0, bad, first_nap); break;
        case 2: got = wait_for<4>(rs, off, tag, t0, bad, first_nap); break;
        case 3: got = wait_for<8>(rs, off, tag, t0, bad, first_nap); break;
        case 4: got = wait_for<16>(rs, off, tag, t0, bad, first_nap); break;
        case 5: got = wait_for<32>(rs, off, tag, t0, bad, first_nap); break;
        default: got = wait_for<64>(rs, off, tag, t0, bad, first_nap); break;
      }
      dst[hpos] = got;
    }
    __syncthreads();
  }
  if (bad) atomicAdd(A.fail, 1ull);
  A.out[(size_t)wg * 512 + t] = acc;
}

int main() {
  const int sides[] = {2, 4, 8};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int side : sides) {
    const int G = side * side * side, iters = 400;
    Args A = {};
    A.side = side; A.iters = iters;
    A.bytes = (size_t)2 * G * 6 * 64 * sizeof(Rec);
    hipMalloc((void **)&A.recs, A.bytes); hipMalloc((void **)&A.fail, 8); hipMalloc((void **)&A.out, (size_t)G * 512 * 8);
    double ref = 0.0;
    for (int variant = 0; variant < 16; variant++) {
      if ((variant & 7) == 7) continue;
      A.variant = variant;
      float best = 1e30f; u64 fail = 0; double sum = 0.0;
      for (int rep = 0; rep < 5; rep++) {
        hipMemset(A.recs, 0, A.bytes); hipMemset(A.fail, 0, 8);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(bricks, dim3(G), dim3(512), 0, 0, A);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        hipMemcpy(&fail, A.fail, 8, hipMemcpyDeviceToHost);
        double *h = (double *)malloc((size_t)G * 512 * 8);
        hipMemcpy(h, A.out, (size_t)G * 512 * 8, hipMemcpyDeviceToHost);
        sum = 0.0; for (size_t i = 0; i < (size_t)G * 512; i++) sum += h[i];
        free(h);
        if (fail) break;
      }
      if (variant == 0) ref = sum;
      printf("G=%3d  nap %2d x 64 clocks between polls%s: %6.3f us per exchange + sweep   timeouts %llu  %s\n", G, 1 << (variant & 7), (variant & 8) ? " and before the first" : "                     ",
             best * 1e3 / iters, fail, sum == ref ? "same result" : "DIFFERENT RESULT");
    }
    hipFree(A.recs); hipFree(A.fail); hipFree(A.out);
  }
  return 0;
}

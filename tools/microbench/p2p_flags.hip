// p2p_flags.hip -- what a halo exchange BETWEEN WORKGROUPS of one launch costs: G workgroups of 1024 lanes stand for the 16^3 bricks of a small level
// (G = 8: the 32^3 level, G = 64: the 64^3 level, G = 27 / 125: odd grids); per iteration every workgroup publishes its six faces (6 x 256 cells) and
// waits for its six neighbours' faces.  Variants:
//   0  16-byte cells {value, sequence number}, written through (sc1) and polled by the lane that needs the value: ONE hop per exchange
//   1  8-byte values written through, then one flag per workgroup (agent-scope release / acquire as the compiler emits them): data -> flag -> data = three hops
//   2  cooperative-groups style: a central counter everybody adds to and polls (what grid.sync() does), values read after it
// and, for scale, the same traffic as one kernel launch per iteration.  A poll gives up after 50 ms (nothing here may hang the GPU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned long long u64;
struct alignas(16) Cell { double v; u64 seq; };
typedef unsigned __attribute__((ext_vector_type(4))) u4;

__device__ __forceinline__ void store16_sc1(Cell *p, double v, u64 seq) {
  u4 w; w.x = (unsigned)__double_as_longlong(v); w.y = (unsigned)(__double_as_longlong(v) >> 32); w.z = (unsigned)seq; w.w = (unsigned)(seq >> 32);
  // (s_nop: the data registers of a > 8-byte store must not be written by the VALU in the cycles after issue; the compiler cannot pad an asm store itself)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" :: "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ Cell load16_sc1(const Cell *p) {
  u4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  Cell c; c.v = __longlong_as_double((long long)(((u64)w.y << 32) | w.x)); c.seq = ((u64)w.w << 32) | w.z;
  return c;
}

struct Args { Cell *cells; double *vals; u64 *flags; u64 *counter; u64 *fail; double *out; int gx, gy, gz, iters, variant; };

__device__ __forceinline__ int nbr_of(const Args &A, int wg, int f) {       // periodic grid of workgroups: always six neighbours
  int x = wg % A.gx, y = (wg / A.gx) % A.gy, z = wg / (A.gx * A.gy);
  const int d = (f & 1) ? 1 : -1;
  if (f < 2) x = (x + d + A.gx) % A.gx; else if (f < 4) y = (y + d + A.gy) % A.gy; else z = (z + d + A.gz) % A.gz;
  return x + A.gx * (y + A.gy * z);
}

__global__ __launch_bounds__(1024) void cluster(const Args A) {
  const int wg = blockIdx.x, t = threadIdx.x, G = A.gx * A.gy * A.gz;
  __shared__ double acc_s[1024];
  double acc = (double)(wg + 1) * 1e-3 + t * 1e-6;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool bad = false;
  for (int it = 1; it <= A.iters; it++) {
    const int par = it & 1;
    for (int c = t; c < 1536; c += 1024) {
      const int f = c / 256, k = c % 256;
      const size_t slot = (((size_t)par * G + wg) * 6 + f) * 256 + k;
      const double v = acc + 1e-9 * c;
      if (A.variant == 0) store16_sc1(A.cells + slot, v, (u64)it);
      else __hip_atomic_store(A.vals + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (A.variant == 1) {
      __syncthreads();           // every lane's stores are issued ...
      if (t == 0) __hip_atomic_store(A.flags + wg, (u64)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // ... and performed before the flag (release)
      if (t < 6) {
        const int n = nbr_of(A, wg, t);
        while (__hip_atomic_load(A.flags + n, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (u64)it)
          if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull * 20) { bad = true; break; }
      }
      __syncthreads();
    } else if (A.variant == 2) {
      __threadfence();
      __syncthreads();
      if (t == 0) {
        __hip_atomic_fetch_add(A.counter, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(A.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (u64)it * (u64)G)
          if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull * 20) { bad = true; break; }
      }
      __syncthreads();
    }
    double got = 0.0;
    for (int c = t; c < 1536; c += 1024) {
      const int f = c / 256, k = c % 256, n = nbr_of(A, wg, f);
      const size_t slot = (((size_t)par * G + n) * 6 + (f ^ 1)) * 256 + k;
      if (A.variant == 0) {
        Cell x = load16_sc1(A.cells + slot);
        while (x.seq != (u64)it) {
          if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull * 20) { bad = true; break; }
          x = load16_sc1(A.cells + slot);
        }
        got += x.v;
      } else got += __hip_atomic_load(A.vals + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    acc_s[t] = got;
    __syncthreads();             // the "sweep": everybody sees what arrived (stands for the LDS halo being complete)
    acc = acc * 0.5 + acc_s[(t + 1) & 1023] * 0.25;
    __syncthreads();
  }
  if (bad) atomicAdd(A.fail, 1ull);
  A.out[(size_t)wg * 1024 + t] = acc;
}

// the same traffic, one launch per iteration
__global__ __launch_bounds__(1024) void one_step(const Args A, int it) {
  const int wg = blockIdx.x, t = threadIdx.x, G = A.gx * A.gy * A.gz, par = it & 1;
  __shared__ double acc_s[1024];
  double acc = A.out[(size_t)wg * 1024 + t];
  double got = 0.0;
  for (int c = t; c < 1536; c += 1024) {
    const int f = c / 256, k = c % 256, n = nbr_of(A, wg, f);
    got += A.vals[(((size_t)(par ^ 1) * G + n) * 6 + (f ^ 1)) * 256 + k];
  }
  acc_s[t] = got;
  __syncthreads();
  acc = acc * 0.5 + acc_s[(t + 1) & 1023] * 0.25;
  for (int c = t; c < 1536; c += 1024) A.vals[(((size_t)par * G + wg) * 6 + c / 256) * 256 + c % 256] = acc + 1e-9 * c;
  A.out[(size_t)wg * 1024 + t] = acc;
}

// host model of variants 0..2 (they compute the same thing)
static void model(int gx, int gy, int gz, int iters, std::vector<double> &out) {
  const int G = gx * gy * gz;
  std::vector<double> acc((size_t)G * 1024), nxt((size_t)G * 1024), got(1024);
  for (int wg = 0; wg < G; wg++) for (int t = 0; t < 1024; t++) acc[(size_t)wg * 1024 + t] = (double)(wg + 1) * 1e-3 + t * 1e-6;
  auto nbr = [&](int wg, int f) { int x = wg % gx, y = (wg / gx) % gy, z = wg / (gx * gy); const int d = (f & 1) ? 1 : -1;
    if (f < 2) x = (x + d + gx) % gx; else if (f < 4) y = (y + d + gy) % gy; else z = (z + d + gz) % gz; return x + gx * (y + gy * z); };
  for (int it = 1; it <= iters; it++) {
    for (int wg = 0; wg < G; wg++) {
      for (int t = 0; t < 1024; t++) {
        double g = 0.0;
        for (int c = t; c < 1536; c += 1024) { const int f = c / 256, n = nbr(wg, f); const int src_c = (f ^ 1) * 256 + c % 256; g += acc[(size_t)n * 1024 + (src_c % 1024)] + 1e-9 * src_c; }
        got[t] = g;
      }
      for (int t = 0; t < 1024; t++) nxt[(size_t)wg * 1024 + t] = acc[(size_t)wg * 1024 + t] * 0.5 + got[(t + 1) & 1023] * 0.25;
    }
    acc.swap(nxt);
  }
  out = acc;
}

int main() {
  const int grids[][3] = {{2, 2, 2}, {3, 3, 3}, {4, 4, 4}, {5, 5, 5}, {8, 8, 4}};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (auto &g : grids) {
    const int G = g[0] * g[1] * g[2], iters = 400;
    Args A = {};
    A.gx = g[0]; A.gy = g[1]; A.gz = g[2]; A.iters = iters;
    const size_t ncell = (size_t)2 * G * 6 * 256;
    hipMalloc((void **)&A.cells, ncell * sizeof(Cell)); hipMalloc((void **)&A.vals, ncell * 8); hipMalloc((void **)&A.flags, G * 8);
    hipMalloc((void **)&A.counter, 8); hipMalloc((void **)&A.fail, 8); hipMalloc((void **)&A.out, (size_t)G * 1024 * 8);
    std::vector<double> want, have((size_t)G * 1024);
    model(g[0], g[1], g[2], iters, want);
    for (int variant = 0; variant < 3; variant++) {
      A.variant = variant;
      float best = 1e30f; u64 fail = 0; size_t wrong = 0;
      for (int rep = 0; rep < 3; rep++) {
        hipMemset(A.cells, 0, ncell * sizeof(Cell)); hipMemset(A.vals, 0, ncell * 8); hipMemset(A.flags, 0, G * 8); hipMemset(A.counter, 0, 8); hipMemset(A.fail, 0, 8);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(cluster, dim3(G), dim3(1024), 0, 0, A);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        hipMemcpy(&fail, A.fail, 8, hipMemcpyDeviceToHost);
        hipMemcpy(have.data(), A.out, have.size() * 8, hipMemcpyDeviceToHost);
        wrong = 0;
        // the lane -> face-cell map of the model is the kernel's (value of cell c of workgroup n = acc of lane c % 1024 + 1e-9 c)
        for (size_t i = 0; i < have.size(); i++) if (have[i] != want[i]) wrong++;
        if (fail) break;
      }
      printf("G=%3d (%dx%dx%d) variant %d: %7.2f us per exchange   timeouts %llu  wrong values %zu of %zu\n", G, g[0], g[1], g[2], variant, best * 1e3 / iters, fail, wrong, have.size());
    }
    { // one launch per iteration
      hipMemset(A.vals, 0, ncell * 8); hipMemset(A.out, 0, (size_t)G * 1024 * 8);
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int it = 1; it <= iters; it++) hipLaunchKernelGGL(one_step, dim3(G), dim3(1024), 0, 0, A, it);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("G=%3d one launch per exchange: %7.2f us\n", G, ms * 1e3 / iters);
    }
    hipFree(A.cells); hipFree(A.vals); hipFree(A.flags); hipFree(A.counter); hipFree(A.fail); hipFree(A.out);
  }
  return 0;
}

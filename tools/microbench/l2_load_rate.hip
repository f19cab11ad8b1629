// l2_load_rate.hip -- how fast ONE CU (and the chip) can pull cache-resident data through vector loads, by load width and by waves per CU.
// hipcc --offload-arch=gfx950 -O3 -o l2_load_rate l2_load_rate.hip && ./l2_load_rate
// Each wave streams its own contiguous slice of a buffer that fits the L2 / MALL (repeatedly), with 8 independent loads in flight; the
// sum goes to a sink so nothing is optimised away.  Reported: GB/s per CU and for the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T>
__global__ void stream_kernel(const T *__restrict__ buf, size_t elems_per_wave, int reps, double *sink) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const T *p = buf + (size_t)wave * elems_per_wave;
  double acc = 0.0;
  for (int r = 0; r < reps; r++) {
    for (size_t o = lane; o + 7 * 64 < elems_per_wave; o += 8 * 64) {
      T v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = p[o + q * 64];
#pragma unroll
      for (int q = 0; q < 8; q++) acc += ((const double *)&v[q])[0];
    }
  }
  if (acc == 12345.678) sink[0] = acc;
}
struct alignas(16) d2 { double x, y; };
template <typename T> static void run(const char *name, int waves_per_cu, size_t total_bytes) {
  const int cus = 256, waves = cus * waves_per_cu, threads = 256, blocks = waves * 64 / threads;
  const size_t per_wave = total_bytes / waves / sizeof(T);
  T *buf; double *sink;
  hipMalloc(&buf, total_bytes); hipMemset(buf, 0, total_bytes); hipMalloc(&sink, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 40;
  hipLaunchKernelGGL(stream_kernel<T>, dim3(blocks), dim3(threads), 0, 0, buf, per_wave, 2, sink);
  hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(stream_kernel<T>, dim3(blocks), dim3(threads), 0, 0, buf, per_wave, reps, sink); hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double gbs = (double)per_wave * sizeof(T) * waves * reps / (ms * 1e-3) / 1e9;
  printf("%-10s %2d waves/CU  buffer %5.0f MB: %8.1f GB/s chip, %6.1f GB/s per CU\n", name, waves_per_cu, total_bytes / 1e6, gbs, gbs / cus);
  hipFree(buf); hipFree(sink);
}
int main() {
  for (size_t mb : {24, 160}) for (int w : {4, 8, 16, 32}) { run<double>("dwordx2", w, mb << 20); run<d2>("dwordx4", w, mb << 20); }
  return 0;
}

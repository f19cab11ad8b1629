// streams2.hip -- does the ~5.1 TB/s ceiling of a 9-read + 1-write kernel depend on WHERE the streams sit?
//  (a) stream s at base + s * (bytes + pad) for several pads (bank / channel aliasing between streams)
//  (b) row-interleaved layout: the 1 KiB rows of the 10 vectors alternate ([k][j][vector][i]), i.e. one wave touches 10 KiB contiguous
#include <hip/hip_runtime.h>
#include <stdio.h>
struct alignas(16) d2 { double x, y; };
template <int R>
__global__ __launch_bounds__(256) void sep(const d2 *base, d2 *out, size_t n2, size_t stride2) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += step) {
    d2 a = {0, 0};
#pragma unroll
    for (int s = 0; s < R; s++) { const d2 v = base[s * stride2 + i]; a.x += v.x; a.y += v.y; }
    out[i] = a;
  }
}
// rows of 64 d2 (1 KiB); row r of vector s at ((r * (R + 1)) + s) * 64
template <int R>
__global__ __launch_bounds__(256) void inter(d2 *buf, size_t rows) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, waves = (size_t)gridDim.x * blockDim.x / 64;
  const int lane = threadIdx.x % 64;
  for (size_t r = wave; r < rows; r += waves) {
    d2 *row = buf + r * (R + 1) * 64;
    d2 a = {0, 0};
#pragma unroll
    for (int s = 0; s < R; s++) { const d2 v = row[s * 64 + lane]; a.x += v.x; a.y += v.y; }
    row[R * 64 + lane] = a;
  }
}
int main() {
  const size_t n2 = (size_t)8 * 2230800 / 2;        // 16-byte elements per stream (143 MB)
  const size_t pads[] = {0, 16, 256, 4096, 8192 + 256, 65536 + 512, (1 << 20) + 4096 + 256, (size_t)3 << 20};
  d2 *big; hipMalloc((void **)&big, (n2 * 16 + ((size_t)4 << 20)) * 11); hipMemset(big, 0, (n2 * 16 + ((size_t)4 << 20)) * 11);
  d2 *out; hipMalloc((void **)&out, n2 * 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (size_t pad : pads) {
    const size_t stride2 = n2 + pad / 16;
    for (int g : {2048, 32768}) {
      for (int it = 0; it < 2; it++) hipLaunchKernelGGL((sep<9>), dim3(g), dim3(256), 0, 0, big, out, n2, stride2);
      hipEventRecord(a);
      for (int it = 0; it < 10; it++) hipLaunchKernelGGL((sep<9>), dim3(g), dim3(256), 0, 0, big, out, n2, stride2);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("separate 9r1w pad=%9zu B grid=%6d : %7.1f us %6.2f TB/s\n", pad, g, ms / 10 * 1e3, 10.0 * n2 * 16 / (ms / 10 * 1e-3) / 1e12);
    }
  }
  const size_t rows = n2 / 64;
  for (int g : {1024, 2048, 8192, 32768}) {
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((inter<9>), dim3(g), dim3(256), 0, 0, big, rows);
    hipEventRecord(a);
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL((inter<9>), dim3(g), dim3(256), 0, 0, big, rows);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("row-interleaved 9r1w grid=%6d : %7.1f us %6.2f TB/s\n", g, ms / 10 * 1e3, 10.0 * rows * 64 * 16 / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}

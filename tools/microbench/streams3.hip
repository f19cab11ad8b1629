// streams3.hip -- HBM ceiling with a working set far beyond the 256 MB infinity cache: one 1.43 GB stream read / copied
#include <hip/hip_runtime.h>
#include <stdio.h>
struct alignas(16) d2 { double x, y; };
template <int MODE, int U>   // 0 read, 1 copy, 2 write
__global__ __launch_bounds__(256) void k(const d2 *in, d2 *out, size_t n2) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  d2 a = {0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += step * U) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t j = i + u * step;
      if (j < n2) { if (MODE == 2) out[j] = d2{1.0, 2.0}; else { const d2 v = in[j]; if (MODE == 1) out[j] = v; else { a.x += v.x; a.y += v.y; } } }
    }
  }
  if (MODE == 0 && a.x == 1.2345e300) out[0] = a;
}
template <int MODE, int U> void run(const d2 *in, d2 *out, size_t n2, int g, const char *tag, double streams) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<MODE, U>), dim3(g), dim3(256), 0, 0, in, out, n2);
  hipEventRecord(a);
  for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k<MODE, U>), dim3(g), dim3(256), 0, 0, in, out, n2);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-6s unroll=%d grid=%6d : %8.1f us %6.2f TB/s\n", tag, U, g, ms / 5 * 1e3, streams * n2 * 16 / (ms / 5 * 1e-3) / 1e12);
}
int main() {
  const size_t n2 = (size_t)10 * 8 * 2230800 / 2;   // 1.43 GB
  d2 *in, *out; hipMalloc((void **)&in, n2 * 16); hipMalloc((void **)&out, n2 * 16); hipMemset(in, 0, n2 * 16); hipMemset(out, 0, n2 * 16);
  for (int g : {2048, 8192, 65536}) {
    run<0, 4>(in, out, n2, g, "read", 1); run<0, 8>(in, out, n2, g, "read", 1);
    run<1, 4>(in, out, n2, g, "copy", 2); run<2, 4>(in, out, n2, g, "write", 1);
  }
  return 0;
}

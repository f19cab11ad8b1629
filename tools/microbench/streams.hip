// streams.hip -- what HBM bandwidth does an MI355X deliver for R read streams + W write streams of 134 MB each
// (the shape of one smoother sweep: 8-9 reads, 1-2 writes), as a function of access width, loads in flight per lane and
// grid size?  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/streams.hip -o gpurun_out/streams ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
struct alignas(16) d2 { double x, y; };
struct Ptrs { const d2 *r[10]; d2 *w[2]; };
template <int R, int W, int UNROLL>
__global__ __launch_bounds__(256) void k(Ptrs P, size_t n2) {   // n2 = number of 16-byte elements per stream
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride * UNROLL) {
    d2 acc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { acc[u].x = 0; acc[u].y = 0; }
#pragma unroll
    for (int s = 0; s < R; s++) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) { const size_t j = i + u * stride; if (j < n2) { const d2 v = P.r[s][j]; acc[u].x += v.x; acc[u].y += v.y; } }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { const size_t j = i + u * stride; if (j < n2) { for (int s = 0; s < W; s++) P.w[s][j] = acc[u]; if (W == 0 && acc[u].x == 1.2345e300) P.w[0][0] = acc[u]; } }
  }
}
template <int R, int W, int U> static void run(Ptrs P, size_t n2, int grid, const char *tag) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL((k<R, W, U>), dim3(grid), dim3(256), 0, 0, P, n2);
  hipEventRecord(a);
  const int reps = 10;
  for (int it = 0; it < reps; it++) hipLaunchKernelGGL((k<R, W, U>), dim3(grid), dim3(256), 0, 0, P, n2);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)(R + W) * n2 * 16.0;
  printf("%-10s R=%d W=%d unroll=%d grid=%6d : %7.1f us  %6.2f TB/s\n", tag, R, W, U, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}
int main() {
  const size_t n2 = (size_t)8 * 2230800 / 2 * 1;   // one vector of the 256^3 level incl. padding: 8 boxes x 2,230,800 doubles
  Ptrs P;
  for (int s = 0; s < 10; s++) { void *p; hipMalloc(&p, n2 * 16); hipMemset(p, 0, n2 * 16); P.r[s] = (const d2 *)p; }
  for (int s = 0; s < 2; s++) { void *p; hipMalloc(&p, n2 * 16); P.w[s] = (d2 *)p; }
  hipDeviceSynchronize();
  const int grids[] = {1024, 2048, 4096, 8192, 32768};
  for (int g : grids) {
    run<1, 0, 4>(P, n2, g, "read1");
    run<1, 1, 4>(P, n2, g, "copy");
    run<4, 1, 2>(P, n2, g, "4r1w");
    run<8, 1, 1>(P, n2, g, "8r1w");
    run<8, 1, 2>(P, n2, g, "8r1w");
    run<9, 1, 1>(P, n2, g, "9r1w");
    run<8, 2, 1>(P, n2, g, "8r2w");
    run<8, 2, 2>(P, n2, g, "8r2w");
  }
  return 0;
}

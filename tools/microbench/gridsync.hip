// gridsync.hip -- cost of a device-wide barrier (cooperative groups grid.sync) vs a kernel boundary for G workgroups of 1024 lanes
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(1024) void coop(double *buf, int iters) {
  cg::grid_group g = cg::this_grid();
  const size_t me = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
  double v = buf[me];
  for (int it = 0; it < iters; it++) {
    buf[me] = v + 1.0;
    g.sync();
    v = buf[(me + 1024) % n] * 0.5;       // read what another workgroup wrote
  }
  buf[me] = v;
}
__global__ __launch_bounds__(1024) void one(double *buf) {
  const size_t me = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
  buf[me] = buf[(me + 1024) % n] * 0.5 + 1.0;
}
int main() {
  double *buf; hipMalloc((void **)&buf, 256 * 1024 * 8); hipMemset(buf, 0, 256 * 1024 * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int G : {8, 64, 256}) {
    int iters = 200;
    void *args[] = {&buf, &iters};
    hipLaunchCooperativeKernel((const void *)coop, dim3(G), dim3(1024), args, 0, 0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipError_t e = hipLaunchCooperativeKernel((const void *)coop, dim3(G), dim3(1024), args, 0, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("G=%3d cooperative: %s  %.2f us per sync\n", G, hipGetErrorString(e), ms * 1e3 / iters);
    hipEventRecord(a);
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(one, dim3(G), dim3(1024), 0, 0, buf);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("G=%3d kernel boundary:     %.2f us per launch\n", G, ms * 1e3 / iters);
  }
  return 0;
}

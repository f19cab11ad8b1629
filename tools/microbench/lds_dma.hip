// lds_dma.hip -- does __builtin_amdgcn_global_load_lds(..., 16, ...) (global_load_lds_dwordx4, gfx950) do what the sweep-pair kernel needs of it?  Each lane of a
// wave hands in its OWN global address (16 bytes); the data must land at (wave-uniform LDS base) + lane * 16 and be there after s_waitcnt vmcnt(0); several
// such loads into different rows must be independent.  Prints the number of 16-byte slots that differ from what plain loads give (0: yes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(1))) const void *gvoid;
typedef __attribute__((address_space(3))) void *lvoid;
struct alignas(16) p2 { double x, y; };
__global__ __launch_bounds__(640) void k(const double *src, double *dst, int stride, int rows, int nop) {
  extern __shared__ p2 lds[];
  const int lane = threadIdx.x, w = threadIdx.y;
  p2 (*row)[8][64] = reinterpret_cast<p2 (*)[8][64]>(lds);
  for (int r = 0; r < rows; r++) {
    const double *p = src + (size_t)(w * rows + r) * stride + 2 * ((lane * 7 + r) % 64);
    __builtin_amdgcn_global_load_lds((gvoid)p, (lvoid)&row[w][r][0], 16, 0, 0);
    if (nop) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  for (int r = 0; r < rows; r++) { const p2 v = row[w][r][lane]; dst[((size_t)(w * rows + r) * 64 + lane) * 2] = v.x; dst[((size_t)(w * rows + r) * 64 + lane) * 2 + 1] = v.y; }
}
int main() {
  const int NW = 10, rows = 8, stride = 160;
  std::vector<double> h((size_t)NW * rows * stride);
  for (size_t i = 0; i < h.size(); i++) h[i] = (double)i;
  double *d, *o; hipMalloc((void **)&d, h.size() * 8); hipMalloc((void **)&o, NW * rows * 64 * 16);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 8 * 64 * 16);
  for (int nop = 0; nop < 2; nop++) {
    hipMemset(o, 0, NW * rows * 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64, NW), NW * 8 * 64 * 16, 0, d, o, stride, rows, nop);
    std::vector<double> got((size_t)NW * rows * 64 * 2); hipMemcpy(got.data(), o, got.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0, shown = 0;
    for (int w = 0; w < NW; w++) for (int r = 0; r < rows; r++) for (int lane = 0; lane < 64; lane++) {
      const size_t src = (size_t)(w * rows + r) * stride + 2 * ((lane * 7 + r) % 64);
      const double gx = got[((size_t)(w * rows + r) * 64 + lane) * 2], gy = got[((size_t)(w * rows + r) * 64 + lane) * 2 + 1];
      if (gx != (double)src || gy != (double)(src + 1)) { bad++; if (shown++ < 12) printf("nop %d w %d r %d lane %d: got %.0f %.0f want %zu %zu\n", nop, w, r, lane, gx, gy, src, src + 1); }
    }
    printf("nop %d: %d bad of %d\n", nop, bad, NW * rows * 64);
  }
  return 0;
}

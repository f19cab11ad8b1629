// ipc_probe3.hip -- when does hipStreamWaitEvent on an interprocess event fail?  Child (opener) records behind queued work and raises a flag in
// shared memory at once; parent (creator) spins on the flag and waits for the event immediately, optionally with work queued on its own stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <atomic>
__global__ void spin_kernel(long long cycles) { const long long t0 = clock64(); while (clock64() - t0 < cycles) { } }
struct Shared { std::atomic<int> handle_ready, recorded, done; hipIpcEventHandle_t h; };
#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("[%s] %s -> %s\n", who, #c, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
int main(int argc, char **argv) {
  const long long child_spin = argc > 1 ? atoll(argv[1]) : 0, parent_spin = argc > 2 ? atoll(argv[2]) : 0;
  const int child_kernels = argc > 3 ? atoi(argv[3]) : 1;
  Shared *S = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  new (S) Shared();
  pid_t pid = fork();
  if (pid == 0) {
    const char *who = "child";
    hipStream_t s; hipEvent_t ev;
    CK(hipSetDevice(0)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    while (!S->handle_ready.load()) { }
    CK(hipIpcOpenEventHandle(&ev, S->h));
    for (int r = 0; r < 3; r++) {
      for (int q = 0; q < child_kernels; q++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, child_spin);
      CK(hipEventRecord(ev, s));
      S->recorded.store(r + 1);
      while (S->done.load() < r + 1) { }
    }
    CK(hipStreamSynchronize(s));
    return 0;
  }
  const char *who = "parent";
  hipStream_t s; hipEvent_t ev;
  CK(hipSetDevice(0)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
  CK(hipIpcGetEventHandle(&S->h, ev));
  S->handle_ready.store(1);
  for (int r = 0; r < 3; r++) {
    if (parent_spin) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, parent_spin);
    while (S->recorded.load() < r + 1) { }
    hipError_t e = hipStreamWaitEvent(s, ev, 0);
    int tries = 0;
    while (e != hipSuccess && tries < 1000) { (void)hipGetLastError(); usleep(100); e = hipStreamWaitEvent(s, ev, 0); tries++; }
    printf("[parent] round %d: hipStreamWaitEvent -> %s after %d retries\n", r, hipGetErrorString(e), tries); fflush(stdout);
    hipStreamSynchronize(s);
    S->done.store(r + 1);
  }
  int st; waitpid(pid, &st, 0);
  printf("child exit %d\n", WEXITSTATUS(st));
  return 0;
}

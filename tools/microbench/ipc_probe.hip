// ipc_probe.hip -- can two processes on ONE MI355X share device memory and events?  (what a node-local peer-to-peer transport needs)
// parent: allocates, exports a memory handle and an interprocess event handle; child: opens both, copies into the parent's buffer on its own
// stream, records the event; parent: waits on the event in stream order, checks the data.  Prints what works.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/wait.h>
__global__ void spin_kernel(long long cycles, double *p) { const long long t0 = clock64(); while (clock64() - t0 < cycles) { } if (p) p[0] = 1.0; }
#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("[%s] %s -> %s\n", who, #c, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
struct Msg { hipIpcMemHandle_t mem; hipIpcEventHandle_t ev; };
int main() {
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 2;
  const size_t n = 1 << 20;
  pid_t pid = fork();                       // before any HIP call
  if (pid == 0) {
    const char *who = "child";
    Msg m; if (read(p2c[0], &m, sizeof m) != (ssize_t)sizeof m) return 3;
    double *peer = nullptr; hipEvent_t ev; hipStream_t s;
    CK(hipSetDevice(0));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipIpcOpenMemHandle((void **)&peer, m.mem, hipIpcMemLazyEnablePeerAccess));
    hipError_t ee = hipIpcOpenEventHandle(&ev, m.ev);
    printf("[child] hipIpcOpenEventHandle -> %s\n", hipGetErrorString(ee));
    double *mine; CK(hipMalloc((void **)&mine, n * sizeof(double)));
    double *h = (double *)malloc(n * sizeof(double)); for (size_t i = 0; i < n; i++) h[i] = 0.5 * (double)i;
    CK(hipMemcpy(mine, h, n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, (long long)(getenv("PROBE_SPIN") ? atoll(getenv("PROBE_SPIN")) : 0), (double *)nullptr);   // the copy and the record sit behind this
    CK(hipMemcpyAsync(peer, mine, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (ee == hipSuccess) { hipError_t er = hipEventRecord(ev, s); printf("[child] hipEventRecord(ipc event) -> %s\n", hipGetErrorString(er)); }
    char ok = (ee == hipSuccess) ? 'E' : 'S';
    if (ok == 'S') CK(hipStreamSynchronize(s));
    if (write(c2p[1], &ok, 1) != 1) return 3;
    CK(hipStreamSynchronize(s));
    char bye; if (read(p2c[0], &bye, 1) != 1) return 3;      // keep the mapping alive until the parent has checked
    CK(hipIpcCloseMemHandle(peer));
    fflush(stdout);
    return 0;
  }
  const char *who = "parent";
  double *buf; hipEvent_t ev; hipStream_t s; Msg m;
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipMalloc((void **)&buf, n * sizeof(double)));
  CK(hipMemset(buf, 0, n * sizeof(double)));
  CK(hipDeviceSynchronize());
  CK(hipIpcGetMemHandle(&m.mem, buf));
  hipError_t e1 = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess);
  printf("[parent] hipEventCreateWithFlags(interprocess) -> %s\n", hipGetErrorString(e1));
  hipError_t e2 = (e1 == hipSuccess) ? hipIpcGetEventHandle(&m.ev, ev) : e1;
  printf("[parent] hipIpcGetEventHandle -> %s\n", hipGetErrorString(e2));
  if (write(p2c[1], &m, sizeof m) != (ssize_t)sizeof m) return 3;
  char how; if (read(c2p[0], &how, 1) != 1) return 3;
  if (how == 'E') { hipError_t ew = hipStreamWaitEvent(s, ev, 0); printf("[parent] hipStreamWaitEvent(ipc event) -> %s\n", hipGetErrorString(ew)); }
  double *h = (double *)malloc(n * sizeof(double));
  CK(hipMemcpyAsync(h, buf, n * sizeof(double), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  size_t bad = 0; for (size_t i = 0; i < n; i++) if (h[i] != 0.5 * (double)i) bad++;
  printf("[parent] peer copy through hipIpc memory handle: %zu of %zu values wrong; ordering by %s\n", bad, n, how == 'E' ? "interprocess EVENT" : "sender-side stream synchronise");
  char bye = 'x'; if (write(p2c[1], &bye, 1) != 1) return 3;
  int st; waitpid(pid, &st, 0);
  printf("[parent] child exit %d\n", WEXITSTATUS(st));
  return bad ? 1 : 0;
}

// ipc_probe2.hip -- symmetric version: BOTH processes create interprocess events (several), export them, open the peer's and record those; each
// then waits for its OWN events (recorded by the peer).  Mirrors what kernels/comm_ipc.hip does.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/wait.h>
#define CK(c) do { hipError_t e_ = (c); printf("[%s] %s -> %s\n", who, #c, hipGetErrorString(e_)); fflush(stdout); if (e_ != hipSuccess) return 1; } while (0)
int run(const char *who, int rfd, int wfd) {
  hipStream_t s; hipEvent_t mine[4], theirs[4]; hipIpcEventHandle_t hm[4], ht[4];
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int i = 0; i < 4; i++) { CK(hipEventCreateWithFlags(&mine[i], hipEventDisableTiming | hipEventInterprocess)); CK(hipIpcGetEventHandle(&hm[i], mine[i])); }
  if (write(wfd, hm, sizeof hm) != (ssize_t)sizeof hm) return 3;
  if (read(rfd, ht, sizeof ht) != (ssize_t)sizeof ht) return 3;
  for (int i = 0; i < 4; i++) CK(hipIpcOpenEventHandle(&theirs[i], ht[i]));
  for (int i = 0; i < 4; i++) CK(hipEventRecord(theirs[i], s));
  char c = 'x'; if (write(wfd, &c, 1) != 1) return 3;
  if (read(rfd, &c, 1) != 1) return 3;
  for (int i = 0; i < 4; i++) CK(hipStreamWaitEvent(s, mine[i], 0));
  CK(hipStreamSynchronize(s));
  if (write(wfd, &c, 1) != 1) return 3;
  if (read(rfd, &c, 1) != 1) return 3;
  return 0;
}
int main() {
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 2;
  pid_t pid = fork();
  if (pid == 0) return run("child", p2c[0], c2p[1]);
  int r = run("parent", c2p[0], p2c[1]);
  int st; waitpid(pid, &st, 0);
  printf("parent %d child %d\n", r, WEXITSTATUS(st));
  return r;
}

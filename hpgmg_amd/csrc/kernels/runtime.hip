// runtime.hip -- device memory, stream and event hooks behind include/hpgmg_hip.h.
// Vector storage of every level is device memory obtained here; this is the
// replacement of MALLOC()/FREE() in the reference's level.c:25-40.
#include <stdio.h>
#include <string.h>
#include "common.hpp"

namespace hpgmg {
hipStream_t g_stream = nullptr;
static char g_last_error[256] = "";
int record_error(hipError_t e, const char *where) {
  snprintf(g_last_error, sizeof g_last_error, "%s: %s", where, hipGetErrorString(e));
  fprintf(stderr, "hpgmg_hip: %s\n", g_last_error);
  return (int)e;
}
}  // namespace hpgmg
using namespace hpgmg;

extern "C" {
int hpgmg_hip_graph_flush(void);

int hpgmg_hip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
// The launch stream.  Unless the caller chose one, the library creates its own on first use: the legacy null stream
// cannot be captured into a hipGraph (hipErrorStreamCaptureUnsupported), and it would serialise against every other
// blocking stream of the process (e.g. a framework's).
static bool g_stream_chosen = false;
static void ensure_stream() {
  if (g_stream_chosen) return;
  g_stream_chosen = true;
  hipStream_t s = nullptr;
  (void)hipDeviceSynchronize();
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) g_stream = s; else (void)hipGetLastError();
}
int hpgmg_hip_set_device(int dev) { HPGMG_CHECK(hipSetDevice(dev)); ensure_stream(); return 0; }
void hpgmg_hip_set_stream(void *s) { g_stream = (hipStream_t)s; g_stream_chosen = true; }
void *hpgmg_hip_get_stream(void) { ensure_stream(); return (void *)g_stream; }
int hpgmg_hip_sync(void) { hpgmg_hip_graph_flush(); HPGMG_CHECK(hipStreamSynchronize(g_stream)); return 0; }
const char *hpgmg_hip_last_error(void) { return g_last_error; }

void *hpgmg_hip_malloc(size_t bytes) {
  ensure_stream();
  void *p = nullptr;
  hpgmg_hip_graph_flush();
  if (bytes == 0) bytes = 8;
  if (hipMalloc(&p, bytes) != hipSuccess) { record_error(hipGetLastError(), "hipMalloc"); return nullptr; }
  if (hipMemsetAsync(p, 0, bytes, g_stream) != hipSuccess) { record_error(hipGetLastError(), "hipMemsetAsync"); }
  return p;
}
void hpgmg_hip_free(void *p) { hpgmg_hip_graph_flush(); if (p) { hipStreamSynchronize(g_stream); (void)hipFree(p); } }
void *hpgmg_hip_host_malloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) { record_error(hipErrorOutOfMemory, "hpgmg_hip_host_malloc"); return nullptr; }
  memset(p, 0, bytes);
  return p;
}
void hpgmg_hip_host_free(void *p) { if (p) { hpgmg_hip_graph_flush(); hipStreamSynchronize(g_stream); (void)hipHostFree(p); } }
int hpgmg_hip_memcpy_h2d(void *d, const void *s, size_t n) { hpgmg_hip_graph_flush(); HPGMG_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, g_stream)); HPGMG_CHECK(hipStreamSynchronize(g_stream)); return 0; }
int hpgmg_hip_memcpy_d2h(void *d, const void *s, size_t n) { hpgmg_hip_graph_flush(); HPGMG_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, g_stream)); HPGMG_CHECK(hipStreamSynchronize(g_stream)); return 0; }
int hpgmg_hip_memcpy_d2d(void *d, const void *s, size_t n) { hpgmg_hip_graph_flush(); HPGMG_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, g_stream)); return 0; }
int hpgmg_hip_memset0(void *d, size_t n) { hpgmg_hip_graph_flush(); HPGMG_CHECK(hipMemsetAsync(d, 0, n, g_stream)); return 0; }

void *hpgmg_hip_stream_create(void) { hipStream_t s; if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr; return (void *)s; }
void hpgmg_hip_stream_destroy(void *s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
int hpgmg_hip_stream_wait_event(void *e) { HPGMG_CHECK(hipStreamWaitEvent(g_stream, (hipEvent_t)e, 0)); return 0; }
void *hpgmg_hip_event_create(void) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; return (void *)e; }
void hpgmg_hip_event_destroy(void *e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
int hpgmg_hip_event_record(void *e) { HPGMG_CHECK(hipEventRecord((hipEvent_t)e, g_stream)); return 0; }
double hpgmg_hip_event_elapsed_ms(void *a, void *b) {
  float ms = 0.f;
  if (hipEventSynchronize((hipEvent_t)b) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.0;
  return (double)ms;
}

}  // extern "C"

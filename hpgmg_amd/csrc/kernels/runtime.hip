// runtime.hip -- device memory, stream and event hooks behind include/hpgmg_hip.h.
// Vector storage of every level is device memory obtained here; this is the
// replacement of MALLOC()/FREE() in the reference's level.c:25-40.
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <dlfcn.h>
#include <vector>
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
// Every cache in this library that is keyed on a device address is told when that address goes away: a later allocation may land on the same
// address with other content (the tile orders of two-part launches keyed on a neighbour table, the hipIpc handles / peer mappings of the
// node-local transport keyed on an allocation).
void hpgmg_hip_tile_part_forget(const void *p);
void hpgmg_hip_ipc_forget(const void *p);
void hpgmg_hip_free(void *p) {
  hpgmg_hip_graph_flush();
  if (!p) return;
  hipStreamSynchronize(g_stream);
  hpgmg_hip_tile_part_forget(p);
  hpgmg_hip_ipc_forget(p);
  (void)hipFree(p);
}
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


// ---- device-time attribution (reference level.h:162-196, mg.c:54-161: the per-level timing table) ----------------
// Launches are asynchronous, so host clocks around them say nothing about where the DEVICE spent its time.  A timer here is
// a hipEvent pair recorded on the launch stream around an operator; the elapsed device time is added to the caller's
// accumulator (seconds) when the pool is flushed -- at the latest when it is full, normally when the table is printed or
// reset.  Inside a hipGraph capture nothing is recorded (-1).
// A slot is free, open (begin recorded) or closed (end recorded); only closed slots are settled, so a flush that happens while outer
// ticks are still open (MGVCycle's total around its operators) leaves those alone, and a slot id carries the slot's generation, so an
// id that outlived its slot is ignored instead of closing somebody else's tick.
static const int kTimerSettleAt = 8192;          // closed pairs that may pile up before timer_begin settles them
enum { T_FREE = 0, T_OPEN = 1, T_CLOSED = 2 };
struct TimerSlot { hipEvent_t ev[2]; double *acc; int state; unsigned gen; bool made; };
static TimerSlot *g_tslot = nullptr;
static int g_tcap = 0, g_tn = 0, g_tclosed = 0;   // capacity, slots ever used, closed and not yet settled
static int *g_tfree = nullptr; static int g_tfree_n = 0;
int hpgmg_hip_graph_is_open(void);
int hpgmg_hip_timer_flush(void) {
  for (int p = 0; p < g_tn; p++) {
    TimerSlot &t = g_tslot[p];
    if (t.state != T_CLOSED) continue;
    float ms = 0.f;
    if (hipEventSynchronize(t.ev[1]) == hipSuccess && hipEventElapsedTime(&ms, t.ev[0], t.ev[1]) == hipSuccess && t.acc) *t.acc += 1e-3 * (double)ms;
    else (void)hipGetLastError();
    t.state = T_FREE; t.gen++; t.acc = nullptr;
    g_tfree[g_tfree_n++] = p;
  }
  g_tclosed = 0;
  return 0;
}
int hpgmg_hip_timer_begin(double *acc_seconds) {
  if (hpgmg_hip_graph_is_open()) return -1;
  ensure_stream();
  if (g_tclosed >= kTimerSettleAt) hpgmg_hip_timer_flush();
  int p;
  if (g_tfree_n > 0) p = g_tfree[--g_tfree_n];
  else {
    if (g_tn == g_tcap) {
      const int cap = g_tcap ? 2 * g_tcap : 1024;
      TimerSlot *ns = (TimerSlot *)realloc(g_tslot, (size_t)cap * sizeof(TimerSlot));
      int *nf = (int *)realloc(g_tfree, (size_t)cap * sizeof(int));
      if (ns) g_tslot = ns;
      if (nf) g_tfree = nf;
      if (!ns || !nf) return -1;
      for (int q = g_tcap; q < cap; q++) { g_tslot[q].state = T_FREE; g_tslot[q].gen = 0; g_tslot[q].made = false; g_tslot[q].acc = nullptr; }
      g_tcap = cap;
    }
    p = g_tn++;
  }
  TimerSlot &t = g_tslot[p];
  if (!t.made) {
    if (hipEventCreate(&t.ev[0]) != hipSuccess || hipEventCreate(&t.ev[1]) != hipSuccess) { (void)hipGetLastError(); g_tfree[g_tfree_n++] = p; return -1; }
    t.made = true;
  }
  t.acc = acc_seconds; t.state = T_OPEN;
  (void)hipEventRecord(t.ev[0], g_stream);
  return (int)(((t.gen & 0x7ffu) << 20) | (unsigned)p);          // 20 bits of slot, 11 of generation
}
void hpgmg_hip_timer_end(int id) {
  if (id < 0) return;
  const int p = id & 0xfffff;
  if (p >= g_tn) return;
  TimerSlot &t = g_tslot[p];
  if (t.state != T_OPEN || (t.gen & 0x7ffu) != (((unsigned)id >> 20) & 0x7ffu)) return;       // a stale id
  if (hipEventRecord(t.ev[1], g_stream) != hipSuccess) { (void)hipGetLastError(); t.acc = nullptr; }
  t.state = T_CLOSED; g_tclosed++;
}
// an accumulator is going away (level destroyed): settle what is pending; ticks still open on it are dropped
void hpgmg_hip_timer_forget(const void *lo, const void *hi) {
  bool any = false;
  for (int p = 0; p < g_tn; p++) {
    TimerSlot &t = g_tslot[p];
    if (t.state == T_FREE || (const void *)t.acc < lo || (const void *)t.acc >= hi) continue;
    if (t.state == T_OPEN) t.acc = nullptr; else any = true;
  }
  if (any) hpgmg_hip_timer_flush();
}

// ---- roctx ranges (per level / operator markers for rocprofv3 --marker-trace), resolved at run time so the
// library has no hard dependency on the profiler's marker library; HPGMG_ROCTX=1 turns them on ----
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)(void);
static roctx_push_t g_roctx_push = nullptr;
static roctx_pop_t g_roctx_pop = nullptr;
static int g_roctx_state = -1;   // -1 unknown, 0 off, 1 on
static int roctx_ready() {
  if (g_roctx_state < 0) {
    g_roctx_state = 0;
    const char *e = getenv("HPGMG_ROCTX");
    if (e && e[0] == '1') {
      void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (h) { g_roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA"); g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop"); }
      if (g_roctx_push && g_roctx_pop) g_roctx_state = 1;
      else fprintf(stderr, "hpgmg_hip: HPGMG_ROCTX=1 but no roctx library could be loaded\n");
    }
  }
  return g_roctx_state;
}
int hpgmg_hip_range_enabled(void) { return roctx_ready(); }
void hpgmg_hip_range_push(const char *name) { if (roctx_ready()) g_roctx_push(name); }
void hpgmg_hip_range_pop(void) { if (roctx_ready()) g_roctx_pop(); }

}  // extern "C"

// ---- two-part launches of the tiled kernels (common.hpp) ----
namespace hpgmg {
int g_tile_part = 0;
struct PartOrder { const int *nbr; int num_boxes, ti, tj, ck, part, walls; int *d_order; int grid, per_xcd, count; };
static std::vector<PartOrder> g_part_orders;
const int *tile_part_order(const hpgmg_hip_level *L, int tiles_i, int tiles_j, int chunks_k, int part, bool walls_to_part2, int *grid, int *per_xcd, int *count) {
  for (const PartOrder &o : g_part_orders)
    if (o.nbr == L->box_nbr && o.num_boxes == L->num_boxes && o.ti == tiles_i && o.tj == tiles_j && o.ck == chunks_k && o.part == part && o.walls == (int)walls_to_part2) {
      *grid = o.grid; *per_xcd = o.per_xcd; *count = o.count; return o.d_order;
    }
  const int total = L->num_boxes * chunks_k * tiles_j * tiles_i;
  std::vector<int> nbr(6 * (size_t)L->num_boxes), sel;
  *grid = 0; *per_xcd = 0; *count = 0;
  if (!L->box_nbr || hipMemcpy(nbr.data(), L->box_nbr, nbr.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
  for (int l = 0; l < total; l++) {
    int t = l;
    const int ti = t % tiles_i; t /= tiles_i;
    const int tj = t % tiles_j; t /= tiles_j;
    const int ck = t % chunks_k; t /= chunks_k;
    const int *nb = &nbr[6 * (size_t)t];
    const bool at[6] = { ti == 0, ti == tiles_i - 1, tj == 0, tj == tiles_j - 1, ck == 0, ck == chunks_k - 1 };
    bool later = false;
    for (int d = 0; d < 6; d++) if (at[d] && (nb[d] >= L->num_boxes || nb[d] == -2 || (walls_to_part2 && nb[d] == -1))) later = true;
    if (later == (part == 2)) sel.push_back(l);
  }
  PartOrder o = { L->box_nbr, L->num_boxes, tiles_i, tiles_j, chunks_k, part, (int)walls_to_part2, nullptr, 0, 0, (int)sel.size() };
  if (!sel.empty()) {
    o.per_xcd = ((int)sel.size() + kXcds - 1) / kXcds; o.grid = o.per_xcd * kXcds;
    sel.resize((size_t)o.grid, total);                                     // the padding of the grid: nothing to do
    if (hipMalloc(&o.d_order, sel.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(o.d_order, sel.data(), sel.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(o.d_order); return nullptr; }
  }
  g_part_orders.push_back(o);
  *grid = o.grid; *per_xcd = o.per_xcd; *count = o.count;
  return o.d_order;
}
}  // namespace hpgmg
// the neighbour table a cached order was derived from is being freed (hpgmg_hip_free): drop the orders and their device copies
extern "C" void hpgmg_hip_tile_part_forget(const void *p) {
  using hpgmg::g_part_orders;
  for (size_t n = 0; n < g_part_orders.size();) {
    if ((const void *)g_part_orders[n].nbr != p) { n++; continue; }
    if (g_part_orders[n].d_order) (void)hipFree(g_part_orders[n].d_order);
    g_part_orders[n] = g_part_orders.back(); g_part_orders.pop_back();
  }
}
extern "C" void hpgmg_hip_set_tile_part(int part) { hpgmg::g_tile_part = (part == 1 || part == 2) ? part : 0; }

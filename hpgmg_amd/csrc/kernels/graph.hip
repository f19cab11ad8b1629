// graph.hip -- hipGraph capture/replay of launch-bound segments of the multigrid cycle.
//
// Levels of 64^3 cells and below are launch-latency bound: a level visit there is a dozen
// kernels of a few microseconds each.  The host driver (hpgmg_amd/csrc/host/mg.c) brackets
// the work done on those levels between two bottom solves as a SEGMENT, identified by a key
// that is the same every solve.  A segment is executed eagerly the first time (so lazily
// created device mirrors exist), captured into a hipGraph the second time, and from then on
// REPLAYED: while a replay segment is open every launcher in this library returns at once
// (nothing is enqueued), and closing the segment launches the instantiated graph.  Kernel
// arguments are baked into the graph, which is valid because a segment's operation sequence,
// vector ids and coefficients are fixed after MGBuild; the cache is dropped whenever a level is
// released or the configuration changes.  Reductions / copies that synchronise with the host
// close the open segment first (they cannot be part of a graph).
#include <stdio.h>
#include <stdlib.h>
#include <unordered_map>
#include "common.hpp"

namespace hpgmg {
enum { SEG_NONE = 0, SEG_EAGER, SEG_CAPTURE, SEG_REPLAY };
struct Entry { int seen = 0; hipGraphExec_t exec = nullptr; };
static std::unordered_map<long long, Entry> g_cache;
static int g_enabled = 1;
static int g_state = SEG_NONE;
static long long g_key = 0;
static long long g_stats[3] = {0, 0, 0};   // eager, captured, replayed segments
int g_skip_launches = 0;                   // read by every launcher (common.hpp)
}  // namespace hpgmg
using namespace hpgmg;

extern "C" {

void hpgmg_hip_graph_enable(int on) { g_enabled = on; }
int hpgmg_hip_graph_enabled(void) { return g_enabled; }

int hpgmg_hip_graph_end(void) {
  const int state = g_state;
  g_state = SEG_NONE;
  g_skip_launches = 0;
  if (state == SEG_CAPTURE) {
    hipGraph_t graph = nullptr;
    HPGMG_CHECK(hipStreamEndCapture(g_stream, &graph));
    size_t nodes = 0;
    if (graph) HPGMG_CHECK(hipGraphGetNodes(graph, nullptr, &nodes));
    if (!graph || nodes == 0) {                      // nothing was launched (a rank that owns no box of these levels): stay eager
      if (graph) HPGMG_CHECK(hipGraphDestroy(graph));
      g_cache[g_key].seen = 2;
      g_stats[0]++;
      return 0;
    }
    hipGraphExec_t exec = nullptr;
    HPGMG_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HPGMG_CHECK(hipGraphDestroy(graph));
    g_cache[g_key].exec = exec;
    HPGMG_CHECK(hipGraphLaunch(exec, g_stream));
    g_stats[1]++;
  } else if (state == SEG_REPLAY) {
    HPGMG_CHECK(hipGraphLaunch(g_cache[g_key].exec, g_stream));
    g_stats[2]++;
  } else if (state == SEG_EAGER) {
    g_stats[0]++;
  }
  return 0;
}

// Open a segment (closing any segment still open).  Returns the mode chosen.
int hpgmg_hip_graph_begin(long long key) {
  if (g_state != SEG_NONE) { if (int e = hpgmg_hip_graph_end()) return -e; }
  if (!g_enabled) return SEG_NONE;
  Entry &e = g_cache[key];
  g_key = key;
  if (e.exec) { g_state = SEG_REPLAY; g_skip_launches = 1; }
  else if (e.seen == 2) g_state = SEG_EAGER;          // known to be empty
  else if (e.seen) {
    const hipError_t ce = hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal);
    if (ce != hipSuccess) {
      static bool told = false;
      if (!told && getenv("HPGMG_GRAPH_DEBUG")) { fprintf(stderr, "hpgmg_hip: hipStreamBeginCapture failed: %s (stream %p)\n", hipGetErrorString(ce), (void *)g_stream); told = true; }
      (void)hipGetLastError(); g_state = SEG_EAGER;
    }
    else g_state = SEG_CAPTURE;
  } else { e.seen = 1; g_state = SEG_EAGER; }
  return g_state;
}

int hpgmg_hip_graph_is_open(void) { return g_state != SEG_NONE; }

// Host-synchronising operations call this first: they cannot live inside a captured segment.
int hpgmg_hip_graph_flush(void) { return (g_state != SEG_NONE) ? hpgmg_hip_graph_end() : 0; }

void hpgmg_hip_graph_reset(void) {
  if (g_state == SEG_CAPTURE) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(g_stream, &g); if (g) (void)hipGraphDestroy(g); }
  g_state = SEG_NONE; g_skip_launches = 0;
  (void)hipStreamSynchronize(g_stream);
  for (auto &kv : g_cache) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  g_cache.clear();
}

void hpgmg_hip_graph_stats(long long out[3]) { out[0] = g_stats[0]; out[1] = g_stats[1]; out[2] = g_stats[2]; }

}  // extern "C"

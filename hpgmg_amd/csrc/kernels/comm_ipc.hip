// comm_ipc.hip -- a second transport for one node: direct peer copies between the processes' device buffers (hipIpc memory handles; over
// xGMI between GPUs, inside the HBM when the ranks share a GPU) ordered against both streams by host functions on shared counters, no collective library and no staging through
// the host.  Same two callbacks as comm_rccl.hip (include/hpgmg_mg.h: hpgmg_transport), so the plugin does not know which one it drives.
//
// Reference call sites replaced (finite-volume/source/operators/): exchange_boundary.c:33-97, restriction.c:128-192,
// interpolation_p*.c:74-139 (MPI_Irecv / MPI_Isend / MPI_Waitall -> sendrecv) and misc.c:276,324,373 (MPI_Allreduce of one double -> allreduce).
//
// One message (sender s -> receiver r) of a sendrecv() phase, all on the two ranks' launch streams:
//     r: enqueues READY(r, s) := n (everything r issued so far -- the kernels that still read the receive buffer -- precedes it) and posts a
//        descriptor of the receive buffer (hipIpc handle of its allocation + offset) in the channel (r, s) of a small shared-memory segment;
//     s: sees the descriptor, enqueues "wait until READY(r, s) >= n", then the copy  its send buffer -> r's receive buffer  (stream-ordered
//        after its pack kernel), then SENT(r, s) := n, and bumps the channel's `sent` counter;
//     r: sees the counter, enqueues "wait until SENT(r, s) >= n" and returns: the unpack kernel it launches next runs after the data landed.
// READY / SENT are counters in the shared segment, set and awaited by HOST FUNCTIONS in stream order (hipLaunchHostFunc): the mechanism HIP's own
// interprocess events are built on.  (Those events were tried first: hipStreamWaitEvent on an event another process had recorded answered
// "invalid argument" from the solver's streams on ROCm 7.2 -- in a two-process probe the same calls work, tools/microbench/ipc_probe*.hip.)
// The calling host threads only exchange the descriptor and the `sent` counter per message (a few cache lines in shared memory, microseconds);
// neither side synchronises its stream, and the data never touches host memory.  Channel counters pair the n-th send of s to r with the n-th receive of r from s -- the
// order both sides derive from the level's message plan, exactly what MPI's non-overtaking rule gives the reference.
// Scalars (8 bytes, already on the host when a reduction is called) go through the shared-memory segment itself and are reduced in rank order:
// the same association on every rank and from run to run.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <time.h>
#include <sched.h>
#include <sys/mman.h>
#include <atomic>
#include <vector>
#include "common.hpp"

namespace hpgmg {
constexpr int kIpcMaxRanks = 16, kIpcMaxVals = 16;
struct IpcDesc { hipIpcMemHandle_t mem; long long offset, size; int tag, pad; unsigned long long gen; };      // gen: which export of the receiver's this handle is (a freed allocation's address may come back)
struct IpcChannel {                                  // data flows sender -> receiver; lives at [receiver][sender]
  std::atomic<unsigned long long> posted, sent;
  std::atomic<unsigned long long> ready_done, sent_done;      // set in STREAM order: the receiver's stream has reached receive n / the sender's copy n has landed
  IpcDesc desc;                                      // of receive number `posted`
  std::atomic<unsigned long long> red_seq, red_ack;  // scalars: receiver-side copy of the sender's values / how many the receiver has consumed
  double red_val[kIpcMaxVals];
};
struct IpcRank { std::atomic<int> alive; };
struct IpcSegment {
  std::atomic<unsigned long long> magic;
  std::atomic<int> arrived, departed;
  int size;
  IpcRank rank[kIpcMaxRanks];
  IpcChannel ch[kIpcMaxRanks][kIpcMaxRanks];
};
constexpr unsigned long long kIpcMagic = 0x68706967636d6763ull;

static IpcSegment *g_seg = nullptr;
static char g_seg_name[128];
static int g_irank = 0, g_isize = 1;
static unsigned long long g_n_posted[kIpcMaxRanks], g_n_sent[kIpcMaxRanks], g_red_out[kIpcMaxRanks], g_red_in[kIpcMaxRanks];
static long long g_ipc_messages = 0, g_ipc_doubles = 0;
struct ExportedMem { void *base; size_t size; hipIpcMemHandle_t h; unsigned long long gen; };
struct MappedMem { int peer; hipIpcMemHandle_t h; void *base; unsigned long long gen; };
static unsigned long long g_export_gen = 0;
static std::vector<ExportedMem> g_exported;
static std::vector<MappedMem> g_mapped;

static double wall(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static void ipc_die(const char *what) { fprintf(stderr, "hpgmg_hip (ipc transport, rank %d of %d): %s\n", g_irank, g_isize, what); fflush(stderr); abort(); }
// a peer that never answers must not hang the node: every wait gives up after HPGMG_IPC_TIMEOUT seconds (default 120)
static double ipc_timeout(void) { static double t = -1; if (t < 0) { const char *e = getenv("HPGMG_IPC_TIMEOUT"); t = (e && *e) ? atof(e) : 120.0; } return t; }
template <class F> static void spin_until(F done, const char *what) {
  const double t0 = wall();
  for (unsigned long long n = 0; !done(); n++) {
    if ((n & 0x3ff) == 0x3ff) { if (wall() - t0 > ipc_timeout()) ipc_die(what); sched_yield(); }
    __builtin_ia32_pause();
  }
}
static int ipc_hip_fail(hipError_t e, const char *where) { fprintf(stderr, "hpgmg_hip (ipc transport, rank %d): %s: %s\n", g_irank, where, hipGetErrorString(e)); return (int)e; }
static int ipc_debug(void) { static int d = -1; if (d < 0) { const char *e = getenv("HPGMG_IPC_DEBUG"); d = (e && e[0] == '1'); } return d; }
#define IPC_OK(call) do {                                                                                                                    \
    hipError_t e_ = (call);                                                                                                                  \
    if (ipc_debug()) { fprintf(stderr, "[ipc %d] %s -> %s (stream %p)\n", g_irank, #call, hipGetErrorString(e_), (void *)g_stream); fflush(stderr); } \
    if (e_ != hipSuccess) { ipc_hip_fail(e_, #call); abort(); } } while (0)

// stream-ordered "set counter" / "wait for counter": host functions (they run on the runtime's callback thread, in stream order)
// Their arguments live on the heap from the enqueue to the end of the callback: a host thread may run a whole F-cycle of messages ahead of its
// stream, so no fixed ring of slots is safe to reuse.
struct FlagOp { std::atomic<unsigned long long> *flag; unsigned long long value; };
static FlagOp *flag_op(std::atomic<unsigned long long> *flag, unsigned long long value) {
  FlagOp *f = new FlagOp;
  f->flag = flag; f->value = value;
  return f;
}
static void host_set_flag(void *p) { const FlagOp *f = (const FlagOp *)p; f->flag->store(f->value, std::memory_order_release); delete f; }
static void host_wait_flag(void *p) {
  const FlagOp *f = (const FlagOp *)p;
  const double t0 = wall();
  for (unsigned long long n = 0; f->flag->load(std::memory_order_acquire) < f->value; n++) {
    if ((n & 0xfff) == 0xfff && wall() - t0 > ipc_timeout()) { fprintf(stderr, "hpgmg_hip (ipc transport, rank %d): a peer's stream never reached the point this stream waits for\n", g_irank); fflush(stderr); abort(); }
    if ((n & 0xff) == 0xff) sched_yield();      // several ranks on one GPU share the host cores: the peer this callback waits for may need this one
    __builtin_ia32_pause();
  }
  delete f;
}
static const ExportedMem &export_handle(const void *p, long long *offset) {
  void *base = nullptr; size_t size = 0;
  IPC_OK(hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)p));
  *offset = (long long)((const char *)p - (const char *)base);
  for (const ExportedMem &e : g_exported) if (e.base == base && e.size == size) return e;
  ExportedMem e; e.base = base; e.size = size; e.gen = ++g_export_gen;
  IPC_OK(hipIpcGetMemHandle(&e.h, base));
  g_exported.push_back(e);
  return g_exported.back();
}
// (peer, gen) names one allocation of the peer's for as long as it lives; a mapping of the same handle bytes under an older generation is of an
// allocation the peer has freed since (hpgmg_hip_ipc_forget on its side): closed before the new one is opened
static void *map_peer(int peer, const hipIpcMemHandle_t &h, unsigned long long gen) {
  for (const MappedMem &m : g_mapped) if (m.peer == peer && m.gen == gen) return m.base;
  for (size_t n = 0; n < g_mapped.size();) {
    if (g_mapped[n].peer != peer || memcmp(&g_mapped[n].h, &h, sizeof h) != 0) { n++; continue; }
    IPC_OK(hipStreamSynchronize(g_stream));               // copies of mine into the stale mapping have drained
    (void)hipIpcCloseMemHandle(g_mapped[n].base);
    g_mapped[n] = g_mapped.back(); g_mapped.pop_back();
  }
  MappedMem m; m.peer = peer; m.h = h; m.base = nullptr; m.gen = gen;
  IPC_OK(hipIpcOpenMemHandle(&m.base, h, hipIpcMemLazyEnablePeerAccess));
  g_mapped.push_back(m);
  return m.base;
}
}  // namespace hpgmg
using namespace hpgmg;

extern "C" {
int hpgmg_hip_graph_flush(void);

// `name`: what the ranks of one job agree on (a POSIX shared-memory name, "/..."); rank 0 creates the segment, the others attach to it
int hpgmg_hip_ipc_init(const char *name, int rank, int size) {
  if (size < 1 || size > kIpcMaxRanks || rank < 0 || rank >= size || !name || name[0] != '/' || strlen(name) >= sizeof g_seg_name) return record_error(hipErrorInvalidValue, "ipc_init: 1..16 ranks, a name that starts with '/'");
  g_irank = rank; g_isize = size;
  strcpy(g_seg_name, name);
  int fd = -1;
  if (rank == 0) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(IpcSegment)) != 0) { perror("hpgmg_hip: shm_open"); return record_error(hipErrorInvalidValue, "ipc_init: cannot create the shared-memory segment"); }
  } else {
    const double t0 = wall();
    while ((fd = shm_open(name, O_RDWR, 0600)) < 0) { if (wall() - t0 > ipc_timeout()) return record_error(hipErrorInvalidValue, "ipc_init: the shared-memory segment never appeared"); usleep(1000); }
    off_t len = 0;
    while ((len = lseek(fd, 0, SEEK_END)) < (off_t)sizeof(IpcSegment)) { if (wall() - t0 > ipc_timeout()) return record_error(hipErrorInvalidValue, "ipc_init: the shared-memory segment stayed empty"); usleep(1000); }
  }
  g_seg = (IpcSegment *)mmap(nullptr, sizeof(IpcSegment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (g_seg == (IpcSegment *)MAP_FAILED) { g_seg = nullptr; return record_error(hipErrorInvalidValue, "ipc_init: mmap"); }
  if (rank == 0) { g_seg->size = size; g_seg->magic.store(kIpcMagic, std::memory_order_release); }       // a fresh segment is zero-filled
  else spin_until([&] { return g_seg->magic.load(std::memory_order_acquire) == kIpcMagic; }, "rank 0 never initialised the segment");
  if (g_seg->size != size) return record_error(hipErrorInvalidValue, "ipc_init: the ranks disagree on the job size");
  for (int p = 0; p < size; p++) g_n_posted[p] = g_n_sent[p] = g_red_out[p] = g_red_in[p] = 0;
  g_seg->rank[rank].alive.store(1, std::memory_order_release);
  g_seg->arrived.fetch_add(1);
  spin_until([&] { return g_seg->arrived.load() >= size; }, "not every rank reached ipc_init");
  return 0;
}

void hpgmg_hip_ipc_finalize(void) {
  if (!g_seg) return;
  (void)hipStreamSynchronize(g_stream);
  g_seg->departed.fetch_add(1);
  spin_until([&] { return g_seg->departed.load() >= g_isize; }, "not every rank reached ipc_finalize");      // nobody unmaps what a peer may still be copying into
  for (const MappedMem &m : g_mapped) (void)hipIpcCloseMemHandle(m.base);
  g_mapped.clear(); g_exported.clear();
  munmap(g_seg, sizeof(IpcSegment)); g_seg = nullptr;
  if (g_irank == 0) shm_unlink(g_seg_name);
}
long long hpgmg_hip_ipc_message_count(void) { return g_ipc_messages; }
// an allocation is being freed (hpgmg_hip_free): its exported handle must not be handed out for whatever lands on that address next
void hpgmg_hip_ipc_forget(const void *p) {
  for (size_t n = 0; n < g_exported.size();) {
    if (g_exported[n].base != p) { n++; continue; }
    g_exported[n] = g_exported.back(); g_exported.pop_back();
  }
}

// signature = hpgmg_transport.sendrecv (include/hpgmg_mg.h); buffers are device memory
void hpgmg_hip_ipc_sendrecv(void *ctx, int nrecv, double *const *rbuf, const int *rsize, const int *rrank,
                            int nsend, double *const *sbuf, const int *ssize, const int *srank, int tag) {
  (void)ctx;
  hpgmg_hip_graph_flush();
  if (!g_seg) ipc_die("transport used before hpgmg_hip_ipc_init");
  if (nrecv + nsend == 0) return;

  std::vector<int> r_state((size_t)nrecv, 0), s_state((size_t)nsend, 0);       // 0 waiting for the channel, 1 in flight, 2 done
  int open = nrecv + nsend;
  const double t0 = wall();
  unsigned long long spins = 0;
  while (open > 0) {
    for (int n = 0; n < nrecv; n++) {
      const int s = rrank[n];
      IpcChannel &c = g_seg->ch[g_irank][s];
      if (r_state[n] == 0 && n > 0 && rrank[n - 1] == s && r_state[n - 1] != 2) continue;      // receives from one rank are posted in plan order, like the sends
      if (r_state[n] == 0 && c.sent.load(std::memory_order_acquire) == g_n_posted[s]) {       // the previous receive on this channel has been served: post this one
        long long off = 0;
        const ExportedMem &ex = export_handle(rbuf[n], &off);
        c.desc.mem = ex.h; c.desc.gen = ex.gen; c.desc.offset = off; c.desc.size = rsize[n]; c.desc.tag = tag;
        // READY: whatever this rank issued so far -- in particular the kernels that still read the receive buffer -- precedes the incoming copy
        IPC_OK(hipLaunchHostFunc(g_stream, host_set_flag, flag_op(&c.ready_done, g_n_posted[s] + 1)));
        c.posted.store(++g_n_posted[s], std::memory_order_release);
        r_state[n] = 1;
      } else if (r_state[n] == 1 && c.sent.load(std::memory_order_acquire) == g_n_posted[s]) {
        IPC_OK(hipLaunchHostFunc(g_stream, host_wait_flag, flag_op(&c.sent_done, g_n_posted[s])));      // the copy into rbuf[n] precedes what I launch next
        r_state[n] = 2; open--;
      }
    }
    for (int n = 0; n < nsend; n++) {
      const int d = srank[n];
      IpcChannel &c = g_seg->ch[d][g_irank];
      if (s_state[n] != 0 || (n > 0 && srank[n - 1] == d && s_state[n - 1] == 0)) continue;      // messages to one rank go in order
      if (c.posted.load(std::memory_order_acquire) != g_n_sent[d] + 1) continue;                 // the receiver has not posted this one yet
      if (c.desc.size != (long long)ssize[n] || c.desc.tag != tag) {
        fprintf(stderr, "hpgmg_hip (ipc transport): rank %d sends %d doubles (tag %d) to rank %d, which expects %lld (tag %d)\n", g_irank, ssize[n], tag, d, c.desc.size, c.desc.tag);
        abort();
      }
      double *peer = (double *)((char *)map_peer(d, c.desc.mem, c.desc.gen) + c.desc.offset);
      IPC_OK(hipLaunchHostFunc(g_stream, host_wait_flag, flag_op(&c.ready_done, g_n_sent[d] + 1)));      // the receiver is done with the buffer's previous content
      IPC_OK(hipMemcpyAsync(peer, sbuf[n], (size_t)ssize[n] * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
      IPC_OK(hipLaunchHostFunc(g_stream, host_set_flag, flag_op(&c.sent_done, g_n_sent[d] + 1)));
      c.sent.store(++g_n_sent[d], std::memory_order_release);
      g_ipc_messages++; g_ipc_doubles += ssize[n];
      s_state[n] = 2; open--;
    }
    if ((++spins & 0x3ff) == 0) { if (wall() - t0 > ipc_timeout()) ipc_die("a message was never matched by its peer (sendrecv)"); sched_yield(); }
  }
}

// signature = hpgmg_transport.allreduce: n host doubles, in place, over `ranks` (sorted, contains me); reduced in rank order
void hpgmg_hip_ipc_allreduce(void *ctx, double *vals, int n, int op, const int *ranks, int nranks) {
  (void)ctx;
  if (nranks <= 1) return;
  if (!g_seg) ipc_die("transport used before hpgmg_hip_ipc_init");
  if (n < 1 || n > kIpcMaxVals) ipc_die("allreduce of more than 16 values");
  for (int q = 0; q < nranks; q++) {                    // my values into every member's copy of my channel, once it has consumed the previous ones
    const int p = ranks[q];
    if (p == g_irank) continue;
    IpcChannel &c = g_seg->ch[p][g_irank];
    spin_until([&] { return c.red_ack.load(std::memory_order_acquire) == g_red_out[p]; }, "a peer never consumed the previous reduction");
    memcpy(c.red_val, vals, (size_t)n * sizeof(double));
    c.red_seq.store(++g_red_out[p], std::memory_order_release);
  }
  double acc[kIpcMaxVals]; bool first = true;
  for (int q = 0; q < nranks; q++) {
    const int p = ranks[q];
    double theirs[kIpcMaxVals];
    if (p == g_irank) memcpy(theirs, vals, (size_t)n * sizeof(double));
    else {
      IpcChannel &c = g_seg->ch[g_irank][p];
      spin_until([&] { return c.red_seq.load(std::memory_order_acquire) == g_red_in[p] + 1; }, "a peer never joined the reduction");
      memcpy(theirs, c.red_val, (size_t)n * sizeof(double));
      c.red_ack.store(++g_red_in[p], std::memory_order_release);
    }
    for (int v = 0; v < n; v++) acc[v] = first ? theirs[v] : (op == 0 ? (theirs[v] > acc[v] ? theirs[v] : acc[v]) : acc[v] + theirs[v]);      // HPGMG_REDUCE_MAX = 0, HPGMG_REDUCE_SUM = 1
    first = false;
  }
  memcpy(vals, acc, (size_t)n * sizeof(double));
}

}  // extern "C"

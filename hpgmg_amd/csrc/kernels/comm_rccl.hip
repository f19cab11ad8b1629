// comm_rccl.hip -- the transport that replaces the reference's MPI calls when one process
// drives one MI355X: RCCL point-to-point over xGMI, enqueued on the SAME stream as the
// pack / unpack kernels, so a ghost exchange is  pack kernel -> grouped send/recv -> unpack
// kernel  in stream order with no host synchronisation.
//
// Reference call sites replaced (finite-volume/source/operators/):
//   exchange_boundary.c:33-97, restriction.c:128-192, interpolation_p*.c:74-139
//     MPI_Irecv* + MPI_Isend* + MPI_Waitall   ->  hpgmg_hip_rccl_sendrecv (one ncclGroup)
//   misc.c:276,324,373  MPI_Allreduce(1 double, SUM|MAX, sub-communicator)
//                                              ->  hpgmg_hip_rccl_allreduce over the listed ranks
// Message sizes are small (a 128^2 face = 128 KiB, 4 faces per neighbour), every neighbour
// pair has its own xGMI link, so one grouped call per exchange is the right granularity.
// The scalar reductions involve only the ranks active on that level (the reference's per-level
// MPI_Comm_split).  A MAX over every rank of the job -- norm() on a level all ranks share, the
// only reduction of a Dirichlet solve that crosses ranks once the coarse levels are gathered --
// is one ncclAllReduce (a maximum is exact in any order).  A SUM over every rank of the job (dot() and mean() on a level all ranks
// share: ten per iteration of a host-driven BiCGStab, solvers/bicgstab.c:14-97) is ONE ncclAllGather of the partials followed by an
// addition in rank order on the host: one collective, and the same association on every rank and from run to run, which the golden
// numbers need (ncclAllReduce(ncclSum) would add in whatever order the ring or tree RCCL picked).  Reductions over a SUBSET of the
// ranks (levels below the agglomeration point; the reference: MPI_Comm_split per level, mg.c:985-993) run the same two ways on a
// sub-communicator made by ncclCommSplit when MGBuild announces the set (hpgmg_hip_rccl_prepare_subset: collective over the whole
// job); a set nobody announced, or whose split failed, falls back to an all-to-all of 8-byte messages, reduced the same way.
// (Written without a multi-GPU node to run it on: the split path has never executed; the fallback is what the one-GPU tests cover.)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>
#include "common.hpp"

namespace hpgmg {
static ncclComm_t g_comm = nullptr;
static int g_rank = 0, g_size = 1;
constexpr int kRedMax = 16;           // values per reduction call (the path reduces one)
static double *g_red_dev = nullptr;   // [(g_size + 1) * kRedMax] staging: every rank's partials, then my own
static double *g_red_host = nullptr;  // pinned, same size
static long long g_allgathers = 0;
// sub-communicators: one per distinct set of ranks announced by MGBuild; comm == nullptr: announced, but this rank is no member / the split failed
struct SubComm { int n; int *ranks; ncclComm_t comm; int ok; };
constexpr int kMaxSubComms = 64;
static SubComm g_sub[kMaxSubComms];
static int g_nsub = 0;
static long long g_subset_collectives = 0, g_subset_alltoalls = 0;
static SubComm *find_sub(const int *ranks, int n) {
  for (int s = 0; s < g_nsub; s++) if (g_sub[s].n == n && memcmp(g_sub[s].ranks, ranks, (size_t)n * sizeof(int)) == 0) return &g_sub[s];
  return nullptr;
}
static int nccl_fail(ncclResult_t r, const char *where) {
  fprintf(stderr, "hpgmg_hip: %s: %s\n", where, ncclGetErrorString(r));
  return 1000 + (int)r;
}
#define NCCL_OK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return nccl_fail(r_, #call); } while (0)
}  // namespace hpgmg
using namespace hpgmg;

extern "C" {
int hpgmg_hip_graph_flush(void);

int hpgmg_hip_rccl_unique_id(char *out128) {
  ncclUniqueId id;
  NCCL_OK(ncclGetUniqueId(&id));
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int hpgmg_hip_rccl_init(const char *id128, int rank, int size) {
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  NCCL_OK(ncclCommInitRank(&g_comm, size, id, rank));
  g_rank = rank; g_size = size;
  HPGMG_CHECK(hipMalloc((void **)&g_red_dev, (size_t)(size + 1) * kRedMax * sizeof(double)));
  HPGMG_CHECK(hipHostMalloc((void **)&g_red_host, (size_t)(size + 1) * kRedMax * sizeof(double), hipHostMallocDefault));
  return 0;
}

// every rank of the job, same sets in the same order (MGBuild): color = the set's number for its members, no color for the others
void hpgmg_hip_rccl_prepare_subset(void *ctx, const int *ranks, int nranks) {
  (void)ctx;
  if (!g_comm || nranks < 2 || nranks >= g_size || find_sub(ranks, nranks) || g_nsub >= kMaxSubComms) return;
  SubComm &S = g_sub[g_nsub];
  S.n = nranks; S.ranks = (int *)malloc((size_t)nranks * sizeof(int)); memcpy(S.ranks, ranks, (size_t)nranks * sizeof(int)); S.comm = nullptr; S.ok = 0;
  int member = 0;
  for (int q = 0; q < nranks; q++) if (ranks[q] == g_rank) member = 1;
  static const int off = [] { const char *e = getenv("HPGMG_RCCL_SUBCOMM"); return (e && e[0] == '0') ? 1 : 0; }();      // 0: keep the all-to-all (every rank must say the same)
  if (!off) {
    ncclComm_t sub = nullptr;
    const ncclResult_t r = ncclCommSplit(g_comm, member ? g_nsub : NCCL_SPLIT_NOCOLOR, g_rank, &sub, nullptr);      // key = my rank: the members keep their order
    int fine = (r == ncclSuccess && (sub != nullptr || !member)) ? 1 : 0;
    if (!fine) nccl_fail(r, "ncclCommSplit (subset reductions fall back to the all-to-all)");
    // every rank learns whether EVERY rank got its part: members that disagreed about the path would wait for each other forever
    int *flag_host = (int *)g_red_host, *flag_dev = (int *)g_red_dev;
    *flag_host = fine;
    if (hipMemcpyAsync(flag_dev, flag_host, sizeof(int), hipMemcpyHostToDevice, g_stream) != hipSuccess ||
        ncclAllReduce(flag_dev, flag_dev, 1, ncclInt, ncclMin, g_comm, g_stream) != ncclSuccess ||
        hipMemcpyAsync(flag_host, flag_dev, sizeof(int), hipMemcpyDeviceToHost, g_stream) != hipSuccess || hipStreamSynchronize(g_stream) != hipSuccess) { fprintf(stderr, "hpgmg_hip: agreeing on a sub-communicator failed\n"); abort(); }
    if (*flag_host) { S.comm = sub; S.ok = member ? 1 : 0; }
    else { if (sub) ncclCommDestroy(sub); S.comm = nullptr; S.ok = 0; }
  }
  g_nsub++;
}
long long hpgmg_hip_rccl_subset_collectives(void) { return g_subset_collectives; }      // subset reductions done as ONE collective on a sub-communicator / as an all-to-all
long long hpgmg_hip_rccl_subset_alltoalls(void) { return g_subset_alltoalls; }

void hpgmg_hip_rccl_finalize(void) {
  for (int s = 0; s < g_nsub; s++) { if (g_sub[s].comm) { hipStreamSynchronize(g_stream); ncclCommDestroy(g_sub[s].comm); } free(g_sub[s].ranks); }
  g_nsub = 0;
  if (g_comm) { hipStreamSynchronize(g_stream); ncclCommDestroy(g_comm); g_comm = nullptr; }
  if (g_red_dev) { (void)hipFree(g_red_dev); g_red_dev = nullptr; }
  if (g_red_host) { (void)hipHostFree(g_red_host); g_red_host = nullptr; }
}

// signature = hpgmg_transport.sendrecv (include/hpgmg_mg.h); buffers are device memory
void hpgmg_hip_rccl_sendrecv(void *ctx, int nrecv, double *const *rbuf, const int *rsize, const int *rrank,
                             int nsend, double *const *sbuf, const int *ssize, const int *srank, int tag) {
  hpgmg_hip_graph_flush();
  (void)ctx; (void)tag;   // ordering inside one stream + one group per phase makes tags unnecessary
  if (!g_comm) { fprintf(stderr, "hpgmg_hip: RCCL transport used before hpgmg_hip_rccl_init\n"); abort(); }
  ncclResult_t r = ncclGroupStart();
  for (int n = 0; n < nrecv && r == ncclSuccess; n++) r = ncclRecv(rbuf[n], (size_t)rsize[n], ncclDouble, rrank[n], g_comm, g_stream);
  for (int n = 0; n < nsend && r == ncclSuccess; n++) r = ncclSend(sbuf[n], (size_t)ssize[n], ncclDouble, srank[n], g_comm, g_stream);
  if (r == ncclSuccess) r = ncclGroupEnd();
  if (r != ncclSuccess) { nccl_fail(r, "grouped ncclSend/ncclRecv"); abort(); }
}

// MAX of n (<= job size) host doubles over EVERY rank of the communicator, in place: misc.c:324 MPI_Allreduce(MPI_MAX) on a level all ranks share
int hpgmg_hip_rccl_allreduce_max_world(double *vals, int n) {
  if (!g_comm || n > kRedMax || n < 1) return record_error(hipErrorInvalidValue, "rccl_allreduce_max_world");
  memcpy(g_red_host, vals, (size_t)n * sizeof(double));
  HPGMG_CHECK(hipMemcpyAsync(g_red_dev, g_red_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_stream));
  NCCL_OK(ncclAllReduce(g_red_dev, g_red_dev, (size_t)n, ncclDouble, ncclMax, g_comm, g_stream));
  HPGMG_CHECK(hipMemcpyAsync(g_red_host, g_red_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  HPGMG_CHECK(hipStreamSynchronize(g_stream));             // the caller needs the value on the host, as after MPI_Allreduce
  memcpy(vals, g_red_host, (size_t)n * sizeof(double));
  return 0;
}

// SUM (or MAX) of n host doubles over EVERY rank, in place, as one ncclAllGather + a reduction in rank order on the host (misc.c:276,373)
long long hpgmg_hip_rccl_allgather_count(void) { return g_allgathers; }
int hpgmg_hip_rccl_allreduce_ordered_world(double *vals, int n, int op) {
  if (!g_comm || n > kRedMax || n < 1) return record_error(hipErrorInvalidValue, "rccl_allreduce_ordered_world");
  double *mine_dev = g_red_dev + (size_t)g_size * kRedMax, *mine_host = g_red_host + (size_t)g_size * kRedMax;
  memcpy(mine_host, vals, (size_t)n * sizeof(double));
  HPGMG_CHECK(hipMemcpyAsync(mine_dev, mine_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_stream));
  NCCL_OK(ncclAllGather(mine_dev, g_red_dev, (size_t)n, ncclDouble, g_comm, g_stream));       // rank r's values land at [r * n, (r + 1) * n)
  HPGMG_CHECK(hipMemcpyAsync(g_red_host, g_red_dev, (size_t)g_size * n * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  HPGMG_CHECK(hipStreamSynchronize(g_stream));             // the caller needs the value on the host, as after MPI_Allreduce
  for (int v = 0; v < n; v++) {
    double acc = g_red_host[v];
    for (int r = 1; r < g_size; r++) { const double x = g_red_host[(size_t)r * n + v]; if (op == 0) acc = (x > acc) ? x : acc; else acc += x; }
    vals[v] = acc;
  }
  g_allgathers++;
  return 0;
}

// signature = hpgmg_transport.allreduce: n host doubles, in place, over `ranks` (sorted, contains me)
void hpgmg_hip_rccl_allreduce(void *ctx, double *vals, int n, int op, const int *ranks, int nranks) {
  (void)ctx;
  hpgmg_hip_graph_flush();
  if (nranks <= 1) return;
  if (!g_comm) { fprintf(stderr, "hpgmg_hip: RCCL transport used before hpgmg_hip_rccl_init\n"); abort(); }
  if (op == 0 && nranks == g_size && n <= kRedMax) { if (hpgmg_hip_rccl_allreduce_max_world(vals, n)) abort(); return; }
  if (nranks == g_size && n <= kRedMax) { if (hpgmg_hip_rccl_allreduce_ordered_world(vals, n, op)) abort(); return; }      // sums: one collective, rank-ordered association
  if (SubComm *S = find_sub(ranks, nranks)) if (S->ok && S->comm && n <= kRedMax) {
    // the level's sub-communicator (its ranks are the members in rank order: key = rank): a maximum as ONE ncclAllReduce, a sum as ONE ncclAllGather + the
    // addition in member order on the host -- what the whole-job forms above do
    double *mine_dev = g_red_dev + (size_t)g_size * kRedMax, *mine_host = g_red_host + (size_t)g_size * kRedMax;
    memcpy(mine_host, vals, (size_t)n * sizeof(double));
    ncclResult_t r = ncclSuccess;
    if (hipMemcpyAsync(mine_dev, mine_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_stream) != hipSuccess) abort();
    if (op == 0) r = ncclAllReduce(mine_dev, g_red_dev, (size_t)n, ncclDouble, ncclMax, S->comm, g_stream);
    else         r = ncclAllGather(mine_dev, g_red_dev, (size_t)n, ncclDouble, S->comm, g_stream);
    if (r != ncclSuccess) { nccl_fail(r, "subset collective"); abort(); }
    const size_t back = (op == 0) ? (size_t)n : (size_t)nranks * n;
    if (hipMemcpyAsync(g_red_host, g_red_dev, back * sizeof(double), hipMemcpyDeviceToHost, g_stream) != hipSuccess || hipStreamSynchronize(g_stream) != hipSuccess) abort();
    for (int v = 0; v < n; v++) {
      double acc = g_red_host[v];
      if (op != 0) for (int q = 1; q < nranks; q++) acc += g_red_host[(size_t)q * n + v];
      vals[v] = acc;
    }
    g_subset_collectives++;
    return;
  }
  g_subset_alltoalls++;
  for (int v = 0; v < n; v++) {   // n is 1 everywhere on the path; keep the general form simple
    g_red_host[g_rank] = vals[v];
    hipMemcpyAsync(g_red_dev + g_rank, g_red_host + g_rank, sizeof(double), hipMemcpyHostToDevice, g_stream);
    ncclResult_t r = ncclGroupStart();
    for (int q = 0; q < nranks && r == ncclSuccess; q++) if (ranks[q] != g_rank) r = ncclRecv(g_red_dev + ranks[q], 1, ncclDouble, ranks[q], g_comm, g_stream);
    for (int q = 0; q < nranks && r == ncclSuccess; q++) if (ranks[q] != g_rank) r = ncclSend(g_red_dev + g_rank, 1, ncclDouble, ranks[q], g_comm, g_stream);
    if (r == ncclSuccess) r = ncclGroupEnd();
    if (r != ncclSuccess) { nccl_fail(r, "scalar all-to-all"); abort(); }
    hipMemcpyAsync(g_red_host, g_red_dev, (size_t)g_size * sizeof(double), hipMemcpyDeviceToHost, g_stream);
    hipStreamSynchronize(g_stream);
    double acc = g_red_host[ranks[0]];
    for (int q = 1; q < nranks; q++) {
      const double x = g_red_host[ranks[q]];
      if (op == 0) acc = (x > acc) ? x : acc; else acc += x;   // HPGMG_REDUCE_MAX = 0, HPGMG_REDUCE_SUM = 1
    }
    vals[v] = acc;
  }
}

}  // extern "C"

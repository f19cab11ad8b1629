"""Worker for tests/test_multirank_gloo.py: one rank of a world_size-N CPU job.

Runs the host layer (level/list construction for num_ranks > 1, pack/unpack lists, per-level active
rank sets, host-driven BiCGStab) with the CPU oracle operators, the MPI replacement being a
transport whose two callbacks are implemented with torch.distributed (gloo).  The same two
callbacks are implemented with RCCL in the product (hpgmg_amd/csrc/kernels/comm_rccl.hip).
Prints one JSON line per rank.
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hpgmg_testlib import Backend, VARIANTS  # noqa: E402

c_int, c_dbl, vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p
PD, PI = ctypes.POINTER(c_dbl), ctypes.POINTER(c_int)
SENDRECV = ctypes.CFUNCTYPE(None, vp, c_int, ctypes.POINTER(PD), PI, PI, c_int, ctypes.POINTER(PD), PI, PI, c_int)
ALLREDUCE = ctypes.CFUNCTYPE(None, vp, PD, c_int, c_int, PI, c_int)
PREPARE = ctypes.CFUNCTYPE(None, vp, PI, c_int)


class Transport(ctypes.Structure):      # include/hpgmg_mg.h hpgmg_transport
    _fields_ = [("rank", c_int), ("size", c_int), ("ctx", vp), ("sendrecv", SENDRECV), ("allreduce", ALLREDUCE), ("prepare_subset", PREPARE)]


def view(ptr, n):
    return torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(n,)))


def main():
    variant, log2, per_rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    backend = sys.argv[4] if len(sys.argv) > 4 else "oracle"
    dist.init_process_group(backend="gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    stats = {"messages": 0, "doubles": 0, "allreduces": 0}
    K = None
    if backend == "hip":        # every rank of the job shares the box's one GPU; message buffers are device memory
        import hpgmg_amd as H
        K = H.load_kernels()
        assert K.hpgmg_hip_set_device(0) == 0

    def sendrecv(ctx, nrecv, rbuf, rsize, rrank, nsend, sbuf, ssize, srank, tag):
        reqs, staged = [], []
        for n in range(nrecv):
            if K is None:
                host = view(rbuf[n], rsize[n])
            else:
                host = torch.empty(rsize[n], dtype=torch.float64)
                staged.append((ctypes.cast(rbuf[n], vp), host))
            reqs.append(dist.irecv(host, src=rrank[n], tag=tag))
        for n in range(nsend):
            if K is None:
                host = view(sbuf[n], ssize[n])
            else:           # stands in for ncclSend on the launch stream: stage the packed device buffer through the host
                host = torch.empty(ssize[n], dtype=torch.float64)
                assert K.hpgmg_hip_memcpy_d2h(vp(host.data_ptr()), ctypes.cast(sbuf[n], vp), ssize[n] * 8) == 0
            reqs.append(dist.isend(host, dst=srank[n], tag=tag))
            stats["messages"] += 1; stats["doubles"] += ssize[n]
        for r in reqs:
            r.wait()
        bad = os.environ.get("HPGMG_TEST_CORRUPT_FROM")      # negative test of the self-test: what rank `bad` sent arrives with one wrong element
        if bad is not None and K is None:
            for n in range(nrecv):
                if rrank[n] == int(bad) and rsize[n] > 3:
                    rbuf[n][3] += 1.0
        for dev, host in staged:
            assert K.hpgmg_hip_memcpy_h2d(dev, vp(host.data_ptr()), host.numel() * 8) == 0

    groups = {}      # sub-communicators announced by MGBuild (every rank, same order): dist.new_group is collective over the job, like ncclCommSplit

    def prepare_subset(ctx, ranks, nranks):
        members = tuple(ranks[q] for q in range(nranks))
        stats["subsets_announced"] = stats.get("subsets_announced", 0) + 1
        if members not in groups and os.environ.get("HPGMG_TEST_SUBCOMM", "1") != "0":
            groups[members] = dist.new_group(list(members))

    def allreduce(ctx, vals, n, op, ranks, nranks):
        stats["allreduces"] += 1
        members = [ranks[q] for q in range(nranks)]
        mine = torch.tensor([vals[v] for v in range(n)], dtype=torch.float64)
        if tuple(members) in groups:      # ONE collective on the set's sub-communicator, the partials added in member order (what comm_rccl.hip does on a split communicator)
            stats["subset_collectives"] = stats.get("subset_collectives", 0) + 1
            parts = [torch.empty(n, dtype=torch.float64) for _ in members]
            dist.all_gather(parts, mine, group=groups[tuple(members)])
            for v in range(n):
                acc = parts[0][v].item()
                for q in range(1, len(members)):
                    x = parts[q][v].item()
                    acc = max(acc, x) if op == 0 else acc + x
                vals[v] = acc
            return
        if nranks < size:
            stats["subset_alltoalls"] = stats.get("subset_alltoalls", 0) + 1
        got = {rank: mine}
        reqs = []
        for r in members:
            if r != rank:
                got[r] = torch.empty(n, dtype=torch.float64)
                reqs.append(dist.irecv(got[r], src=r, tag=9999))
        for r in members:
            if r != rank:
                reqs.append(dist.isend(mine, dst=r, tag=9999))
        for q in reqs:
            q.wait()
        for v in range(n):
            acc = got[members[0]][v].item()
            for r in members[1:]:
                x = got[r][v].item()
                acc = max(acc, x) if op == 0 else acc + x
            vals[v] = acc

    be = Backend.hip() if backend == "hip" else Backend.oracle()
    ipc = backend == "hip" and os.environ.get("HPGMG_TEST_TRANSPORT") == "ipc"
    if ipc:     # the product's node-local transport: peer copies through hipIpc handles ordered by interprocess events (kernels/comm_ipc.hip); gloo only starts the job
        be.lib.hpgmg_transport_init_ipc.argtypes = [ctypes.c_char_p, c_int, c_int]
        nonce = [os.getpid() if rank == 0 else None]      # one segment per JOB: a crashed earlier run with the same port may have left its segment in /dev/shm
        dist.broadcast_object_list(nonce, src=0)
        assert be.lib.hpgmg_transport_init_ipc(("/hpgmg_test_%s_%d" % (os.environ.get("MASTER_PORT", "0"), nonce[0])).encode(), rank, size) == 0
    else:
        cb = Transport(rank, size, None, SENDRECV(sendrecv), ALLREDUCE(allreduce), PREPARE(prepare_subset))
        be.lib.hpgmg_set_transport(ctypes.byref(cb))
    # first contact, as bench.py does it: a known pattern to and from every rank, a maximum, a rank-ordered sum, a sum over ranks {0, 1}
    msg = ctypes.create_string_buffer(512)
    be.lib.hpgmg_transport_selftest.restype = c_int
    be.lib.hpgmg_transport_selftest.argtypes = [ctypes.c_char_p, c_int]
    stats["selftest"] = be.lib.hpgmg_transport_selftest(msg, 512)
    stats["selftest_message"] = msg.value.decode()
    if os.environ.get("HPGMG_TEST_CORRUPT_FROM") is not None:      # the negative test stops here: the damaged message must have been noticed
        print("RESULT " + json.dumps({"rank": rank, "stats": stats}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    assert stats["selftest"] == 0, stats["selftest_message"]
    be.configure(**VARIANTS[variant])
    s = be.solver_cli(log2, per_rank, rank=rank, ranks=size)
    norms = s.three_sizes()
    err, order = s.richardson()
    repeat = ["%1.15e" % s.fmg(0) for _ in range(3)]     # same solve again: second run is captured as hipGraphs, third replays them (HIP)
    levels = []
    for l in range(s.num_levels()):
        lv = s.level(l)
        levels.append({"dim": lv.dim, "box_dim": lv.box_dim, "my_boxes": lv.num_boxes, "active": lv.info[12]})
    if backend == "hip":
        be.lib.hpgmg_overlap_count.restype = ctypes.c_longlong
        stats["overlapped_exchanges"] = be.lib.hpgmg_overlap_count()
        be.lib.hpgmg_pair_remote_smooths.restype = ctypes.c_longlong
        stats["pair_remote_smooths"] = be.lib.hpgmg_pair_remote_smooths()      # smooth() calls run as sweep pairs across rank boundaries
        counts = (ctypes.c_longlong * 2)()
        K.hpgmg_hip_pair_launch_counts(counts)
        stats["pair_launches"], stats["pair_remote_launches"] = counts[0], counts[1]
        for name, fn in (("fv4_rb_smooths", be.lib.hpgmg_fv4_rb_smooths), ("rb27_passes", be.lib.hpgmg_rb27_passes), ("image_exchanges", be.lib.hpgmg_image_exchanges),
                         ("fused_residuals_remote", be.lib.hpgmg_fused_residuals_remote), ("interp_folded_remote", be.lib.hpgmg_interp_folded_remote)):
            fn.restype = ctypes.c_longlong          # one-pass red + black smoothers, refreshes of the images of neighbouring ranks' boxes
            stats[name] = fn()
    s.destroy()
    if ipc:
        K.hpgmg_hip_ipc_message_count.restype = ctypes.c_longlong
        stats["messages"] = K.hpgmg_hip_ipc_message_count()
        stats["transport"] = "ipc"
        dist.barrier()
        be.lib.hpgmg_transport_finalize_ipc()
    print("RESULT " + json.dumps({"rank": rank, "norms": ["%1.15e" % v for v in norms], "err": "%1.15e" % err,
                                  "order": "%0.3f" % order, "levels": levels, "stats": stats, "repeat": repeat}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

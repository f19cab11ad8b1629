"""kernels/comm_ipc.hip on its own: two (and three) processes on the box's one GPU exchange device buffers through the hipIpc peer-copy transport
and reduce scalars through its shared-memory segment -- the C-ABI entry points called directly, no solver involved."""
import ctypes
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, size, name, q):
    try:
        sys.path.insert(0, ROOT)
        if os.environ.get("HPGMG_TEST_WITH_TORCH") == "1":
            import torch  # noqa: F401  (bench.py and the multi-rank tests run with torch -- and the HIP runtime it brings -- loaded)
            torch.cuda.is_available()
        import hpgmg_amd as H
        K = H.load_kernels()
        vp, c_int, P = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER
        assert K.hpgmg_hip_set_device(0) == 0
        K.hpgmg_hip_ipc_init.argtypes = [ctypes.c_char_p, c_int, c_int]
        assert K.hpgmg_hip_ipc_init(name.encode(), rank, size) == 0, K.hpgmg_hip_last_error()
        K.hpgmg_hip_ipc_sendrecv.restype = None
        K.hpgmg_hip_ipc_sendrecv.argtypes = [vp, c_int, P(vp), P(c_int), P(c_int), c_int, P(vp), P(c_int), P(c_int), c_int]
        K.hpgmg_hip_ipc_allreduce.restype = None
        K.hpgmg_hip_ipc_allreduce.argtypes = [vp, P(ctypes.c_double), c_int, c_int, P(c_int), c_int]
        n = 4096
        peers = [p for p in range(size) if p != rank]
        d_send = {p: K.hpgmg_hip_malloc(n * 8) for p in peers}
        d_recv = {p: K.hpgmg_hip_malloc(n * 8) for p in peers}
        ok = True
        for it in range(20):                     # the same buffers again and again: READY must keep a sender from overwriting what is still being read
            for p in peers:
                src = np.arange(n, dtype=np.float64) + 1000.0 * rank + 10.0 * p + 0.001 * it
                assert K.hpgmg_hip_memcpy_h2d(vp(d_send[p]), src.ctypes.data_as(vp), n * 8) == 0
            rb = (vp * len(peers))(*[d_recv[p] for p in peers]); sb = (vp * len(peers))(*[d_send[p] for p in peers])
            sizes = (c_int * len(peers))(*[n] * len(peers)); ranks = (c_int * len(peers))(*peers)
            K.hpgmg_hip_ipc_sendrecv(None, len(peers), rb, sizes, ranks, len(peers), sb, sizes, ranks, 0x30 + it % 4)
            for p in peers:
                got = np.empty(n, dtype=np.float64)
                assert K.hpgmg_hip_memcpy_d2h(got.ctypes.data_as(vp), vp(d_recv[p]), n * 8) == 0      # stream-ordered after the exchange
                want = np.arange(n, dtype=np.float64) + 1000.0 * p + 10.0 * rank + 0.001 * it
                ok = ok and np.array_equal(got, want)
            val = (ctypes.c_double * 2)(0.1 * (rank + 1) + it, float(rank))
            allr = (c_int * size)(*range(size))
            K.hpgmg_hip_ipc_allreduce(None, val, 2, 1, allr, size)                                    # sums in rank order
            acc = 0.0
            for r in range(size):
                acc = (0.1 * (r + 1) + it) if r == 0 else acc + (0.1 * (r + 1) + it)
            ok = ok and val[0] == acc and val[1] == sum(range(size))
            mx = (ctypes.c_double * 1)(float(rank * 3 - it))
            K.hpgmg_hip_ipc_allreduce(None, mx, 1, 0, allr, size)
            ok = ok and mx[0] == float((size - 1) * 3 - it)
        K.hpgmg_hip_ipc_message_count.restype = ctypes.c_longlong
        msgs = K.hpgmg_hip_ipc_message_count()
        K.hpgmg_hip_ipc_finalize()
        q.put((rank, ok, msgs))
    except Exception as e:      # noqa: BLE001
        q.put((rank, False, repr(e)))


@pytest.mark.parametrize("size", [2, 3])
def test_peer_copies_and_ordered_reductions_between_processes_on_one_gpu(size):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/hpgmg_ipc_test_%d_%d" % (os.getpid(), size)
    procs = [ctx.Process(target=worker, args=(r, size, name, q)) for r in range(size)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        res.append(q.get(timeout=300))
    for p in procs:
        p.join(60)
    assert all(ok is True for _, ok, _ in res), res
    assert all(m == 20 * (size - 1) for _, _, m in res), res

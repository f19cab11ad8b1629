"""The RCCL transport itself (hpgmg_amd/csrc/kernels/comm_rccl.hip) on the one GPU the test box has: a size-1 communicator
whose single rank sends a packed buffer to itself through the same grouped ncclSend/ncclRecv call the halo exchange uses,
on the library's launch stream, between device-side writes and reads.  What a single GPU cannot show -- several ranks --
is covered for everything above the two RCCL calls by tests/test_gpu_multirank.py (host-staged transport)."""
import ctypes

import numpy as np
import pytest

import hpgmg_amd as H

pytestmark = pytest.mark.gpu


def test_rccl_sendrecv_to_self_and_trivial_allreduce():
    K = H.load_kernels()
    c_int, vp, P = ctypes.c_int, ctypes.c_void_p, ctypes.POINTER
    K.hpgmg_hip_rccl_unique_id.argtypes = [ctypes.c_char_p]; K.hpgmg_hip_rccl_unique_id.restype = c_int
    K.hpgmg_hip_rccl_init.argtypes = [ctypes.c_char_p, c_int, c_int]; K.hpgmg_hip_rccl_init.restype = c_int
    K.hpgmg_hip_rccl_sendrecv.argtypes = [vp, c_int, P(vp), P(c_int), P(c_int), c_int, P(vp), P(c_int), P(c_int), c_int]
    K.hpgmg_hip_rccl_sendrecv.restype = None
    K.hpgmg_hip_rccl_allreduce.argtypes = [vp, P(ctypes.c_double), c_int, c_int, P(c_int), c_int]; K.hpgmg_hip_rccl_allreduce.restype = None
    K.hpgmg_hip_rccl_finalize.restype = None
    assert K.hpgmg_hip_set_device(0) == 0
    ident = ctypes.create_string_buffer(128)
    assert K.hpgmg_hip_rccl_unique_id(ident) == 0
    assert K.hpgmg_hip_rccl_init(ident.raw, 0, 1) == 0
    try:
        n = 4 * 128 * 128                                   # one neighbour's message at config 2: 4 faces of 128^2 doubles
        src = np.arange(n, dtype=np.float64) * 0.5 - 7.0
        K.hpgmg_hip_malloc.restype = vp; K.hpgmg_hip_malloc.argtypes = [ctypes.c_size_t]
        d_s, d_r = K.hpgmg_hip_malloc(n * 8), K.hpgmg_hip_malloc(n * 8)
        assert d_s and d_r
        assert K.hpgmg_hip_memcpy_h2d(vp(d_s), src.ctypes.data_as(vp), n * 8) == 0
        rbuf, sbuf = (vp * 1)(d_r), (vp * 1)(d_s)
        sizes, ranks = (c_int * 1)(n), (c_int * 1)(0)
        K.hpgmg_hip_rccl_sendrecv(None, 1, rbuf, sizes, ranks, 1, sbuf, sizes, ranks, 0x11)
        got = np.empty(n, dtype=np.float64)
        assert K.hpgmg_hip_memcpy_d2h(got.ctypes.data_as(vp), vp(d_r), n * 8) == 0     # stream-ordered after the exchange
        assert np.array_equal(got, src)
        val = (ctypes.c_double * 1)(3.25)
        K.hpgmg_hip_rccl_allreduce(None, val, 1, 0, (c_int * 1)(0), 1)                   # one active rank: identity
        K.hpgmg_hip_rccl_allreduce_max_world.argtypes = [P(ctypes.c_double), c_int]
        assert K.hpgmg_hip_rccl_allreduce_max_world(val, 1) == 0                      # the ncclAllReduce(max) the transport issues on levels every rank shares
        assert val[0] == 3.25
        K.hpgmg_hip_rccl_allreduce_ordered_world.argtypes = [P(ctypes.c_double), c_int, c_int]
        K.hpgmg_hip_rccl_allgather_count.restype = ctypes.c_longlong
        pair = (ctypes.c_double * 2)(0.1, -2.5)
        assert K.hpgmg_hip_rccl_allreduce_ordered_world(pair, 2, 1) == 0               # the ncclAllGather + rank-ordered sum of a whole-job dot() / mean()
        assert (pair[0], pair[1]) == (0.1, -2.5) and K.hpgmg_hip_rccl_allgather_count() == 1
        K.hpgmg_hip_free(vp(d_s)); K.hpgmg_hip_free(vp(d_r))
    finally:
        K.hpgmg_hip_rccl_finalize()

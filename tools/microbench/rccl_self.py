"""Latency of one grouped ncclSend/ncclRecv exchange as the halo exchange issues it (comm_rccl.hip), measured with a
size-1 communicator sending to itself: 3 messages per exchange (the 3 neighbour ranks of an octant) of the given size.
It contains RCCL's launch + kernel overhead but no xGMI hop.  Run on the GPU box:  PYTHONPATH=. python tools/microbench/rccl_self.py"""
import ctypes, time
import hpgmg_amd as H

K = H.load_kernels()
c_int, vp, P = ctypes.c_int, ctypes.c_void_p, ctypes.POINTER
K.hpgmg_hip_rccl_unique_id.argtypes = [ctypes.c_char_p]
K.hpgmg_hip_rccl_init.argtypes = [ctypes.c_char_p, c_int, c_int]
K.hpgmg_hip_rccl_sendrecv.argtypes = [vp, c_int, P(vp), P(c_int), P(c_int), c_int, P(vp), P(c_int), P(c_int), c_int]
K.hpgmg_hip_rccl_sendrecv.restype = None
K.hpgmg_hip_malloc.restype = vp; K.hpgmg_hip_malloc.argtypes = [ctypes.c_size_t]
K.hpgmg_hip_set_device(0)
ident = ctypes.create_string_buffer(128)
assert K.hpgmg_hip_rccl_unique_id(ident) == 0 and K.hpgmg_hip_rccl_init(ident.raw, 0, 1) == 0
for n in (4 * 32 * 32, 4 * 64 * 64, 4 * 128 * 128):          # doubles per message: 4 faces of 32^2 / 64^2 / 128^2
    m = 3
    sb = (vp * m)(*[K.hpgmg_hip_malloc(n * 8) for _ in range(m)]); rb = (vp * m)(*[K.hpgmg_hip_malloc(n * 8) for _ in range(m)])
    sizes, ranks = (c_int * m)(*[n] * m), (c_int * m)(*[0] * m)
    for _ in range(20):
        K.hpgmg_hip_rccl_sendrecv(None, m, rb, sizes, ranks, m, sb, sizes, ranks, 0)
    K.hpgmg_hip_sync()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        K.hpgmg_hip_rccl_sendrecv(None, m, rb, sizes, ranks, m, sb, sizes, ranks, 0)
    t_issue = time.perf_counter() - t0
    K.hpgmg_hip_sync()
    t_all = time.perf_counter() - t0
    print(f"{m} x {n * 8 / 1024:7.0f} KiB per exchange: host issue {t_issue / reps * 1e6:6.1f} us, completed {t_all / reps * 1e6:6.1f} us per exchange")
K.hpgmg_hip_rccl_finalize()

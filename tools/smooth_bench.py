#!/usr/bin/env python3
"""Micro-benchmark of the fine-level operators of config 2 (7 8, VC Helmholtz): hipEvent time per call."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpgmg_amd as H
K = H.load_kernels(); lib = H.load_driver(); lib.hpgmg_set_verbose(0)
align = [int(x) for x in os.environ.get("ALIGN", "4,4,4,32").split(",")]
lib.hpgmg_set_box_alignment(*align)
smoother = {"cheby": H.SMOOTH_CHEBY, "gsrb": H.SMOOTH_GSRB}[os.environ.get("SMOOTHER", "cheby")]
lib.hpgmg_configure(ctypes.byref(H.Config(H.OP_7PT, smoother, 1, 1)))
log2 = int(os.environ.get("LOG2", "7"))
s = lib.hpgmg_solver_create(log2, 8, H.BC_DIRICHLET, 0, 1)
L = lib.hpgmg_solver_level(s, 0)
info = (ctypes.c_int * H.INFO_COUNT)(); lib.hpgmg_level_info(L, info)
cells = info[H.INFO_NUM_MY_BOXES] * info[H.INFO_BOX_DIM] ** 3
def timeit(fn, n=20):
    fn(); K.hpgmg_hip_sync()
    a, b = K.hpgmg_hip_event_create(), K.hpgmg_hip_event_create()
    K.hpgmg_hip_event_record(a)
    for _ in range(n): fn()
    K.hpgmg_hip_event_record(b)
    return K.hpgmg_hip_event_elapsed_ms(a, b) / n * 1e3
sweeps = 4
t = timeit(lambda: lib.smooth(L, H.VECTOR_U, H.VECTOR_F, 1.0, 1.0))
bpc = 72 if smoother == H.SMOOTH_CHEBY else 64
print(f"smooth   : {t/sweeps:8.1f} us/sweep  {bpc*cells/(t/sweeps*1e-6)/1e9:7.1f} GB/s  (jStride={info[H.INFO_JSTRIDE]})")
t = timeit(lambda: lib.residual(L, H.VECTOR_TEMP, H.VECTOR_U, H.VECTOR_F, 1.0, 1.0))
print(f"residual : {t:8.1f} us        {56*cells/(t*1e-6)/1e9:7.1f} GB/s")
L1 = lib.hpgmg_solver_level(s, 1)
t = timeit(lambda: lib.restriction(L1, H.VECTOR_R, L, H.VECTOR_TEMP, H.RESTRICT_CELL))
print(f"restrict : {t:8.1f} us        {9*cells/(t*1e-6)/1e9:7.1f} GB/s")
t = timeit(lambda: lib.interpolation_vcycle(L, H.VECTOR_U, 1.0, L1, H.VECTOR_U))
print(f"interp p0: {t:8.1f} us        {17*cells/(t*1e-6)/1e9:7.1f} GB/s")
t = timeit(lambda: lib.interpolation_fcycle(L, H.VECTOR_U, 0.0, L1, H.VECTOR_U))
print(f"interp p1: {t:8.1f} us        {17*cells/(t*1e-6)/1e9:7.1f} GB/s")
t = timeit(lambda: lib.norm(L, H.VECTOR_U))
print(f"norm     : {t:8.1f} us        {8*cells/(t*1e-6)/1e9:7.1f} GB/s")
t = timeit(lambda: lib.scale_vector(L, H.VECTOR_R, 1.0, H.VECTOR_F))
print(f"scale    : {t:8.1f} us        {16*cells/(t*1e-6)/1e9:7.1f} GB/s")
t = timeit(lambda: lib.zero_vector(L, H.VECTOR_U))
print(f"zero     : {t:8.1f} us        {8*cells/(t*1e-6)/1e9:7.1f} GB/s")
lib.hpgmg_solver_destroy(s)

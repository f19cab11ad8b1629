#!/usr/bin/env python3
"""Experiment (library built with EXTRA_HIPFLAGS=-DHPGMG_EXP_TIMELINE): where one launch of the one-box small-level smooth
(HPGMG_SMALL_FUSED=2: fv4 GSRB on an 8^3 level, image of the box in LDS) spends its time."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HPGMG_SMALL_FUSED"] = "2"
import hpgmg_amd as H
K = H.load_kernels(); lib = H.load_driver(); lib.hpgmg_set_verbose(0)
lib.hpgmg_configure(ctypes.byref(H.Config(H.OP_FV4, H.SMOOTH_GSRB, 0, 1)))
s = lib.hpgmg_solver_create(3, 1, H.BC_DIRICHLET, 0, 1)        # one box of 8^3
L = lib.hpgmg_solver_level(s, 0)
K.hpgmg_hip_malloc.restype = ctypes.c_void_p
buf = K.hpgmg_hip_malloc(256 * 8)
K.hpgmg_hip_exp_timeline.argtypes = [ctypes.c_void_p]
for _ in range(3): lib.smooth(L, H.VECTOR_U, H.VECTOR_F, 0.0, 1.0)
lib.hpgmg_operators_flush(); K.hpgmg_hip_sync()
K.hpgmg_hip_exp_timeline(buf)
lib.smooth(L, H.VECTOR_U, H.VECTOR_F, 0.0, 1.0)
lib.hpgmg_operators_flush(); K.hpgmg_hip_sync()
K.hpgmg_hip_exp_timeline(None)
host = np.zeros(256, dtype=np.uint64)
K.hpgmg_hip_memcpy_d2h(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(buf), 256 * 8)
n = int(host[255]); t = host[:n].astype(np.float64) * 0.01
print("marks", n, "total us %.1f" % (t[-1] - t[0]))
print("phases us (start, image in | per half sweep: copies, boundary, stencil | ... | image out):", np.round(np.diff(t), 2).tolist())

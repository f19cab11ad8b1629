#!/usr/bin/env python3
"""Experiment (library built with EXTRA_HIPFLAGS=-DHPGMG_EXP_TIMELINE): where one V-cycle tail launch (8^3 -> 1^3 and back) spends its time.
Marks: kernel start; per level visit: loaded (start barrier passed), swept, residual formed, visit done; per bottom solve: done."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpgmg_amd as H
K = H.load_kernels(); lib = H.load_driver(); lib.hpgmg_set_verbose(0)
lib.hpgmg_configure(ctypes.byref(H.Config(H.OP_7PT, H.SMOOTH_CHEBY, 1, 1)))
s = lib.hpgmg_solver_create(5, 8, H.BC_DIRICHLET, 0, 1)        # 64^3: levels 64, 32, 16, 8, 4, 2, 1
K.hpgmg_hip_malloc.restype = ctypes.c_void_p
buf = K.hpgmg_hip_malloc(64 * 8)
for _ in range(3): lib.hpgmg_solver_fmg(s, 0)
K.hpgmg_hip_exp_tail_timeline.argtypes = [ctypes.c_void_p]
K.hpgmg_hip_exp_tail_timeline(buf)
lib.hpgmg_solver_fmg(s, 0)                                      # the LAST tail launch of the solve (V-cycle from 64^3) leaves its record
K.hpgmg_hip_sync()
host = np.zeros(64, dtype=np.uint64)
K.hpgmg_hip_memcpy_d2h(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(buf), 64 * 8)
n = int(host[63]); t = host[:n].astype(np.float64) * 0.01
print("marks", n, "total us %.1f" % (t[-1] - t[0]))
print("intervals us:", np.round(np.diff(t), 2).tolist())

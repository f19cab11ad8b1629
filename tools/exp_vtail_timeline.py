#!/usr/bin/env python3
"""Experiment (library built with EXTRA_HIPFLAGS=-DHPGMG_EXP_TIMELINE): where one launch of the V-cycle tail below a level of one box
(small_vtail_kernel: levels of 8^3, 4^3, 2^3 cells) spends its time.  usage: exp_vtail_timeline.py [fv4|27pt|fv2] [gsrb|cheby]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HPGMG_SMALL_VTAIL"] = "1"
import hpgmg_amd as H
op = {"fv4": H.OP_FV4, "27pt": H.OP_27PT, "fv2": H.OP_FV2}[sys.argv[1] if len(sys.argv) > 1 else "fv4"]
sm = {"gsrb": H.SMOOTH_GSRB, "cheby": H.SMOOTH_CHEBY}[sys.argv[2] if len(sys.argv) > 2 else "gsrb"]
K = H.load_kernels(); lib = H.load_driver(); lib.hpgmg_set_verbose(0)
lib.hpgmg_configure(ctypes.byref(H.Config(op, sm, 0, 1)))
s = lib.hpgmg_solver_create(3, 1, H.BC_DIRICHLET, 0, 1)        # one box of 8^3
K.hpgmg_hip_malloc.restype = ctypes.c_void_p
buf = K.hpgmg_hip_malloc(256 * 8)
K.hpgmg_hip_exp_timeline.argtypes = [ctypes.c_void_p]
for _ in range(3): lib.hpgmg_solver_fmg(s, 0)
K.hpgmg_hip_sync()
K.hpgmg_hip_exp_timeline(buf)
lib.hpgmg_solver_fmg(s, 0)
K.hpgmg_hip_sync()
K.hpgmg_hip_exp_timeline(None)
host = np.zeros(256, dtype=np.uint64)
K.hpgmg_hip_memcpy_d2h(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(buf), 256 * 8)
n = int(host[255]); t = host[:n].astype(np.float64) * 0.01
print("marks", n, "total us %.1f" % (t[-1] - t[0]), "shader clock MHz %.0f" % ((float(host[254]) - float(host[253])) / (t[-1] - t[0])))
print("per level down: image, bc words, smooth, residual, restrict + write back | bottom | per level up: image, interp + bc words, smooth, write back")
print(np.round(np.diff(t), 2).tolist())

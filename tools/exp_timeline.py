#!/usr/bin/env python3
"""Experiment (needs a library built with EXTRA_HIPFLAGS=-DHPGMG_EXP_TIMELINE): per-step timeline of one wave of the sweep-pair kernel."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpgmg_amd as H
K = H.load_kernels(); lib = H.load_driver(); lib.hpgmg_set_verbose(0)
lib.hpgmg_configure(ctypes.byref(H.Config(H.OP_7PT, H.SMOOTH_CHEBY, 1, 1)))
log2 = int(os.environ.get("LOG2", "7"))
s = lib.hpgmg_solver_create(log2, 8, H.BC_DIRICHLET, 0, 1)
L = lib.hpgmg_solver_level(s, 0)
K.hpgmg_hip_malloc.restype = ctypes.c_void_p
buf = K.hpgmg_hip_malloc(4096 * 8)
K.hpgmg_hip_exp_timeline.argtypes = [ctypes.c_void_p]
for _ in range(3): lib.smooth(L, H.VECTOR_U, H.VECTOR_F, 1.0, 1.0)
K.hpgmg_hip_sync()
K.hpgmg_hip_exp_timeline(buf)
lib.smooth(L, H.VECTOR_U, H.VECTOR_F, 1.0, 1.0)      # two pair launches: the second overwrites the first's record
K.hpgmg_hip_sync()
K.hpgmg_hip_exp_timeline(None)
host = np.zeros(4096, dtype=np.uint64)
K.hpgmg_hip_memcpy_d2h(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(buf), 4096 * 8)
n = int(host[4095]); t = host[:n].astype(np.float64) * 0.01    # 100 MHz -> us
steps = n // 5
d = np.diff(t.reshape(steps, 5), axis=1)
gap = t.reshape(steps, 5)[1:, 0] - t.reshape(steps, 5)[:-1, 4]
print(f"steps {steps}; per step us: loads {d[:,0].mean():.2f}  x1 {d[:,1].mean():.2f}  x2+store {d[:,2].mean():.2f}  lds+barrier {d[:,3].mean():.2f}  loop {gap.mean():.2f}  total {(t[-1]-t[0])/steps:.2f}")
print("first steps:", np.round(d[:4], 2).tolist())

#!/usr/bin/env python3
"""Experiment (library built with EXTRA_HIPFLAGS=-DHPGMG_EXP_TIMELINE): where one workgroup of the fv4 red + black kernel spends a step.
Wave 0 and the last (ring) wave of a workgroup in the middle of the grid record the 100 MHz clock at nine points of every step."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpgmg_amd as H
K = H.load_kernels(); lib = H.load_driver(); lib.hpgmg_set_verbose(0)
lib.hpgmg_configure(ctypes.byref(H.Config(H.OP_FV4, H.SMOOTH_GSRB, 0, 1)))
log2 = int(os.environ.get("LOG2", "7")); boxes = int(os.environ.get("BOXES", "64"))
s = lib.hpgmg_solver_create(log2, boxes, H.BC_DIRICHLET, 0, 1)
L = lib.hpgmg_solver_level(s, 0)
K.hpgmg_hip_malloc.restype = ctypes.c_void_p
NB = 16384 + 3 * 8192
buf = K.hpgmg_hip_malloc(NB * 8)
K.hpgmg_hip_exp_timeline_fv4.argtypes = [ctypes.c_void_p]
lib.hpgmg_smooth_in_cycle.restype = ctypes.c_int
lib.hpgmg_smooth_in_cycle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
for _ in range(2): lib.hpgmg_smooth_in_cycle(L, H.VECTOR_U, H.VECTOR_F, 0.0, 1.0)
K.hpgmg_hip_sync()
K.hpgmg_hip_memset0.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
K.hpgmg_hip_memset0(buf, NB * 8)
K.hpgmg_hip_exp_timeline_fv4(buf)
lib.hpgmg_smooth_in_cycle(L, H.VECTOR_U, H.VECTOR_F, 0.0, 1.0)      # three passes: each overwrites the record of the one before
K.hpgmg_hip_sync()
K.hpgmg_hip_exp_timeline_fv4(None)
host = np.zeros(NB, dtype=np.uint64)
K.hpgmg_hip_memcpy_d2h(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(buf), NB * 8)
names = ["barrier A", "issue loads", "R own", "R ring", "barrier B", "BC", "barrier C", "B", "pd", "barrier D", "wait loads + LDS stores"]
for wsel in range(8):
    label = f"wave {wsel}"
    t = host[wsel * 2048: wsel * 2048 + 2040]
    n = int(np.count_nonzero(t)); steps = n // 12
    if steps < 4:
        print(label, "no record", n); continue
    tt = t[:steps * 12].astype(np.float64).reshape(steps, 12) * 0.01    # us
    d = np.diff(tt, axis=1)[4:-4]                                       # steady state
    tot = (tt[-4, 0] - tt[4, 0]) / (steps - 8)
    print(f"{label}: {steps} steps, {tot:.2f} us per step; " + "  ".join(f"{nm} {v:.2f}" for nm, v in zip(names, d.mean(axis=0))) + f"  loop {tot - d.mean(axis=0).sum():.2f}")

wg = host[16384:].reshape(8192, 3).astype(np.float64) * 0.01
wg = wg[wg[:, 0] > 0]
t0 = wg[:, 0].min()
start, first, end = wg[:, 0] - t0, wg[:, 1] - wg[:, 0], wg[:, 2] - wg[:, 0]
print(f"{len(wg)} workgroups; launch span {wg[:, 2].max() - t0:.0f} us; workgroup duration min/mean/max {end.min():.0f}/{end.mean():.0f}/{end.max():.0f} us; prologue + first step min/mean/max {first.min():.0f}/{first.mean():.0f}/{first.max():.0f} us")
order = np.argsort(start)
print("start times (us) of every 64th workgroup in start order:", np.round(start[order][::64], 0).tolist())
print("durations of the same:", np.round(end[order][::64], 0).tolist())

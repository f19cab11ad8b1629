#!/usr/bin/env python3
"""Experiment: GPU time of one in-cycle smooth() (4 Chebyshev sweeps) on a level of 2^LOG2 boxes^(1/3) cells, for whatever HPGMG_* switches
the environment sets (they are read once per process, so one process per setting).  Prints the mean over REPS calls (hipEvents)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpgmg_amd as H
K = H.load_kernels(); lib = H.load_driver(); lib.hpgmg_set_verbose(0)
op = {"7pt": H.OP_7PT, "fv4": H.OP_FV4, "27pt": H.OP_27PT}[os.environ.get("OP", "7pt")]
sm = H.SMOOTH_CHEBY if os.environ.get("SMOOTHER", "cheby") == "cheby" else H.SMOOTH_GSRB
lib.hpgmg_configure(ctypes.byref(H.Config(op, sm, 1 if op == H.OP_7PT else 0, 1)))
log2 = int(os.environ.get("LOG2", "6")); boxes = int(os.environ.get("BOXES", "8")); level = int(os.environ.get("LEVEL", "0"))
reps = int(os.environ.get("REPS", "50"))
s = lib.hpgmg_solver_create(log2, boxes, H.BC_DIRICHLET, 0, 1)
L = lib.hpgmg_solver_level(s, level)
lib.hpgmg_smooth_in_cycle.restype = ctypes.c_int
lib.hpgmg_smooth_in_cycle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
a, b = (1.0, 1.0) if op == H.OP_7PT else (0.0, 1.0)
for _ in range(5): lib.hpgmg_smooth_in_cycle(L, H.VECTOR_U, H.VECTOR_F, a, b)
K.hpgmg_hip_sync()
e0, e1 = K.hpgmg_hip_event_create(), K.hpgmg_hip_event_create()
K.hpgmg_hip_event_record(e0)
for _ in range(reps): lib.hpgmg_smooth_in_cycle(L, H.VECTOR_U, H.VECTOR_F, a, b)
K.hpgmg_hip_event_record(e1)
ms = K.hpgmg_hip_event_elapsed_ms(e0, e1)
print(f"LOG2={log2} BOXES={boxes} LEVEL={level}: {1000.0 * ms / reps:.1f} us per smooth()  ", {k: v for k, v in os.environ.items() if k.startswith('HPGMG_')})

#!/usr/bin/env python3
"""Stress: the same V-cycle from a 64^3 level (bricks: one launch per leg, everything between workgroups as records) repeated N times on the same input --
every repetition must leave the same bytes on every level (a lost / torn / stale record shows as a difference).  usage: python tools/stress_bricks.py [N] [variant]"""
import ctypes, hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpgmg_amd as H
from hpgmg_testlib import VARIANTS, seeded_field, Backend, Level
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
variant = sys.argv[2] if len(sys.argv) > 2 else "7pt-cheby-helm"
hip = Backend.hip()
bad = 0
for geom in ((2, 32), (4, 16), (1, 64), (2, 16)):
    hip.configure(**VARIANTS[variant])
    fine = hip.level(*geom)
    for vid in range(fine.num_vectors):
        d = seeded_field(fine, 2100 + vid)
        if vid >= H.VECTOR_DINV: d = np.abs(d) + 0.5
        fine.write_all(vid, d)
    for vid in range(H.VECTOR_DINV, fine.num_vectors): hip.lib.exchange_boundary(fine.ptr, vid, H.STENCIL_SHAPE_BOX)
    a, b = (1.0, 1.0) if "helm" in variant else (0.0, 1.0)
    mg = hip.lib.hpgmg_mg_create(fine.ptr, a, b, 1)
    hip.lib.rebuild_operator(fine.ptr, None, a, b)
    hip.lib.MGVCycle.restype = None
    hip.lib.MGVCycle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
    hip.lib.hpgmg_mg_num_levels.restype = ctypes.c_int
    hip.lib.hpgmg_vcycle_legs_fused.restype = ctypes.c_int
    hip.lib.hpgmg_vcycle_legs_fused.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
    n = hip.lib.hpgmg_mg_num_levels(mg)
    lv = lambda l: Level(hip, hip.lib.hpgmg_mg_level(mg, l))
    inputs = [[seeded_field(lv(l), 2200 + 10 * l + s) for s in (31, 32, 33)] for l in range(n)]
    chain = (ctypes.c_void_p * n)(*[hip.lib.hpgmg_mg_level(mg, l) for l in range(n)])
    want = {}
    for mode in ("vcycle", "fstep"):
        for rep in range(N):
            for l in range(n):
                x = lv(l)
                for vid, d in zip((H.VECTOR_U, H.VECTOR_F, H.VECTOR_TEMP), inputs[l]): x.write_all(vid, d)
            if mode == "vcycle": hip.lib.MGVCycle(mg, H.VECTOR_U, H.VECTOR_F, a, b, 0)
            elif not hip.lib.hpgmg_vcycle_legs_fused(chain, n, H.VECTOR_U, H.VECTOR_F, a, b, 6): break
            hip.lib.hpgmg_operators_flush()
            h = hashlib.sha256()
            for l in range(n):
                for vid in (H.VECTOR_U, H.VECTOR_F, H.VECTOR_TEMP): h.update(np.ascontiguousarray(lv(l).interior(vid)).tobytes())
            d = h.hexdigest()
            if mode not in want: want[mode] = d
            elif d != want[mode]: bad += 1; print("DIFFERENT", variant, geom, mode, "repetition", rep)
    print(variant, geom, "ok" if not bad else "MISMATCHES so far: %d" % bad, flush=True)
    hip.lib.hpgmg_mg_destroy(mg); fine.destroy()
sys.exit(1 if bad else 0)

"""experiment: time apply_BCs_v4 on a 64-box level (one MI355X)"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import hpgmg_amd as H
from hpgmg_testlib import Backend, VARIANTS, seeded_field
b = Backend.hip()
b.configure(**VARIANTS["fv4-gsrb"])
b.lib.apply_BCs_v4.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
for boxes, dim in ((4, 16), (4, 32), (4, 64), (4, 128)):
    L = b.level(boxes, dim)
    L.write_all(H.VECTOR_U, seeded_field(L, 5)) if dim <= 32 else None
    for shape in (H.STENCIL_SHAPE_NO_CORNERS,):
        for _ in range(20): b.lib.apply_BCs_v4(L.ptr, H.VECTOR_U, shape)
        b.lib.hpgmg_hip_sync()
        t = time.perf_counter()
        for _ in range(300): b.lib.apply_BCs_v4(L.ptr, H.VECTOR_U, shape)
        b.lib.hpgmg_hip_sync()
        print(f"only={os.environ.get('EXP_BC_ONLY','0')} slabs={os.environ.get('EXP_BC_SLABS','auto')} boxes {boxes}^3 x {dim}^3: {(time.perf_counter() - t) / 300 * 1e6:.1f} us per call", flush=True)
    L.destroy()

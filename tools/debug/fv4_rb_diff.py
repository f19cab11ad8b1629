#!/usr/bin/env python3
"""Where does the one-pass fv4 red + black kernel differ from the oracle?  Runs on the GPU box:
   python3 tools/debug/fv4_rb_diff.py <boxes_in_i> <box_dim> <sweeps> [periodic]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import hpgmg_amd as H
from hpgmg_testlib import Backend, VARIANTS, seeded_field

def main():
    nb, bd, sweeps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    bc = H.BC_PERIODIC if len(sys.argv) > 4 else H.BC_DIRICHLET
    hip, ora = Backend.hip(), Backend.oracle()
    lv = []
    for be in (hip, ora):
        be.configure(**VARIANTS["fv4-gsrb"])
        be.lib.hpgmg_set_smooth_sweeps.argtypes = [ctypes.c_int]
        be.lib.hpgmg_set_smooth_sweeps(sweeps)
        lv.append(be.level(nb, bd, bc=bc))
    lh, lo = lv
    rng_scale = 1e-3            # keep the iteration tame so that differences stay local
    for vid in range(lh.num_vectors):
        data = seeded_field(lh, 1700 + vid)
        if vid in (H.VECTOR_DINV,):
            data = (np.abs(data) + 0.5) * rng_scale
        elif vid in (H.VECTOR_BETA_I, H.VECTOR_BETA_J, H.VECTOR_BETA_K, H.VECTOR_ALPHA):
            data = np.abs(data) + 0.5
        lh.write_all(vid, data); lo.write_all(vid, data)
    for l in (lh, lo):
        for vid in (H.VECTOR_BETA_I, H.VECTOR_BETA_J, H.VECTOR_BETA_K, H.VECTOR_DINV):
            l.b.lib.exchange_boundary(l.ptr, vid, H.STENCIL_SHAPE_BOX)
    hip.lib.hpgmg_smooth_in_cycle.restype = ctypes.c_int
    hip.lib.hpgmg_smooth_in_cycle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    K = H.load_kernels(); K.hpgmg_hip_rb_fv4_launch_count.restype = ctypes.c_longlong
    n0 = K.hpgmg_hip_rb_fv4_launch_count()
    hip.lib.hpgmg_smooth_in_cycle(lh.ptr, H.VECTOR_U, H.VECTOR_F, 0.0, 1.0)
    print("rb launches:", K.hpgmg_hip_rb_fv4_launch_count() - n0)
    ora.lib.smooth(lo.ptr, H.VECTOR_U, H.VECTOR_F, 0.0, 1.0)
    g, d = lh.ghosts, lh.box_dim
    tot = 0
    for b in range(lh.num_boxes):
        a = lh.read(b, H.VECTOR_U)[g:g + d, g:g + d, g:g + d]; o = lo.read(b, H.VECTOR_U)[g:g + d, g:g + d, g:g + d]
        bad = np.argwhere(a != o)
        tot += len(bad)
        if len(bad):
            lo_, hi_ = bad.min(0), bad.max(0)
            par = (bad.sum(1) & 1)
            print(f"box {b} low={lh.box_low(b)}: {len(bad)} of {d**3} cells differ; k {lo_[0]}..{hi_[0]} j {lo_[1]}..{hi_[1]} i {lo_[2]}..{hi_[2]}; parity0 {np.sum(par == 0)} parity1 {np.sum(par == 1)}")
            ks = np.bincount(bad[:, 0], minlength=d); js = np.bincount(bad[:, 1], minlength=d); is_ = np.bincount(bad[:, 2], minlength=d)
            print("   per k:", ks.tolist()); print("   per j:", js.tolist()); print("   per i:", is_.tolist())
            for (k, j, i) in bad[:6]:
                print(f"   ({k},{j},{i}) hip {a[k, j, i]!r} oracle {o[k, j, i]!r}")
        else:
            print(f"box {b}: identical")
    print("TOTAL differing cells:", tot)
    if os.environ.get("DIFF_DUMP"):
        out = {}
        for b in range(lh.num_boxes):
            a = lh.read(b, H.VECTOR_U)[g:g + d, g:g + d, g:g + d]; o = lo.read(b, H.VECTOR_U)[g:g + d, g:g + d, g:g + d]
            out[f"bad{b}"] = np.argwhere(a != o).astype(np.int16)
        np.savez_compressed(os.environ["DIFF_DUMP"], **out)

main()

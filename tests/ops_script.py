"""The single-operator script of oracle/op_harness.c, replayed through this repository's operator API.

oracle/op_harness.c drives the REFERENCE's operator layer one operator at a time and dumps what each leaves (whole padded boxes and the
returned scalars); tests/golden/make_ops_golden.py condenses those dumps into tests/golden/ops_golden.json.  `replay()` below makes the same
calls, in the same order, on a `Backend` of hpgmg_testlib (the CPU restatement or the HIP plugin) and returns the same records, so a test is
"replay, then compare every record with the fixture":
    * sha_full      sha256 over the bytes of the whole padded vector, all boxes (interior, ghost zones, row padding),
    * sha_interior  the same over the interior cells only (what the ghost-free HIP launches guarantee: they do not refresh an operand's
                    ghost zones),
    * absmax        max |value| over the interior (diagnostic: says roughly WHERE two runs part when a hash differs),
    * scalars       repr-exact ("%.17g").
"""
import hashlib

import numpy as np

import hpgmg_amd as H

V = dict(TEMP=H.VECTOR_TEMP, U=H.VECTOR_U, F=H.VECTOR_F, E=H.VECTOR_E, R=H.VECTOR_R, DINV=H.VECTOR_DINV,
         BETA_I=H.VECTOR_BETA_I, BETA_J=H.VECTOR_BETA_J, BETA_K=H.VECTOR_BETA_K, ALPHA=H.VECTOR_ALPHA)
RESTRICT_CELL, RESTRICT_FACE_I, RESTRICT_FACE_J, RESTRICT_FACE_K = 0, 1, 2, 3
SHAPE_BOX = 0

GEOMETRIES = [(1, 16), (2, 8)]           # the 16^3 problem as one box and as 2 x 2 x 2 boxes of 8^3
# The sizes at which the BANDWIDTH-BOUND kernels run (they only take boxes of side 128 m resp. 64 m): 2 x 2 x 2 boxes of 128^3 = BASELINE config 2's level for
# the sweep-pair / wide / fused-residual kernels of the 7-point plugin, 2 x 2 x 2 boxes of 64^3 for the one-pass red + black and the LDS-tiled kernels of the
# fv4 and 27-point plugins.  Hashes only, so the fixture stays small; the reference harness needs seconds of CPU for them.
LARGE_CASES = [("7pt-cheby-helm", 2, 128), ("7pt-gsrb", 2, 128), ("fv4-gsrb", 2, 64), ("27pt-gsrb", 2, 64), ("fv4-cheby", 2, 64), ("27pt-cheby", 2, 64)]
HARNESS_VARIANTS = ["7pt-cheby", "7pt-gsrb", "7pt-cheby-helm", "7ptcc-cheby", "7pt-jacobi", "27pt-cheby", "27pt-gsrb", "fv4-gsrb", "fv4-cheby", "fv2-cheby"]


def interior_of(arr, geom):
    """arr: (boxes, volume) -> (boxes, dim, dim, dim) interior cells."""
    d, g, jS, kS = geom["box_dim"], geom["ghosts"], geom["jStride"], geom["kStride"]
    w = d + 2 * g
    a = arr[:, : w * kS].reshape(-1, w, kS)[:, :, : w * jS].reshape(-1, w, w, jS)
    return np.ascontiguousarray(a[:, g:g + d, g:g + d, g:g + d])


def record_of(name, level, vid, arr, geom):
    inner = interior_of(arr, geom)
    return {"name": name, "level": level, "id": vid,
            "sha_full": hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest(),
            "sha_interior": hashlib.sha256(inner.tobytes()).hexdigest(),
            "absmax": "%.17g" % float(np.max(np.abs(inner)))}


def parse_harness_file(path, keep_data=True):
    """What oracle/op_harness.c wrote -> (config, geoms, records, scalars); records keep their arrays under 'data' (keep_data).  Streamed: a dump of the
    256^3 problem is several GB."""
    config, geoms, records, scalars = None, {}, [], {}
    with open(path, "rb") as f:
        while True:
            line = f.readline().decode().rstrip("\n")
            if line == "END":
                break
            tok = line.split()
            if tok[0] == "CONFIG":
                config = {"boxes_in_i": int(tok[1]), "box_dim": int(tok[2]), "a": float(tok[3]), "b": float(tok[4]), "radius": int(tok[5]), "shape": int(tok[6]), "vectors": int(tok[7])}
            elif tok[0] == "GEOM":
                geoms[int(tok[1])] = {"dim": int(tok[2]), "box_dim": int(tok[3]), "ghosts": int(tok[4]), "jStride": int(tok[5]), "kStride": int(tok[6]), "volume": int(tok[7]), "boxes": int(tok[8])}
            elif tok[0] == "SCALAR":
                scalars[tok[1]] = tok[2]
            elif tok[0] == "DUMP":
                name, level, vid, nb, vol = tok[1], int(tok[2]), int(tok[3]), int(tok[4]), int(tok[5])
                arr = np.fromfile(f, dtype=np.float64, count=nb * vol).reshape(nb, vol)
                f.read(1)
                rec = record_of(name, level, vid, arr, geoms[level])
                if keep_data:
                    rec["data"] = arr
                records.append(rec)
            else:
                raise ValueError("unexpected line in harness output: " + line[:60])
    return config, geoms, records, scalars


def replay(be, variant, boxes_in_i, box_dim, keep_data=False):
    """The script of oracle/op_harness.c on backend `be`; returns (geoms, records, scalars)."""
    from hpgmg_testlib import VARIANTS
    cfg = VARIANTS[variant]
    be.configure(**cfg)
    a, b = (1.0, 1.0) if cfg["helmholtz"] else (0.0, 1.0)
    s = be.solver(boxes_in_i, box_dim)
    lib = be.lib
    L0, L1 = s.level(0), s.level(1)
    lv = {0: L0, 1: L1}
    geoms = {l: {"dim": x.dim, "box_dim": x.box_dim, "ghosts": x.ghosts, "jStride": x.jStride, "kStride": x.kStride, "volume": x.volume, "boxes": x.num_boxes} for l, x in lv.items()}
    shape = lib.stencil_get_shape()
    records, scalars = [], {}

    def dump(name, level, vid):
        arr = lv[level].read_all(vid)
        rec = record_of(name, level, vid, arr, geoms[level])
        if keep_data:
            rec["data"] = arr
        records.append(rec)

    def scalar(name, value):
        scalars[name] = "%.17g" % value

    p0, p1 = L0.ptr, L1.ptr
    dump("setup.beta_i", 0, V["BETA_I"]); dump("setup.beta_j", 0, V["BETA_J"]); dump("setup.beta_k", 0, V["BETA_K"])
    if cfg["helmholtz"]:
        dump("setup.alpha", 0, V["ALPHA"])
    dump("setup.f", 0, V["F"]); dump("setup.dinv", 0, V["DINV"])
    scalar("setup.eig0", L0.eigenvalue)
    dump("setup.dinv1", 1, V["DINV"]); dump("setup.beta_i1", 1, V["BETA_I"])
    scalar("setup.eig1", L1.eigenvalue)

    lib.zero_vector(p0, V["U"])
    lib.smooth(p0, V["U"], V["F"], a, b)
    dump("first.smooth.u", 0, V["U"]); dump("first.smooth.temp", 0, V["TEMP"])
    scalar("first.norm_u", lib.norm(p0, V["U"]))
    lib.residual(p0, V["TEMP"], V["U"], V["F"], a, b)
    dump("first.residual", 0, V["TEMP"])
    scalar("first.norm_res", lib.norm(p0, V["TEMP"]))

    lib.random_vector(p0, V["U"])
    lib.scale_vector(p0, V["U"], 0.001, V["U"])
    lib.add_vectors(p0, V["U"], 1.0, V["U"], 0.0001, V["F"])
    dump("field.u", 0, V["U"])
    lib.exchange_boundary(p0, V["U"], shape)
    dump("exchange.u", 0, V["U"])
    lib.apply_BCs(p0, V["U"], shape)
    dump("bcs.u", 0, V["U"])
    lib.exchange_boundary(p0, V["U"], SHAPE_BOX)
    lib.apply_BCs(p0, V["U"], SHAPE_BOX)
    dump("bcs_box.u", 0, V["U"])
    lib.smooth(p0, V["U"], V["F"], a, b)
    dump("smooth.u", 0, V["U"]); dump("smooth.temp", 0, V["TEMP"])
    lib.residual(p0, V["R"], V["U"], V["F"], a, b)
    dump("residual.r", 0, V["R"]); dump("residual.u", 0, V["U"])
    lib.apply_op(p0, V["E"], V["U"], a, b)
    dump("apply_op.e", 0, V["E"])
    scalar("norm_r", lib.norm(p0, V["R"]))
    scalar("dot_u_f", lib.dot(p0, V["U"], V["F"]))
    scalar("mean_u", lib.mean(p0, V["U"]))

    lib.restriction(p1, V["R"], p0, V["R"], RESTRICT_CELL)
    dump("restrict.cell", 1, V["R"])
    lib.restriction(p1, V["E"], p0, V["BETA_I"], RESTRICT_FACE_I)
    dump("restrict.face_i", 1, V["E"])
    lib.restriction(p1, V["U"], p0, V["BETA_J"], RESTRICT_FACE_J)
    dump("restrict.face_j", 1, V["U"])
    lib.restriction(p1, V["TEMP"], p0, V["BETA_K"], RESTRICT_FACE_K)
    dump("restrict.face_k", 1, V["TEMP"])

    lib.interpolation_vcycle(p0, V["U"], 1.0, p1, V["R"])
    dump("interp_v.u", 0, V["U"]); dump("interp_v.coarse", 1, V["R"])
    lib.zero_vector(p0, V["E"])
    lib.interpolation_fcycle(p0, V["E"], 0.0, p1, V["R"])
    dump("interp_f.e", 0, V["E"]); dump("interp_f.coarse", 1, V["R"])

    lib.mul_vectors(p0, V["TEMP"], 2.0, V["U"], V["F"])
    lib.invert_vector(p0, V["E"], 1.0, V["DINV"])
    lib.shift_vector(p0, V["R"], V["R"], 0.5)
    dump("blas.mul", 0, V["TEMP"]); dump("blas.invert", 0, V["E"]); dump("blas.shift", 0, V["R"])
    scalar("error_u_e", lib.error(p0, V["U"], V["E"]))

    lib.zero_vector(p1, V["U"])
    lib.smooth(p1, V["U"], V["R"], a, b)
    dump("coarse.smooth.u", 1, V["U"])
    s.destroy()
    return geoms, records, scalars


def replay_cycle_forms(be, variant, boxes_in_i, box_dim):
    """HIP plugin only: what the CYCLE uses instead of the exported operators -- smooth() in its in-cycle form (sweep pairs without the x3 store, the one-pass
    red + black kernels), residual + norm and residual + restriction + zero_vector as single passes -- on the inputs of the harness script, recorded under the
    names of the reference's records they must equal (interiors: these forms do not refresh ghost zones, and leave VECTOR_TEMP unspecified)."""
    import ctypes
    from hpgmg_testlib import VARIANTS
    cfg = VARIANTS[variant]
    be.configure(**cfg)
    a, b = (1.0, 1.0) if cfg["helmholtz"] else (0.0, 1.0)
    s = be.solver(boxes_in_i, box_dim)
    lib = be.lib
    L0, L1 = s.level(0), s.level(1)
    geoms = {l: {"dim": x.dim, "box_dim": x.box_dim, "ghosts": x.ghosts, "jStride": x.jStride, "kStride": x.kStride, "volume": x.volume, "boxes": x.num_boxes} for l, x in {0: L0, 1: L1}.items()}
    vp, c_int, c_dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.hpgmg_smooth_in_cycle.restype = c_int
    lib.hpgmg_smooth_in_cycle.argtypes = [vp, c_int, c_int, c_dbl, c_dbl]
    lib.hpgmg_residual_restrict_zero_fused.restype = c_int
    lib.hpgmg_residual_restrict_zero_fused.argtypes = [vp, c_int, vp, c_int, c_int, c_dbl, c_dbl, c_int]
    lib.hpgmg_residual_norm_fused.restype = c_int
    lib.hpgmg_residual_norm_fused.argtypes = [vp, c_int, c_int, c_int, c_dbl, c_dbl, ctypes.POINTER(c_dbl)]
    p0, p1 = L0.ptr, L1.ptr
    records, scalars, taken = [], {}, {}
    lib.random_vector(p0, V["U"])
    lib.scale_vector(p0, V["U"], 0.001, V["U"])
    lib.add_vectors(p0, V["U"], 1.0, V["U"], 0.0001, V["F"])
    taken["smooth_in_cycle"] = lib.hpgmg_smooth_in_cycle(p0, V["U"], V["F"], a, b)
    if not taken["smooth_in_cycle"]:
        lib.smooth(p0, V["U"], V["F"], a, b)
    records.append(record_of("smooth.u", 0, V["U"], L0.read_all(V["U"]), geoms[0]))
    out = c_dbl(0.0)
    taken["residual_norm_fused"] = lib.hpgmg_residual_norm_fused(p0, -1, V["U"], V["F"], a, b, ctypes.byref(out))
    if taken["residual_norm_fused"]:
        scalars["norm_r"] = "%.17g" % out.value
    taken["residual_restrict_zero_fused"] = lib.hpgmg_residual_restrict_zero_fused(p1, V["R"], p0, V["U"], V["F"], a, b, V["U"])
    if taken["residual_restrict_zero_fused"]:
        records.append(record_of("restrict.cell", 1, V["R"], L1.read_all(V["R"]), geoms[1]))
    s.destroy()
    return records, scalars, taken

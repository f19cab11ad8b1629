"""The lazy operator queue of the HIP plugin (hpgmg_amd/csrc/host/operators_hip.c, "lazy void operators") against itself switched off.

The queue postpones void operators while they follow the call order of the reference's MGVCycle / FMGSolve / BiCGStab and issues them in fused
forms.  A caller is free to use any other order, so this test feeds it RANDOM operator sequences -- cycle-shaped runs that break off anywhere, wrong
vector ids, repeated and interleaved levels, scalars in between -- once with the queue on and once with HPGMG_LAZY=0 (every operator issued when it
is called), on two identical hierarchies, and requires every scalar and, after every sequence, the interior of EVERY vector of EVERY level to be
equal byte for byte.  200 sequences per plugin."""
import ctypes
import os

import numpy as np
import pytest

import hpgmg_amd as H
from hpgmg_testlib import VARIANTS, seeded_field
from ops_script import interior_of

pytestmark = pytest.mark.gpu
T, U, F, E, R = H.VECTOR_TEMP, H.VECTOR_U, H.VECTOR_F, H.VECTOR_E, H.VECTOR_R


def random_sequence(rng, nlev):
    """A list of (name, args) over levels 0 .. nlev-1 (level indices instead of pointers)."""
    seq = []
    a, b = 0.0, 1.0
    while len(seq) < 40:
        kind = rng.integers(0, 10)
        l = int(rng.integers(0, nlev - 1))
        if kind <= 3:                       # a stretch of MGVCycle's down leg from level l (mg.c:1147-1153), cut anywhere, sometimes with a wrong id
            depth = int(rng.integers(1, nlev - l))
            ops = []
            for q in range(l, l + depth):
                res_id = T if rng.random() < 0.85 else R
                ops += [("smooth", (q, U, R, a, b)), ("residual", (q, res_id, U, R, a, b)), ("restriction", (q + 1, R, q, res_id)), ("zero_vector", (q + 1, U))]
            seq += ops[: int(rng.integers(1, len(ops) + 1))]
        elif kind <= 5:                     # a stretch of the up leg (mg.c:1160-1161)
            top = int(rng.integers(0, nlev - 1))
            ops = []
            for q in range(nlev - 2, top - 1, -1):
                ops += [("interpolation_vcycle", (q, U, 1.0 if rng.random() < 0.9 else 0.5, q + 1, U)), ("smooth", (q, U, R, a, b))]
            start = int(rng.integers(0, len(ops)))
            seq += ops[start: start + int(rng.integers(1, len(ops) - start + 1))]
        elif kind == 6:                     # how FMGSolve starts (mg.c:1266-1277), or just a copy
            seq.append(("scale_vector", (l, R, 1.0 if rng.random() < 0.7 else 0.25, F)))
            if rng.random() < 0.7:
                seq.append(("restriction", (l + 1, R, l, R)))
        elif kind == 7:                     # the convergence check (mg.c:1321-1323), or a lone residual
            seq.append(("residual", (l, T, U, F, a, b)))
            if rng.random() < 0.6:
                seq.append(("norm", (l, T)))
        elif kind == 8:                     # BLAS-1 and scalars as a Krylov solver issues them, on any level (the small-level groups form on the last ones)
            q = int(rng.integers(0, nlev))
            for _ in range(int(rng.integers(1, 5))):
                c = int(rng.integers(0, 4))
                if c == 0: seq.append(("add_vectors", (q, E, 1.0, U, -0.5, R)))
                elif c == 1: seq.append(("mul_vectors", (q, E, 1.0, H.VECTOR_DINV, R)))
                elif c == 2: seq.append(("apply_op", (q, T, E, a, b)))
                else: seq.append(("dot", (q, E, R)))
            if rng.random() < 0.5:
                seq.append(("norm", (q, E)))
        else:
            seq.append((["zero_vector", "norm", "smooth"][int(rng.integers(0, 3))], None))
            name = seq.pop()[0]
            q = int(rng.integers(0, nlev))
            seq.append({"zero_vector": ("zero_vector", (q, E)), "norm": ("norm", (q, U)), "smooth": ("smooth", (q, E, F, a, b))}[name])
    return seq


def run(lib, levels, seq):
    out = []
    for name, args in seq:
        fn = getattr(lib, name)
        if name in ("restriction",):
            lc, idc, lf, idf = args
            fn(levels[lc].ptr, idc, levels[lf].ptr, idf, 0)
        elif name == "interpolation_vcycle":
            lf, idf, pre, lc, idc = args
            fn(levels[lf].ptr, idf, pre, levels[lc].ptr, idc)
        else:
            r = fn(levels[args[0]].ptr, *args[1:])
            if name in ("norm", "dot"):
                out.append(r)
    lib.hpgmg_operators_flush()
    return out


@pytest.mark.parametrize("variant,geom", [("7pt-cheby-helm", (2, 16)), ("7pt-gsrb", (2, 16)), ("fv4-gsrb", (1, 16)), ("27pt-gsrb", (2, 8)), ("fv2-cheby", (1, 16)),
                                          ("7pt-cheby", (2, 128)), ("7pt-cheby-helm", (2, 64))])      # (2, 64): the tile-kernel forms (interpolation fold, one-launch down leg) and the plain ones below
def test_random_operator_sequences_with_and_without_the_queue(hip, variant, geom):
    lib = hip.lib
    lib.hpgmg_set_lazy.argtypes = [ctypes.c_int]
    hip.configure(**VARIANTS[variant])
    big, mid = geom[1] >= 128, geom[1] == 64
    sa, sb = hip.solver(*geom), hip.solver(*geom)
    try:
        nlev = sa.num_levels()
        la, lb = [sa.level(l) for l in range(nlev)], [sb.level(l) for l in range(nlev)]
        for l in range(nlev):                         # the same rough fields on both hierarchies (the coefficients stay the problem's)
            for vid in (T, U, E, R):
                data = seeded_field(la[l], 1000 * l + vid, scale=0.01)
                la[l].write_all(vid, data); lb[l].write_all(vid, data)
        # HPGMG_FUZZ_SEED / HPGMG_FUZZ_SEQUENCES: longer hunts with other seeds (the suite runs seed 0, 200 sequences)
        rng = np.random.default_rng(20260 + len(variant) + 7919 * int(os.environ.get("HPGMG_FUZZ_SEED", "0")))
        for n in range(12 if big else (int(os.environ.get("HPGMG_FUZZ_SEQUENCES", "200")) // (5 if mid else 1))):
            seq = random_sequence(rng, nlev)
            lib.hpgmg_set_lazy(1)
            va = run(lib, la, seq)
            lib.hpgmg_set_lazy(0)
            vb = run(lib, lb, seq)
            assert va == vb, (n, seq, va, vb)
            for l in range(nlev):
                g = {"box_dim": la[l].box_dim, "ghosts": la[l].ghosts, "jStride": la[l].jStride, "kStride": la[l].kStride}
                for vid in (T, U, E, R):
                    if big and l == 0 and n % 4:
                        continue                      # 8 boxes of 128^3: compare the finest level every fourth sequence only
                    a, b = interior_of(la[l].read_all(vid), g), interior_of(lb[l].read_all(vid), g)
                    assert np.array_equal(a, b), (n, l, vid, seq)
    finally:
        lib.hpgmg_set_lazy(1)
        sa.destroy(); sb.destroy()

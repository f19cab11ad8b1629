"""Per-operator pin of the oracle AND of the HIP plugin against the reference itself (SURVEY.md 8c "single-operator harness").

tests/golden/ops_golden.json holds, for every build variant of the reference and for the 16^3 problem cut into one box of 16^3 and into
2 x 2 x 2 boxes of 8^3, what the reference's operators leave after each call of a fixed script (oracle/op_harness.c linked with the
reference's own level.c / operators.<OP>.c / mg.c / solvers.c): sha256 over the whole padded vectors -- interior, ghost zones, row
padding -- and over the interiors, max-abs values, and the returned scalars to 17 digits.  tests/ops_script.py replays that script:
  * on the CPU restatement (`-m "not gpu"`): every vector must equal the reference's byte for byte, ghost cells and VECTOR_TEMP included;
  * on the HIP plugin (`-m gpu`): the same with the three-step exchange / boundary / stencil form (HPGMG_GHOST_FREE=0), and interiors +
    scalars with the product's ghost-free launches, which do not refresh an operand's ghost zones.
"""
import ctypes
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hpgmg_testlib import Backend, have_reference, load_golden  # noqa: E402
from ops_script import GEOMETRIES, HARNESS_VARIANTS, LARGE_CASES, replay, replay_cycle_forms  # noqa: E402

GOLD = load_golden("ops_golden.json")
CASES = [(v, bi, bd) for v in HARNESS_VARIANTS for bi, bd in GEOMETRIES] + LARGE_CASES      # LARGE_CASES: the sizes the bandwidth-bound kernels run at


def compare(gold, records, scalars, key):
    assert [r["name"] for r in records] == [g["name"] for g in gold["records"]]
    bad = [(r["name"], r["absmax"], g["absmax"]) for r, g in zip(records, gold["records"]) if r[key] != g[key]]
    assert not bad, "vectors that differ from the reference (name, max|ours|, max|reference|): %r" % bad[:6]
    assert scalars == gold["scalars"]


@pytest.mark.parametrize("variant,boxes_in_i,box_dim", CASES)
def test_oracle_equals_the_reference_operator_by_operator(variant, boxes_in_i, box_dim):
    gold = GOLD["%s %d %d" % (variant, boxes_in_i, box_dim)]
    geoms, records, scalars = replay(Backend.oracle(), variant, boxes_in_i, box_dim)
    assert {str(k): v for k, v in geoms.items()} == gold["geoms"]
    compare(gold, records, scalars, "sha_full")


def test_known_answers_of_the_survey():
    """7-pt VC Poisson, 16^3, U = 0, one smooth() then residual(): the two max-norms, identical for every box decomposition."""
    for key, u, r in (("7pt-cheby", "0.00017716835929184276", "0.12609106875183329"), ("7pt-gsrb", "0.00012607053412205999", "0.30621392397226965")):
        for geom in ("1 16", "2 8"):
            s = GOLD[key + " " + geom]["scalars"]
            assert s["first.norm_u"] == u and s["first.norm_res"] == r, (key, geom, s)


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (fixtures are regenerated from it)")
@pytest.mark.parametrize("variant", ["7pt-cheby", "fv4-gsrb", "27pt-gsrb"])
def test_fixtures_are_what_the_reference_produces(variant):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import subprocess
    from make_ops_golden import run_harness
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), os.path.join(root, "oracle", "_ref", "opharness-" + variant)], check=True)
    fresh = run_harness(variant, 2, 8)
    assert fresh == GOLD[variant + " 2 8"]


@pytest.mark.gpu
@pytest.mark.parametrize("ghost_free", [0, 1])
@pytest.mark.parametrize("variant,boxes_in_i,box_dim", CASES)
def test_hip_equals_the_reference_operator_by_operator(variant, boxes_in_i, box_dim, ghost_free):
    gold = GOLD["%s %d %d" % (variant, boxes_in_i, box_dim)]
    hip = Backend.hip()
    hip.lib.hpgmg_set_ghost_free.argtypes = [ctypes.c_int]
    hip.lib.hpgmg_set_ghost_free(ghost_free)
    try:
        geoms, records, scalars = replay(hip, variant, boxes_in_i, box_dim)
    finally:
        hip.lib.hpgmg_set_ghost_free(1)
    compare(gold, records, scalars, "sha_interior" if ghost_free else "sha_full")


@pytest.mark.gpu
@pytest.mark.parametrize("variant,boxes_in_i,box_dim", LARGE_CASES)
def test_hip_cycle_forms_equal_the_reference_operators(variant, boxes_in_i, box_dim):
    """What an F-cycle really launches on its bandwidth-bound levels -- smooth() in its in-cycle form (sweep pairs without the next-to-last iterate's store,
    red + black half sweeps in one pass), residual + norm and residual + restriction + zero_vector as single passes -- against the bytes the REFERENCE's separate
    operators leave on the same inputs (chebyshev.c:8-100 / gsrb.c:24-132, residual.c:9-51, restriction.c:104-212, misc.c:287-329)."""
    gold = GOLD["%s %d %d" % (variant, boxes_in_i, box_dim)]
    records, scalars, taken = replay_cycle_forms(Backend.hip(), variant, boxes_in_i, box_dim)
    assert taken["smooth_in_cycle"] == 1 and taken["residual_norm_fused"] == 1 and taken["residual_restrict_zero_fused"] == 1, taken
    want = {g["name"]: g for g in gold["records"]}
    bad = [(r["name"], r["absmax"], want[r["name"]]["absmax"]) for r in records if r["sha_interior"] != want[r["name"]]["sha_interior"]]
    assert len(records) == 2 and not bad, bad
    assert scalars["norm_r"] == gold["scalars"]["norm_r"], (scalars, gold["scalars"]["norm_r"])

"""HIP path vs the reference's golden numbers and vs the CPU oracle, whole F-cycles.

Everything goes through the C-ABI libraries (libhpgmg_fv.so -> libhpgmg_hip.so).  The bar is
bit-exactness of the printed 16-digit max-norms (tolerance stated: 0 ulp on those digits; the
north_star allows 1e-12 relative, which the last test also asserts explicitly).
"""
import pytest

from hpgmg_testlib import VARIANTS, load_golden, split_variant

pytestmark = pytest.mark.gpu
GOLD = load_golden("fcycle_norms.json")


def fmt(x):
    return "%1.15e" % x


SMALL = [("7pt-cheby", "4 8"), ("7pt-cheby", "5 8"), ("7pt-gsrb", "5 8"), ("7pt-cheby-helm", "5 8"), ("7ptcc-cheby", "5 8"),
         ("7pt-jacobi", "4 8"), ("27pt-cheby", "4 8"), ("27pt-cheby", "5 8"), ("27pt-gsrb", "4 8"), ("fv4-gsrb", "4 8"), ("fv4-gsrb", "5 8"), ("fv4-cheby", "4 8"), ("fv2-cheby", "4 8"), ("fv2-cheby", "5 8"), ("7pt-cheby", "4 1"), ("7pt-cheby", "4 27"), ("7pt-cheby-helm", "6 8"),
         ("7pt-cheby-periodic", "4 8"), ("7pt-cheby-periodic", "5 8"), ("7pt-cheby-helm-periodic", "4 8"), ("7pt-gsrb-periodic", "4 8")]


@pytest.mark.parametrize("variant,args", SMALL)
def test_hip_fcycle_matches_reference_golden(hip, variant, args):
    gold = GOLD[f"{variant} {args}"]
    base, bc = split_variant(variant)
    hip.configure(**VARIANTS[base])
    log2, per_rank = map(int, args.split())
    s = hip.solver_cli(log2, per_rank, bc=bc)
    try:
        assert [fmt(v) for v in s.three_sizes()] == gold["norms"]
        if gold["eigenvalue_max"]:
            assert ["%e" % s.level(l).eigenvalue for l in range(s.num_levels())] == gold["eigenvalue_max"]
        if gold.get("lambda_max"):
            assert ["%1.15e" % s.level(l).eigenvalue for l in range(s.num_levels())] == gold["lambda_max"]
        err, order = s.richardson()
        assert fmt(err) == gold["richardson_error"]
        assert "%0.3f" % order == gold["order"]
    finally:
        s.destroy()


@pytest.mark.parametrize("variant", ["7pt-cheby-helm", "7pt-cheby", "7pt-gsrb", "7ptcc-cheby", "27pt-cheby", "fv4-gsrb"])
def test_hip_fcycle_full_size_256(hip, variant):
    """BASELINE.json config 2 (`7 8`, 256^3, 8 boxes of 128^3) and its Poisson/GSRB/CC siblings."""
    gold = GOLD[f"{variant} 7 8"]
    hip.configure(**VARIANTS[variant])
    s = hip.solver_cli(7, 8)
    try:
        got = s.three_sizes()
        assert [fmt(v) for v in got] == gold["norms"]
        for g, r in zip(got, gold["norms"]):
            assert abs(g - float(r)) <= 1e-12 * float(r)        # the north_star's stated tolerance
        err, order = s.richardson()
        assert fmt(err) == gold["richardson_error"] and "%0.3f" % order == gold["order"]
    finally:
        s.destroy()


@pytest.mark.parametrize("variant", ["7pt-cheby-helm", "7pt-cheby", "7ptcc-cheby"])
def test_fp32_smoother_is_tolerance_gated_against_fp64(hip, variant):
    """BASELINE.json config 5 (`7 8`, mixed-precision Chebyshev smoother: fp32 coefficient streams, fp64 iterate and
    arithmetic).  Not bit-exact by construction; gated against the fp64 reference numbers:
    F-cycle residual norm within 2e-4 relative (measured 8e-5), discretisation error (the quantity the solver is for) within 1e-7 (measured 5e-9)
    relative, same convergence order.  Measured on MI355X: 8e-5 and 5e-9."""
    import ctypes
    gold = GOLD[f"{variant} 7 8"]
    hip.lib.hpgmg_set_smoother_precision.argtypes = [ctypes.c_int]
    hip.lib.hpgmg_set_smoother_precision(32)
    try:
        hip.configure(**VARIANTS[variant])
        s = hip.solver_cli(7, 8)
        got = s.three_sizes()
        ref = [float(r) for r in gold["norms"]]
        assert abs(got[0] - ref[0]) <= 2e-4 * ref[0], (got[0], ref[0])
        assert got[0] != ref[0] or variant == "7ptcc-cheby"     # the fp32 streams were really used (CC has only Dinv)
        assert abs(got[1] - ref[1]) <= 2e-4 * ref[1], (got[1], ref[1])   # 128^3: a sweep-pair level too (>= 2 M cells), fp32 streams there as well
        assert fmt(got[2]) == gold["norms"][2]                           # 64^3: no sweep-pair level, fp64 throughout
        err, order = s.richardson()
        assert abs(err - float(gold["richardson_error"])) <= 1e-7 * float(gold["richardson_error"])
        assert "%0.3f" % order == gold["order"]
        s.destroy()
    finally:
        hip.lib.hpgmg_set_smoother_precision(64)


def test_hipgraph_segments_replay_the_same_numbers(hip):
    """hpgmg_set_graphs(1): the <= 64^3 part of the cycle is captured on the second solve and replayed afterwards."""
    import ctypes
    import hpgmg_amd as H
    K = H.load_kernels()
    gold = GOLD["7pt-cheby-helm 6 8"]
    stats = (ctypes.c_longlong * 3)()
    hip.lib.hpgmg_set_graphs.argtypes = [ctypes.c_int]
    hip.lib.hpgmg_set_graphs(1)
    try:
        hip.configure(**VARIANTS["7pt-cheby-helm"])
        s = hip.solver_cli(6, 8)
        K.hpgmg_hip_graph_stats(stats); before = list(stats)
        for _ in range(4):
            assert fmt(s.fmg(0)) == gold["norms"][0]
        K.hpgmg_hip_graph_stats(stats)
        assert stats[1] > before[1] and stats[2] > before[2], (before, list(stats))     # captured, then replayed
        # the same hierarchy solved into ANOTHER vector: the cached graphs bake vector ids in, so the key must change with them
        import numpy as np
        import hpgmg_amd as Hh
        L = hip.lib
        L.hpgmg_solver_mg.restype = ctypes.c_void_p; L.hpgmg_solver_mg.argtypes = [ctypes.c_void_p]
        L.FMGSolve.restype = None
        L.FMGSolve.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        ab = (ctypes.c_double * 2)(); L.hpgmg_solver_coefficients.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]; L.hpgmg_solver_coefficients(s.ptr, ab)
        lv = s.level(0)
        want = lv.interior(Hh.VECTOR_U)
        for _ in range(3):                       # eager, captured, replayed -- with E in place of U
            L.zero_vector(lv.ptr, Hh.VECTOR_E)
            L.FMGSolve(L.hpgmg_solver_mg(s.ptr), 0, Hh.VECTOR_E, Hh.VECTOR_F, ab[0], ab[1], 1e-10)
            assert np.array_equal(lv.interior(Hh.VECTOR_E), want)
        assert fmt(s.fmg(0)) == gold["norms"][0]
        # solves started from different levels must not share segment keys (a key hash that ignored the start level once made a solve
        # from 2h replay the graphs captured for the solve from h)
        for l in range(3):
            s.restrict_rhs(l)
        for l in (1, 1, 2, 0, 0, 2, 1, 2):
            assert fmt(s.fmg(l)) == gold["norms"][l], l
        s.destroy()
    finally:
        hip.lib.hpgmg_set_graphs(0)


@pytest.mark.parametrize("mode", [2, 0])
@pytest.mark.parametrize("variant,args", [("fv4-gsrb", "4 8"), ("27pt-gsrb", "4 8"), ("27pt-cheby", "5 8"), ("fv2-cheby", "4 8"), ("fv4-cheby", "4 8")])
def test_small_levels_as_single_launches_give_the_same_norms(hip, variant, args, mode):
    """hpgmg_set_small_fused: smooth() / residual() of the non-7-point plugins on small levels as one single-workgroup launch each (exchange
    copies + boundary conditions + stencil per sweep inside).  Mode 2 (the default): smooth() on levels of ONE box, on an image of the box in LDS;
    mode 0: separate launches.  The same numbers in both."""
    import ctypes
    gold = GOLD[f"{variant} {args}"]
    hip.lib.hpgmg_set_small_fused.argtypes = [ctypes.c_int]
    hip.lib.hpgmg_set_small_fused(mode)
    try:
        hip.configure(**VARIANTS[variant])
        s = hip.solver_cli(*map(int, args.split()))
        assert [fmt(v) for v in s.three_sizes()] == gold["norms"]
        err, order = s.richardson()
        assert fmt(err) == gold["richardson_error"]
        s.destroy()
    finally:
        hip.lib.hpgmg_set_small_fused(2)


@pytest.mark.parametrize("variant,args", [("fv4-gsrb", "4 8"), ("27pt-gsrb", "4 8"), ("27pt-cheby", "5 8"), ("fv2-cheby", "4 8"), ("fv4-cheby", "4 8")])
def test_vcycle_tail_below_a_one_box_level_is_one_launch_and_gives_the_same_norms(hip, variant, args):
    """27-point / fv2 / fv4: the rest of a V-cycle below a level of ONE box (smooth, residual, restriction, zero_vector per level, the BiCGStab
    bottom solve, interpolation_vcycle and smooth per level upwards) runs as one single-workgroup launch (small_vtail_kernel; the default except for 27-point GSRB,
    hpgmg_set_small_vtail(1) = on for every plugin).  When on it must be taken (launch counter), and the norms are the golden ones with it and
    without it.  (Brick launches off: with them the one-box levels of the 27-point / fv4 plugins are bricks too and only the bottom solve is left of the tail --
    tests/test_gpu_operators.py::test_level_visits_as_one_launch_of_bricks_wide_stencils; this kernel remains the path of fv2 and of HPGMG_BRICK_WIDE=0.)"""
    import ctypes
    import hpgmg_amd as H
    gold = GOLD[f"{variant} {args}"]
    k = H.load_kernels()
    k.hpgmg_hip_small_vtail_launch_count.restype = ctypes.c_longlong
    hip.lib.hpgmg_set_small_vtail.argtypes = [ctypes.c_int]
    hip.lib.hpgmg_set_brick_wide.argtypes = [ctypes.c_int]
    hip.lib.hpgmg_set_brick_wide(0)
    try:
        for on in (1, 0):
            hip.lib.hpgmg_set_small_vtail(on)
            hip.configure(**VARIANTS[variant])
            before = k.hpgmg_hip_small_vtail_launch_count()
            s = hip.solver_cli(*map(int, args.split()))
            assert [fmt(v) for v in s.three_sizes()] == gold["norms"]
            err, order = s.richardson()
            assert fmt(err) == gold["richardson_error"]
            s.destroy()
            taken = k.hpgmg_hip_small_vtail_launch_count() - before
            assert (taken > 0) if on else (taken == 0), (on, taken)
    finally:
        hip.lib.hpgmg_set_small_vtail(2)
        hip.lib.hpgmg_set_brick_wide(1)


@pytest.mark.parametrize("variant,args", [("7pt-cheby", "4 8"), ("7pt-gsrb", "5 8"), ("fv4-gsrb", "4 8"), ("27pt-cheby", "5 8"), ("fv2-cheby", "4 8")])
def test_host_driven_bottom_solve_through_the_small_operator_queue(hip, variant, args):
    """The bottom solve driven from the host (hpgmg_set_fused_bottom(0): host/solvers.c BiCGStab calling mul_vectors, apply_op, dot, add_vectors,
    norm ... as the reference's solvers/bicgstab.c does through operators.h).  On a level of one small box the void operators wait for the dot product
    or norm that follows them and go out with it as ONE launch (small_ops_kernel): the queue must be used (group counter), and the norms are the golden
    ones with it and with every operator a launch of its own (hpgmg_set_small_ops(0))."""
    import ctypes
    gold = GOLD[f"{variant} {args}"]
    lib = hip.lib
    lib.hpgmg_set_fused_bottom.argtypes = [ctypes.c_int]
    lib.hpgmg_set_small_ops.argtypes = [ctypes.c_int]
    lib.hpgmg_small_ops_groups.restype = ctypes.c_longlong
    lib.hpgmg_small_ops_prefetched.restype = ctypes.c_longlong
    try:
        lib.hpgmg_set_fused_bottom(0)
        for on in (1, 0):
            lib.hpgmg_set_small_ops(on)
            hip.configure(**VARIANTS[variant])
            before, answered = lib.hpgmg_small_ops_groups(), lib.hpgmg_small_ops_prefetched()
            s = hip.solver_cli(*map(int, args.split()))
            assert [fmt(v) for v in s.three_sizes()] == gold["norms"]
            err, order = s.richardson()
            assert fmt(err) == gold["richardson_error"]
            s.destroy()
            groups = lib.hpgmg_small_ops_groups() - before
            assert (groups > 0) if on else (groups == 0), (on, groups)
            # dot(As, As) is followed by dot(As, s), norm(r) by dot(r, r0) with no operator in between: the second scalar of each pair comes from the
            # launch that answered the first
            answered = lib.hpgmg_small_ops_prefetched() - answered
            assert (answered > 0) if on else (answered == 0), (on, answered)
    finally:
        lib.hpgmg_set_fused_bottom(1)
        lib.hpgmg_set_small_ops(1)


def test_reference_three_launch_mode_gives_the_same_norms(hip):
    """HPGMG_GHOST_FREE=0 path (exchange_boundary + apply_BCs + stencil, as the reference sequences them)."""
    import ctypes
    gold = GOLD["7pt-gsrb 5 8"]
    hip.lib.hpgmg_set_ghost_free.argtypes = [ctypes.c_int]
    hip.lib.hpgmg_set_ghost_free(0)
    try:
        hip.configure(**VARIANTS["7pt-gsrb"])
        s = hip.solver_cli(5, 8)
        assert [fmt(v) for v in s.three_sizes()] == gold["norms"]
        s.destroy()
    finally:
        hip.lib.hpgmg_set_ghost_free(1)


def test_hip_equals_oracle_live(hip, oracle):
    """Same solve on both builds of the host layer, compared as doubles (no formatting)."""
    for be in (hip, oracle):
        be.configure(**VARIANTS["7pt-cheby-helm"])
    sh, so = hip.solver(2, 16), oracle.solver(2, 16)
    try:
        assert sh.three_sizes() == so.three_sizes()
        lh, lo = sh.level(0), so.level(0)
        import numpy as np
        for vid in (1, 2, 4, 5, 6, 7, 8, 9):   # U F R DINV BETA_* ALPHA
            assert np.array_equal(lh.interior(vid), lo.interior(vid)), vid
    finally:
        sh.destroy(); so.destroy()


@pytest.mark.parametrize("variant,args", [("7pt-cheby-helm", "8 8"), ("7pt-cheby", "8 8"), ("27pt-gsrb", "7 8"), ("27pt-gsrb", "7 64"), ("fv4-gsrb", "7 64")])
def test_hip_fcycle_at_the_stated_sizes_of_configs_3_and_4(hip, variant, args):
    """BASELINE.json config 4 (`8 8`: 512^3 in 8 boxes of 256^3) and config 3 (`7 64`: 512^3 in 64 boxes of 128^3, GSRB, 4th-order fv4
    and 27-point operators) as whole F-cycles at h, 2h, 4h + Richardson, against what the reference binary printed for exactly these
    arguments (tests/golden/make_golden.py; single-rank reading of the 8-GPU configurations: one MI355X holds the 512^3 problem)."""
    gold = GOLD[f"{variant} {args}"]
    hip.configure(**VARIANTS[variant])
    log2, per_rank = map(int, args.split())
    s = hip.solver_cli(log2, per_rank)
    try:
        assert [fmt(v) for v in s.three_sizes()] == gold["norms"]
        err, order = s.richardson()
        assert fmt(err) == gold["richardson_error"] and "%0.3f" % order == gold["order"]
    finally:
        s.destroy()


def test_timing_table_is_device_time(hip):
    """Per-level 'Total' rows filled from hipEvent pairs (HPGMG_TIMERS=device): they must account for the solve --
    the sum over levels within 10 % of the wall time of the timed solves (reference table: mg.c:54-161), every
    operator-class row no larger than its level's Total, and the numbers unchanged by the instrumentation."""
    import ctypes
    hip.configure(**VARIANTS["7pt-cheby-helm"])
    s = hip.solver_cli(7, 8)
    prev = hip.lib.hpgmg_get_timer_mode()
    try:
        hip.lib.hpgmg_set_timer_mode(1)
        solves = 8
        per_solve = s.bench(0, 3, solves)                 # MGResetTimers after the warm-up, like the reference protocol
        assert fmt(hip.lib.hpgmg_solver_fmg(s.ptr, 0)) == GOLD["7pt-cheby-helm 7 8"]["norms"][0]
        solves += 1
        rows = []
        for l in range(s.num_levels()):
            out = (ctypes.c_double * 9)()
            hip.lib.hpgmg_level_timers(s.level(l).ptr, out)
            rows.append(list(out))
        total = sum(r[8] for r in rows) / solves
        assert 0.90 * per_solve <= total <= 1.02 * per_solve, (total, per_solve, rows)
        for r in rows:
            assert all(v >= 0.0 for v in r)
            assert sum(r[:8]) <= 1.05 * r[8] + 1e-4, r
        assert rows[0][0] > 0.25 * rows[0][8]             # the fine level is dominated by the smoother
    finally:
        hip.lib.hpgmg_set_timer_mode(prev)
        s.destroy()

"""INTEGRATION.md Route B, executed: the REFERENCE's own driver (hpgmg-fv.c, mg.c, solvers.c, timers.c and level.c with the
three storage lines patched -- compiled from /root/reference by `make -C oracle ref` into oracle/_ref/routeb-*) linked
against the PRODUCT plugin (hpgmg_amd/csrc/host/operators_hip.c + libhpgmg_hip.so) must print the reference's golden
numbers on the MI355X.  This is the drop-in claim itself: nothing of this repository's driver is in that binary.
The binaries are built where /root/reference exists and travel to the GPU box with the snapshot."""
import os
import re
import subprocess

import pytest

from hpgmg_testlib import ROOT, load_golden

pytestmark = pytest.mark.gpu
GOLD = load_golden("fcycle_norms.json")


def pinned(out):
    keep = []
    for line in out.splitlines():
        m = re.search(r"(f-cycle\s+norm=\S+\s+rel=\S+|v-cycle=\s*\d+\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+)", line)
        if m:
            keep.append(m.group(1))
    return keep


@pytest.mark.parametrize("variant,args", [("7pt-cheby-helm", "5 8"), ("7pt-cheby-helm", "7 8"), ("7pt-gsrb", "5 8"), ("7pt-gsrb", "7 8"),
                                          ("7ptcc-cheby", "5 8"), ("7ptcc-cheby", "7 8"),
                                          # the reference's default build (4th-order operator, GSRB) and the other plugins
                                          ("fv4-gsrb", "5 8"), ("fv4-gsrb", "7 8"), ("27pt-gsrb", "5 8"), ("27pt-gsrb", "7 8"), ("fv2-cheby", "5 8")])
def test_reference_driver_runs_on_the_hip_plugin(variant, args):
    binary = os.path.join(ROOT, "oracle", "_ref", "routeb-" + variant)
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/routeb-* not built (needs /root/reference: make -C oracle ref)")
    gold = GOLD[f"{variant} {args}"]
    env = dict(os.environ, OMP_NUM_THREADS="4", HPGMG_LAZY_REPORT="1")
    out = subprocess.run([binary] + args.split(), capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # the reference's MGVCycle knows none of the fused hooks: the plugin's lazy queue must have recognised its legs and run them fused
    m = re.search(r"hpgmg lazy queue: (\d+) single-launch legs, (\d+) fused large-level units", out.stderr)
    if args == "7 8":       # the pre-smooth of a level too large for the single-launch legs: the queue sees residual(VECTOR_TEMP) next and runs the in-cycle form
        assert int(re.search(r"(\d+) smooths with VECTOR_TEMP proved dead", out.stderr).group(1)) > 0, out.stderr[-500:]
    if variant.startswith("7pt"):
        assert m and int(m.group(1)) > 0 and (int(m.group(2)) > 0 or args != "7 8"), out.stderr[-500:]      # large-level units only exist at 256^3
    elif variant != "27pt-gsrb":         # fv4 / fv2: the legs over their levels of one box (small_vtail_kernel); 27-point GSRB keeps its box kernel
        assert m and int(m.group(1)) > 0, out.stderr[-500:]
    lines = pinned(out.stdout)
    norms = []
    for l in lines:
        m = re.match(r"f-cycle\s+norm=(\S+)", l)
        if m and (not norms or norms[-1] != m.group(1)):
            norms.append(m.group(1))
    # timed section h, 2h, 4h (20 solves each) then the Richardson section h, 2h, 4h once more
    assert norms[:3] == gold["norms"] and norms[-3:] == gold["norms"], norms
    assert [l.split("<")[1] for l in lines if l.startswith("eigenvalue_max")] == gold["eigenvalue_max"]
    assert [l for l in lines if l.startswith("||error||")] == ["||error||=" + gold["richardson_error"]]
    assert [l for l in lines if l.startswith("order=")] == ["order=" + gold["order"]]
    assert "DOF/s=" in out.stdout


@pytest.mark.modes
@pytest.mark.parametrize("variant,args", [("7pt-cheby-vcycle", "5 8"), ("7pt-cheby-vcycle", "6 8"), ("fv4-gsrb-vcycle", "5 8")])
def test_reference_mgsolve_runs_on_the_hip_plugin(variant, args):
    """The reference's OTHER driver of the same plugin: built without -DUSE_FCYCLES its benchmark calls MGSolve (mg.c:1168-1233: V-cycles until
    the residual has dropped by 1e-10, a residual() + norm() after every cycle).  That is a second caller of the lazy operator queue with its own
    call order; every printed norm must equal what the reference binary itself prints for the same arguments (run here, on the host cores)."""
    routeb = os.path.join(ROOT, "oracle", "_ref", "routeb-" + variant)
    ref = os.path.join(ROOT, "oracle", "_ref", "hpgmg-" + variant)
    if not (os.path.exists(routeb) and os.path.exists(ref)):
        pytest.skip("oracle/_ref/*-vcycle not built (needs /root/reference: make -C oracle ref)")
    env = dict(os.environ, OMP_NUM_THREADS="8", HPGMG_LAZY_REPORT="1")
    outs = []
    for exe in (routeb, ref):
        out = subprocess.run([exe] + args.split(), capture_output=True, text=True, env=env, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out)
    keep = lambda o: re.findall(r"(v-cycle=\s*\d+\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+)", o.stdout)
    a, b = keep(outs[0]), keep(outs[1])
    assert len(a) > 50 and a == b, [x for x in zip(a, b) if x[0] != x[1]][:4]
    m = re.search(r"hpgmg lazy queue: (\d+) single-launch legs, (\d+) fused large-level units, (\d+) smooths with VECTOR_TEMP proved dead", outs[0].stderr)
    # (a problem of 64^3 cells is legs from top to bottom -- 7-point since round 5, fv4 since the wide bricks of round 6; brick launches above the tail --:
    #  no level is left to the unit-by-unit path)
    all_legs = args == "5 8"
    assert m and int(m.group(1)) > 0 and (all_legs or int(m.group(3)) > 0), outs[0].stderr[-500:]


@pytest.mark.modes
@pytest.mark.parametrize("variant,args", [("7pt-cheby-helm-mgpcg", "5 8"), ("7pt-cheby-helm-mgpcg", "6 8"), ("27pt-gsrb-mgpcg", "5 8")])
def test_reference_mgpcg_runs_on_the_hip_plugin(variant, args):
    """The reference's THIRD driver of the plugin, which its own main() never calls: MGPCG (mg.c:1500-1605), conjugate gradients preconditioned by
    one V-cycle per iteration.  oracle/mgpcg_harness.c is our main() around it; built once with the reference's operators (the reference itself) and
    once with the reference's mg.c / solvers.c on the product plugin.  It asks things of a plugin the F-cycle never does: every level grows by three
    vectors after MGBuild (create_vectors through the storage hooks), MGVCycle runs on vector ids beyond VECTORS_RESERVED, and dot() on the FINE
    level decides alpha and beta -- so every printed digit depends on the plugin summing in the reference's order (its single-thread order:
    OMP_NUM_THREADS=1 for the reference binary)."""
    routeb = os.path.join(ROOT, "oracle", "_ref", "routeb-" + variant)
    ref = os.path.join(ROOT, "oracle", "_ref", "hpgmg-" + variant)
    if not (os.path.exists(routeb) and os.path.exists(ref)):
        pytest.skip("oracle/_ref/*-mgpcg not built (needs /root/reference: make -C oracle ref)")
    outs = []
    for exe, threads in ((routeb, "8"), (ref, "1")):
        out = subprocess.run([exe] + args.split(), capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=threads, HPGMG_LAZY_REPORT="1"), timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out)
    keep = lambda o: re.findall(r"(iter=\s*\d+\s+norm=\S+\s+rel=\S+|MGPCG solve \d: norm\(u\)=\S+\s+Krylov iterations on the fine level so far=\d+|MGPCG dot\(u,f\)=\S+\s+mean\(u\)=\S+)", o.stdout)
    a, b = keep(outs[0]), keep(outs[1])
    assert len(a) >= 10 and a == b, [x for x in zip(a, b) if x[0] != x[1]][:4]
    m = re.search(r"hpgmg lazy queue: (\d+) single-launch legs, (\d+) fused large-level units, (\d+) smooths with VECTOR_TEMP proved dead", outs[0].stderr)
    assert m and int(m.group(1)) + int(m.group(2)) + int(m.group(3)) > 0, outs[0].stderr[-500:]      # the V-cycles inside still went through the operator queue


@pytest.mark.modes
@pytest.mark.parametrize("variant,args", [("7pt-cheby-cgbottom", "4 27"), ("7pt-cheby-cgbottom", "5 8"), ("27pt-gsrb-cgbottom", "4 27")])
def test_reference_cg_bottom_solver_runs_on_the_hip_plugin(variant, args):
    """The reference built with -DUSE_CG instead of -DUSE_BICGSTAB: its other host-driven bottom solver (solvers/cg.c: diagonally preconditioned
    CG, five extra vectors, a different sequence of residual / mul_vectors / dot / add_vectors / apply_op calls through the plugin's small-operator
    queue).  `4 27` has a 3^3-cell bottom level, so the solver really iterates.  Every pinned line must equal the reference binary's (one OpenMP
    thread there: the bottom level's dot products are sums of a few cells, but the fine-level ones of other paths are not)."""
    routeb = os.path.join(ROOT, "oracle", "_ref", "routeb-" + variant)
    ref = os.path.join(ROOT, "oracle", "_ref", "hpgmg-" + variant)
    if not (os.path.exists(routeb) and os.path.exists(ref)):
        pytest.skip("oracle/_ref/*-cgbottom not built (needs /root/reference: make -C oracle ref)")
    outs = []
    for exe, threads in ((routeb, "8"), (ref, "1")):
        out = subprocess.run([exe] + args.split(), capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=threads), timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out)
    keep = lambda o: re.findall(r"(f-cycle\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+|Bottom solver iterations\s+\d+)", o.stdout)
    a, b = keep(outs[0]), keep(outs[1])
    assert len(a) > 60 and a == b, [x for x in zip(a, b) if x[0] != x[1]][:4]


@pytest.mark.parametrize("variant,args", [("7pt-cheby-periodic", "5 8"), ("7pt-gsrb-periodic", "5 8"), ("7pt-cheby-periodic", "4 27")])
def test_reference_periodic_build_runs_on_the_hip_plugin(variant, args):
    """The reference built with -DUSE_PERIODIC_BC (its driver then removes the mean of f and of every iterate: mean() / shift_vector() inside the
    cycle, hpgmg-fv.c:296-302, mg.c:1176) on the plugin: the same pinned lines as the reference binary (one OpenMP thread there: mean() is a sum
    over the fine level)."""
    routeb = os.path.join(ROOT, "oracle", "_ref", "routeb-" + variant)
    ref = os.path.join(ROOT, "oracle", "_ref", "hpgmg-" + variant)
    if not (os.path.exists(routeb) and os.path.exists(ref)):
        pytest.skip("oracle/_ref/*-periodic not built (needs /root/reference: make -C oracle ref)")
    outs = []
    for exe, threads in ((routeb, "8"), (ref, "1")):
        out = subprocess.run([exe] + args.split(), capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=threads), timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out)
    a, b = pinned(outs[0].stdout), pinned(outs[1].stdout)
    assert len(a) > 60 and a == b, [x for x in zip(a, b) if x[0] != x[1]][:4]


_MODES = pytest.mark.modes      # SURVEY section 2 OUT OF SCOPE: only with -m "gpu and modes"


@pytest.mark.parametrize("variant,args", [("fv4-cheby", "5 8"), ("27pt-cheby", "5 8"),
                                          pytest.param("7pt-jacobi", "5 8", marks=_MODES), pytest.param("7pt-jacobi", "4 27", marks=_MODES),
                                          pytest.param("7pt-cheby-unlimit", "5 8", marks=_MODES),       # -DUNLIMIT_FMG_ITERATIONS: the F-cycle followed by V-cycles until converged (mg.c:1239-1247)
                                          pytest.param("7pt-cheby-ucycle", "5 8", marks=_MODES),       # -DUSE_UCYCLES: MGBuild does not agglomerate, the host-driven bottom solver runs on a level of eight boxes (mg.c:878-893)
                                          pytest.param("fv4-gsrb-ucycle", "5 8", marks=_MODES)])
def test_reference_other_smoothers_run_on_the_hip_plugin(variant, args):
    """The remaining smoother / operator pairs the reference can be built with (-DUSE_JACOBI on the 7-point operator, -DUSE_CHEBY on the 4th-order
    and the 27-point ones) through the reference's own driver on the plugin: the pinned lines of the reference binary."""
    routeb = os.path.join(ROOT, "oracle", "_ref", "routeb-" + variant)
    ref = os.path.join(ROOT, "oracle", "_ref", "hpgmg-" + variant)
    if not (os.path.exists(routeb) and os.path.exists(ref)):
        pytest.skip("oracle/_ref/routeb-" + variant + " not built (needs /root/reference: make -C oracle ref)")
    outs = []
    for exe, threads in ((routeb, "8"), (ref, "1")):
        out = subprocess.run([exe] + args.split(), capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=threads), timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out)
    a, b = pinned(outs[0].stdout), pinned(outs[1].stdout)
    assert len(a) > 60 and a == b, [x for x in zip(a, b) if x[0] != x[1]][:4]

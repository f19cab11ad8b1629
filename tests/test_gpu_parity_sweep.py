"""The parity sweep of tools/parity_sweep.sh as part of the GPU suite: the HIP executable and the CPU-oracle executable (same host layer, same
CLI as the reference's hpgmg-fv) run the same arguments, and every pinned line -- f-cycle norms at h / 2h / 4h (mg.c:1328), eigenvalue bounds of
rebuild_operator, Richardson error and order (mg.c:1128,1130) -- must be identical.  Every plugin and smoother, non-power-of-two decompositions
(27, 125, 216 boxes: 3^3 ... 6^3 bottoms where BiCGStab really iterates, agglomerating ladders) and the periodic builds.  The oracle side of
these odd decompositions is pinned to the reference binary by tests/test_oracle_vs_reference.py."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINNED = re.compile(r"f-cycle|\|\|error\|\||order=|eigenvalue")

CASES = [
    ("", "4 27"), ("--smoother gsrb", "4 125"), ("--helmholtz", "5 216"), ("--const-coeff", "4 27"), ("--smoother jacobi", "4 27"),
    ("--op 27pt", "4 27"), ("--op 27pt --smoother gsrb", "4 125"), ("--op 27pt --smoother gsrb", "6 27"),
    ("--op fv4 --smoother gsrb", "4 27"), ("--op fv4 --smoother gsrb", "4 125"), ("--op fv4 --smoother gsrb", "5 216"), ("--op fv4", "4 27"),
    ("--op fv2", "4 125"), ("--op fv2", "5 27"),
    ("--periodic", "4 27"), ("--periodic --smoother gsrb --helmholtz", "4 64"), ("--smoother gsrb", "4 343"),
]


def pinned_lines(exe, flags, size, **extra_env):
    cmd = [exe, "--warmup", "1", "--solves", "2"] + flags.split() + size.split()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_WAIT_POLICY="passive", **extra_env))
    assert out.returncode == 0, (cmd, out.stdout[-800:], out.stderr[-800:])
    lines = []
    for line in out.stdout.splitlines():
        if PINNED.search(line):
            line = re.sub(r"  done \(.*", "", line)
            line = re.sub(r".*(eigenvalue_max.*)", r"\1", line)
            lines.append(line.strip())
    return lines


@pytest.mark.parametrize("flags,size", CASES)
def test_hip_executable_prints_what_the_oracle_executable_prints(flags, size):
    hip = pinned_lines(os.path.join(ROOT, "hpgmg_amd", "bin", "hpgmg-fv"), flags, size)
    cpu = pinned_lines(os.path.join(ROOT, "oracle", "hpgmg-fv-oracle"), flags, size)
    assert len(hip) >= 10 and any("f-cycle" in l for l in hip)
    assert hip == cpu


_DEFAULT_SHAPE = {}


@pytest.mark.parametrize("flags,size,variant", [("--helmholtz", "7 8", "7pt-cheby-helm"), ("--smoother gsrb", "6 8", "7pt-gsrb"), ("", "6 8", "7pt-cheby"), ("--helmholtz", "8 8", "7pt-cheby-helm")])
@pytest.mark.parametrize("nw,kc", [(10, 0), (12, 0), (16, 0), (10, 16), (12, 32), (16, 8)])
def test_every_launch_shape_of_the_sweep_pair_kernel_prints_the_same_lines(flags, size, variant, nw, kc):
    """The sweep-pair kernel exists with 10, 12 and 16 waves per workgroup and marches k chunks of any length; a cost model picks per launch
    (pair.hip: smooth_pair).  Whatever it picks must not show in the numbers: each shape forced through HPGMG_TUNE_PAIR_NW / HPGMG_TUNE_PAIR_KC
    prints the lines of the default run, whose f-cycle norms are the reference's (tests/golden/fcycle_norms.json).  512^3, 256^3 and 128^3 fine levels:
    the levels of two million cells and more are the ones that take the kernel."""
    import json
    exe = os.path.join(ROOT, "hpgmg_amd", "bin", "hpgmg-fv")
    key = (flags, size)
    if key not in _DEFAULT_SHAPE:
        _DEFAULT_SHAPE[key] = pinned_lines(exe, flags, size)
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fcycle_norms.json")))[variant + " " + size]["norms"]
        printed = [re.search(r"norm=(\S+)", l).group(1) for l in _DEFAULT_SHAPE[key] if "f-cycle" in l]
        assert [g for g in gold if g in printed] == gold, (gold, printed[:6])
    forced = pinned_lines(exe, flags, size, HPGMG_TUNE_PAIR_NW=str(nw), HPGMG_TUNE_PAIR_KC=str(kc))
    assert len(forced) >= 10 and forced == _DEFAULT_SHAPE[key]


MGPCG_LINES = re.compile(r"(iter=\s*\d+\s+norm=\S+\s+rel=\S+|MGPCG solve \d: norm\(u\)=\S+\s+Krylov iterations on the fine level so far=\d+|MGPCG dot\(u,f\)=\S+\s+mean\(u\)=\S+)")


@pytest.mark.modes
@pytest.mark.parametrize("flags,size", [("--helmholtz", "5 8"), ("", "4 27"), ("--op 27pt --smoother gsrb", "5 8"), ("--op fv4 --smoother gsrb", "4 8"), ("--periodic", "4 8")])
def test_mgpcg_of_this_host_layer_on_the_gpu(flags, size):
    """MGPCG of this repository's host layer (host/mg.c, restating the reference's mg.c:1500-1605: conjugate gradients preconditioned with one
    V-cycle per iteration, `hpgmg-fv --mgpcg`) on the HIP plugin against the same on the CPU oracle with ONE OpenMP thread: the iterates hang on
    dot products over the fine level, so every digit is a statement about the order the plugin sums in.  (The oracle side is pinned to the reference's
    own MGPCG by tests/test_oracle_vs_reference.py.)"""
    outs = []
    for exe, threads in ((os.path.join(ROOT, "hpgmg_amd", "bin", "hpgmg-fv"), "8"), (os.path.join(ROOT, "oracle", "hpgmg-fv-oracle"), "1")):
        out = subprocess.run([exe, "--mgpcg"] + flags.split() + size.split(), capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS=threads))
        assert out.returncode == 0, (out.stdout[-800:], out.stderr[-800:])
        outs.append(MGPCG_LINES.findall(out.stdout))
    assert len(outs[0]) >= 8 and outs[0] == outs[1], [x for x in zip(outs[0], outs[1]) if x[0] != x[1]][:4]


@pytest.mark.modes
@pytest.mark.parametrize("flags,size", [("--helmholtz", "5 8"), ("", "4 27"), ("--op 27pt --smoother gsrb", "5 8"), ("--op fv4 --smoother gsrb", "5 8"), ("--op fv2", "4 8"), ("--periodic", "4 8"), ("--helmholtz", "7 8")])
def test_mgsolve_of_this_host_layer_on_the_gpu(flags, size):
    """`hpgmg-fv --vcycles`: the benchmark solving with MGSolve (V-cycles until the residual has dropped by 1e-10, a residual + norm after every cycle;
    the reference built without -DUSE_FCYCLES) on the HIP plugin against the CPU oracle: every v-cycle line of the three problem sizes and the
    Richardson estimate (the oracle side is pinned to the reference by tests/test_oracle_vs_reference.py)."""
    pat = re.compile(r"(v-cycle=\s*\d+\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+)")
    outs = []
    for exe in (os.path.join(ROOT, "hpgmg_amd", "bin", "hpgmg-fv"), os.path.join(ROOT, "oracle", "hpgmg-fv-oracle")):
        out = subprocess.run([exe, "--vcycles", "--warmup", "1", "--solves", "1"] + flags.split() + size.split(), capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_WAIT_POLICY="passive"))
        assert out.returncode == 0, (out.stdout[-800:], out.stderr[-800:])
        outs.append(pat.findall(out.stdout))
    assert len(outs[0]) >= 20 and outs[0] == outs[1], [x for x in zip(outs[0], outs[1]) if x[0] != x[1]][:4]


@pytest.mark.modes
@pytest.mark.parametrize("flags,size", [("--ucycles", "5 8"), ("--ucycles --op fv4 --smoother gsrb", "5 8"), ("--ucycles --op 27pt --smoother gsrb", "4 8"), ("--ucycles --helmholtz", "6 8"),
                                        ("--unlimit", "5 8"), ("--unlimit --op fv4 --smoother gsrb", "4 8"), ("--unlimit --helmholtz", "7 8")])
def test_other_cycle_shapes_of_this_host_layer_on_the_gpu(flags, size):
    """`--ucycles` (the reference's -DUSE_UCYCLES ladder: boxes halved, never merged, so the small levels and the bottom solve run on EIGHT boxes and none
    of the one-box forms applies) and `--unlimit` (V-cycles after the F-cycle until converged) on the HIP plugin against the CPU oracle, every pinned line."""
    pat = re.compile(r"(f-cycle\s+norm=\S+\s+rel=\S+|v-cycle=\s*\d+\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+)")
    outs = []
    for exe in (os.path.join(ROOT, "hpgmg_amd", "bin", "hpgmg-fv"), os.path.join(ROOT, "oracle", "hpgmg-fv-oracle")):
        env = dict(os.environ, OMP_WAIT_POLICY="passive")
        if "--ucycles" in flags:
            env["OMP_NUM_THREADS"] = "1"        # the bottom solver's dot products run over eight boxes: the plugin sums them in the reference's one-thread order
        out = subprocess.run([exe, "--warmup", "1", "--solves", "1"] + flags.split() + size.split(), capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, (out.stdout[-800:], out.stderr[-800:])
        outs.append(pat.findall(out.stdout))
    assert len(outs[0]) >= 10 and outs[0] == outs[1], [x for x in zip(outs[0], outs[1]) if x[0] != x[1]][:4]


@pytest.mark.modes
@pytest.mark.parametrize("flags,size", [("", "4 27"), ("--op 27pt --smoother gsrb", "4 27"), ("--op fv4 --smoother gsrb", "4 27"), ("--helmholtz", "5 8"), ("--op fv2", "4 125")])
def test_cg_bottom_solver_of_this_host_layer_on_the_gpu(flags, size):
    """`--bottom-solver cg`: the reference's other host-driven bottom solver (-DUSE_CG, solvers/cg.c) in this repository's host layer, on the HIP plugin
    against the CPU oracle -- the device bottom solve (BiCGStab) steps aside and the solver's operator calls go through the small-operator queue."""
    pat = re.compile(r"(f-cycle\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+|Bottom solver iterations\s+\d+)")
    outs = []
    for exe in (os.path.join(ROOT, "hpgmg_amd", "bin", "hpgmg-fv"), os.path.join(ROOT, "oracle", "hpgmg-fv-oracle")):
        out = subprocess.run([exe, "--bottom-solver", "cg", "--warmup", "1", "--solves", "2"] + flags.split() + size.split(), capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_WAIT_POLICY="passive"))
        assert out.returncode == 0, (out.stdout[-800:], out.stderr[-800:])
        outs.append(pat.findall(out.stdout))
    assert len(outs[0]) >= 10 and any(l.startswith("Bottom") for l in outs[0]) and outs[0] == outs[1], [x for x in zip(outs[0], outs[1]) if x[0] != x[1]][:4]


@pytest.mark.parametrize("which,flags,size", [("mgsolve", "--helmholtz", "5 8"), ("mgsolve", "--op fv4 --smoother gsrb", "5 8"), ("mgsolve", "--op 27pt --smoother gsrb", "5 8"),
                                              ("mgpcg", "--helmholtz", "5 8"), ("shapes", "--ucycles", "5 8"), ("shapes", "--unlimit", "5 8"), ("cg", "--helmholtz", "5 8")])
def test_one_case_of_every_fenced_off_mode_runs_with_the_hot_path_suite(which, flags, size):
    """The solver modes above are out of scope and only run with -m "gpu and modes" -- but they share MGVCycle's legs with the hot path: brick
    launches, the tails, the sweep pairs, the ordered sums.  One case of each therefore stays in the default `-m gpu` run (a few seconds together), so that a change
    to those legs which breaks a mode fails the round, not a log nobody reads."""
    {"mgsolve": test_mgsolve_of_this_host_layer_on_the_gpu, "mgpcg": test_mgpcg_of_this_host_layer_on_the_gpu,
     "shapes": test_other_cycle_shapes_of_this_host_layer_on_the_gpu, "cg": test_cg_bottom_solver_of_this_host_layer_on_the_gpu}[which](flags, size)

"""Pins the oracle against the REFERENCE ITSELF (only where /root/reference exists).

oracle/Makefile builds (a) the unmodified reference and (b) the reference's own driver
(level.c, mg.c, solvers.c, hpgmg-fv.c) linked against oracle/operators_cpu.c in place of
operators.7pt.c.  Both must print identical pinned lines; and our level_type must be
layout-identical to the reference's (otherwise (b) would not even run).
"""
import os
import re
import subprocess

import pytest

from hpgmg_testlib import ROOT, have_reference

pytestmark = pytest.mark.skipif(not have_reference(), reason="/root/reference not present on this machine")
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.fixture(scope="module")
def ref_build():
    # the Route-B binaries of the `ref` target link the product's kernel library: make sure it exists first
    subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "hpgmg_amd", "csrc")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, stdout=subprocess.DEVNULL)
    return REF


def cli_env(threads="4"):
    """Environment of a reference / oracle CLI run: a few threads that SPIN between the thousands of tiny parallel regions of the coarse
    levels (the suite-wide OMP_WAIT_POLICY=passive of conftest.py, meant for the in-process oracle next to a GPU, makes the reference
    binary ten to fifty times slower)."""
    env = dict(os.environ, OMP_NUM_THREADS=threads)
    env.pop("OMP_WAIT_POLICY", None)
    return env


def pinned_lines(binary, args, threads="4"):
    env = cli_env(threads)
    out = subprocess.run([binary] + args.split(), capture_output=True, text=True, env=env, check=True).stdout
    keep = []
    for line in out.splitlines():
        m = re.search(r"(f-cycle\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+|lambda_max\.\.\. <\S+|attempting to create.*)", line)
        if m:
            keep.append(m.group(1))
    return keep


def test_level_type_layout_matches_reference(ref_build):
    a = subprocess.run([os.path.join(ref_build, "layout_ref")], capture_output=True, text=True, check=True).stdout
    b = subprocess.run([os.path.join(ref_build, "layout_ours")], capture_output=True, text=True, check=True).stdout
    assert a == b and "sizeof(level_type)" in a


@pytest.mark.parametrize("variant", ["7pt-cheby", "7pt-gsrb", "7pt-cheby-helm", "7ptcc-cheby", "7pt-jacobi", "27pt-cheby", "27pt-gsrb", "fv4-gsrb", "fv4-cheby", "fv2-cheby"])
@pytest.mark.parametrize("args", ["4 8", "5 8"])
def test_reference_driver_with_our_operators_prints_reference_numbers(ref_build, variant, args):
    ref = pinned_lines(os.path.join(ref_build, "hpgmg-" + variant), args)
    hyb = pinned_lines(os.path.join(ref_build, "hybrid-" + variant), args)
    assert len(ref) > 60
    assert ref == hyb


@pytest.mark.parametrize("variant,args", [("7pt-cheby-helm-mgpcg", "4 8"), ("27pt-gsrb-mgpcg", "4 8"), ("7pt-cheby-cgbottom", "4 27"), ("7pt-cheby-periodic", "4 8"),
                                          ("7pt-cheby-vcycle", "4 8"), ("7pt-cheby-unlimit", "4 8"), ("7pt-cheby-ucycle", "4 8"), ("fv4-gsrb-ucycle", "4 8")])
def test_the_reference_s_other_drivers_with_our_operators(ref_build, variant, args):
    """The restatement under the callers the F-cycle build never reaches: MGPCG (oracle/mgpcg_harness.c around mg.c:1500: fine-level dot
    products decide every digit), the CG bottom solver (-DUSE_CG), the periodic build (mean / shift_vector in the cycle), MGSolve (no
    -DUSE_FCYCLES), -DUNLIMIT_FMG_ITERATIONS and -DUSE_UCYCLES (no agglomeration: the bottom solver works on a level of eight boxes).  One OpenMP thread on both sides: the reference's sums move with its thread count."""
    def lines(binary):
        out = subprocess.run([binary] + args.split(), capture_output=True, text=True, env=cli_env("1"), check=True).stdout
        return re.findall(r"(f-cycle\s+norm=\S+\s+rel=\S+|v-cycle=\s*\d+\s+norm=\S+\s+rel=\S+|iter=\s*\d+\s+norm=\S+\s+rel=\S+|MGPCG solve \d: \S+.*|MGPCG dot.*|"
                          r"\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+|Bottom solver iterations\s+\d+)", out)
    ref, hyb = lines(os.path.join(ref_build, "hpgmg-" + variant)), lines(os.path.join(ref_build, "hybrid-" + variant))
    assert len(ref) >= 10 and ref == hyb, [x for x in zip(ref, hyb) if x[0] != x[1]][:4]


@pytest.mark.parametrize("variant,flags,args", [("7pt-cheby-helm-mgpcg", ["--helmholtz"], "4 8"), ("27pt-gsrb-mgpcg", ["--op", "27pt", "--smoother", "gsrb"], "4 8")])
def test_our_mgpcg_prints_what_the_reference_s_mgpcg_prints(ref_build, variant, flags, args):
    """MGPCG of OUR host layer (host/mg.c, `hpgmg-fv-oracle --mgpcg`) against the reference's MGPCG (mg.c:1500-1605 under oracle/mgpcg_harness.c),
    one OpenMP thread each: the same iterates to the last digit, the same iteration counts."""
    pat = r"(iter=\s*\d+\s+norm=\S+\s+rel=\S+|MGPCG solve \d: norm\(u\)=\S+\s+Krylov iterations on the fine level so far=\d+|MGPCG dot\(u,f\)=\S+\s+mean\(u\)=\S+)"
    ref = re.findall(pat, subprocess.run([os.path.join(ref_build, "hpgmg-" + variant)] + args.split(), capture_output=True, text=True, env=cli_env("1"), check=True).stdout)
    ours = re.findall(pat, subprocess.run([os.path.join(ROOT, "oracle", "hpgmg-fv-oracle"), "--mgpcg"] + flags + args.split(), capture_output=True, text=True, env=cli_env("1"), check=True).stdout)
    assert len(ref) >= 10 and ref == ours, [x for x in zip(ref, ours) if x[0] != x[1]][:4]


@pytest.mark.parametrize("variant,flags,args", [("7pt-cheby-ucycle", ["--ucycles"], "4 8"), ("fv4-gsrb-ucycle", ["--ucycles", "--op", "fv4", "--smoother", "gsrb"], "4 8"),
                                                ("7pt-cheby-unlimit", ["--unlimit"], "4 8"), ("7pt-cheby-ucycle", ["--ucycles"], "5 8")])
def test_our_host_layer_with_the_reference_s_other_cycle_flags(ref_build, variant, flags, args):
    """`hpgmg-fv-oracle --ucycles` (MGBuild without agglomeration: the reference's -DUSE_UCYCLES ladder, mg.c:878-893) and `--unlimit` (V-cycles after the
    F-cycle until converged: -DUNLIMIT_FMG_ITERATIONS, mg.c:1239-1247) against the reference binaries built with those flags: the level table, every
    f-cycle / v-cycle line, the Richardson estimate."""
    pat = r"(attempting to create a \S+ level from .* boxes|f-cycle\s+norm=\S+\s+rel=\S+|v-cycle=\s*\d+\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+)"
    # one OpenMP thread: with U-cycles the bottom solver's dot products run over eight boxes, and the reference's sum then moves with its thread count
    ref = re.findall(pat, subprocess.run([os.path.join(ref_build, "hpgmg-" + variant)] + args.split(), capture_output=True, text=True, env=cli_env("1"), check=True).stdout)
    ours = re.findall(pat, subprocess.run([os.path.join(ROOT, "oracle", "hpgmg-fv-oracle")] + flags + args.split(), capture_output=True, text=True, env=cli_env("1"), check=True).stdout)
    assert len(ref) > 60 and ref == ours, [x for x in zip(ref, ours) if x[0] != x[1]][:4]


@pytest.mark.parametrize("variant,flags,args", [("7pt-cheby-cgbottom", [], "4 27"), ("27pt-gsrb-cgbottom", ["--op", "27pt", "--smoother", "gsrb"], "4 27"), ("7pt-cheby-cgbottom", [], "4 8")])
def test_our_cg_bottom_solver_against_the_reference_s(ref_build, variant, flags, args):
    """`hpgmg-fv-oracle --bottom-solver cg` (host/solvers.c: the reference's -DUSE_CG choice, solvers/cg.c) against the reference built with -DUSE_CG:
    the pinned lines and the bottom solver's iteration counts of the timing tables (`4 27`: a 3^3-cell bottom level, the solver iterates)."""
    pat = r"(f-cycle\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+|eigenvalue_max<\S+|Bottom solver iterations\s+\d+)"
    ref = re.findall(pat, subprocess.run([os.path.join(ref_build, "hpgmg-" + variant)] + args.split(), capture_output=True, text=True, env=cli_env("1"), check=True).stdout)
    ours = re.findall(pat, subprocess.run([os.path.join(ROOT, "oracle", "hpgmg-fv-oracle"), "--bottom-solver", "cg"] + flags + args.split(), capture_output=True, text=True, env=cli_env("1"), check=True).stdout)
    assert len(ref) > 60 and any(l.startswith("Bottom") for l in ref) and ref == ours, [x for x in zip(ref, ours) if x[0] != x[1]][:4]


@pytest.mark.parametrize("variant,flags,args", [("7pt-cheby-vcycle", [], "4 8"), ("fv4-gsrb-vcycle", ["--op", "fv4", "--smoother", "gsrb"], "4 8")])
def test_our_mgsolve_prints_what_the_reference_s_mgsolve_prints(ref_build, variant, flags, args):
    """`hpgmg-fv-oracle --vcycles` (this repository's MGSolve, host/mg.c) against the reference built without -DUSE_FCYCLES (mg.c:1168-1233): every
    v-cycle line of the three problem sizes, the Richardson estimate."""
    pat = r"(v-cycle=\s*\d+\s+norm=\S+\s+rel=\S+|\|\|error\|\|=\S+|order=\S+)"
    ref = re.findall(pat, subprocess.run([os.path.join(ref_build, "hpgmg-" + variant)] + args.split(), capture_output=True, text=True, env=cli_env(), check=True).stdout)
    ours = re.findall(pat, subprocess.run([os.path.join(ROOT, "oracle", "hpgmg-fv-oracle"), "--vcycles"] + flags + args.split(), capture_output=True, text=True, env=cli_env(), check=True).stdout)
    assert len(ref) > 50 and ref == ours, [x for x in zip(ref, ours) if x[0] != x[1]][:4]


def _masked(text):
    """stdout with every timing figure (and the one line naming threads / backend) replaced: what a log parser keys on stays"""
    import re
    out = []
    for line in text.splitlines():
        if "MPI Tasks of" in line:
            line = "N MPI Tasks of M threads"
        line = re.sub(r"done \([0-9.]+ seconds\)", "done (T seconds)", line)
        line = re.sub(r"\(([0-9.]+) seconds\)", "(T seconds)", line)
        line = re.sub(r"time=\s*[0-9.]+\s+DOF/s=\s*[0-9.e+]+", "time=T DOF/s=R", line)
        if re.match(r"^(smooth|  max|  min|residual|applyOp|BLAS1|Boundary Conditions|Restriction|  local restriction|  pack MPI buffers|  unpack MPI buffers|"
                    r"  MPI_Isend|  MPI_Irecv|  MPI_Waitall|Interpolation|  local interpolation|Ghost Zone Exchange|  local exchange|MPI_collectives|"
                    r"Total by level|   Total time in MGBuild|   Total time in MGSolve|      number of v-cycles|Bottom solver iterations|"
                    r"            Performance|calculating D\^\{-1\}|\s+rebuilding operator)", line):
            line = re.sub(r"[0-9]+\.[0-9]+(e[+-][0-9]+)?", "T", line)
        out.append(line.rstrip())
    return out


@pytest.mark.parametrize("variant,flags,args", [("7pt-cheby", [], "4 8"), ("7pt-cheby-helm", ["--helmholtz"], "4 8"), ("7pt-gsrb", ["--smoother", "gsrb"], "5 8")])
def test_cli_stdout_has_the_reference_layout(ref_build, variant, flags, args):
    """our driver prints the reference's report line for line -- level creation, operator rebuild,
    the 10+10 f-cycle lines per size, the timing table, DOF/s, Richardson error -- so HPGMG log parsers work unchanged.
    Only timing figures (masked here) and the thread/backend banner differ; every pinned number is identical."""
    env = cli_env()
    ref = subprocess.run([os.path.join(ref_build, "hpgmg-" + variant)] + args.split(), capture_output=True, text=True, env=env, check=True).stdout
    ours = subprocess.run([os.path.join(ROOT, "oracle", "hpgmg-fv-oracle")] + flags + args.split(), capture_output=True, text=True, env=env, check=True).stdout
    a, b = _masked(ref), _masked(ours)
    assert len(a) == len(b), (len(a), len(b))
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
    assert not diff, diff[:5]


@pytest.mark.parametrize("variant,flags,args", [
    ("7pt-cheby", [], "4 125"), ("7pt-cheby", [], "4 343"), ("7pt-gsrb", ["--smoother", "gsrb"], "4 216"),
    ("7pt-cheby-helm", ["--helmholtz"], "5 27"), ("fv4-gsrb", ["--op", "fv4", "--smoother", "gsrb"], "4 125"), ("27pt-cheby", ["--op", "27pt"], "4 125"),
    ("7pt-cheby", [], "4 2"), ("7pt-cheby", [], "4 7"), ("7pt-cheby", [], "5 1"),
    # 7^3 boxes with the Helmholtz operator and 6^3 with the 27-point one: non-power-of-two decompositions of these two plugins in place of the
    # costlier `7pt-cheby-helm 4 729` / `27pt-cheby 4 343` (log2_box_dim cannot go below 4)
    ("7pt-cheby-helm", ["--helmholtz"], "4 343"), ("27pt-cheby", ["--op", "27pt"], "4 216"),
])
def test_our_host_layer_and_oracle_against_the_reference_binary_on_odd_decompositions(ref_build, variant, flags, args):
    """Beyond the committed golden cases: box counts that are not powers of two (3^3 ... 9^3 boxes; requested counts that are
    not cubes are rounded down like hpgmg-fv.c:181-197 does), where the coarsening ladder agglomerates differently and the
    bottom solve really iterates.  Our level/MG construction + CPU operators must print what the reference prints."""
    # one thread for the reference: with a 5^3 ... 9^3 bottom grid BiCGStab's dot products matter, and the reference's OpenMP
    # reduction(+) gives run-to-run differences in the last digits with several threads (ours is the 1-thread order by design)
    ref = pinned_lines(os.path.join(ref_build, "hpgmg-" + variant), args, threads="1")
    ours = pinned_lines(os.path.join(ROOT, "oracle", "hpgmg-fv-oracle"), " ".join(flags + args.split()))
    assert len(ref) > 10 and ours == ref

#!/usr/bin/env python3
"""Generate tests/golden/fcycle_norms.json from the REFERENCE itself.

Runs the unmodified reference binaries that oracle/Makefile builds from
/root/reference into oracle/_ref/ (only possible where /root/reference exists)
and records, per build variant and argument pair, every line the reference pins:
the f-cycle residual norms at h, 2h, 4h (mg.c:1328), the Richardson error and
order (mg.c:1128,1130) and the eigenvalue bounds printed by rebuild_operator.
The JSON is data (numbers printed by the reference), committed so the oracle and
the HIP path can be checked on machines without /root/reference.
"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref")
CASES = {
    # `8 8` (512^3 in 8 boxes of 256^3) = BASELINE config 4 at its stated size; `7 64` (512^3 in 64 boxes of 128^3) = config 3's
    "7pt-cheby": ["4 8", "5 8", "6 8", "7 8", "4 1", "5 1", "4 27", "8 8"],
    "7pt-gsrb": ["4 8", "5 8", "6 8", "7 8"],
    "7pt-cheby-helm": ["4 8", "5 8", "6 8", "7 8", "8 8"],
    "7ptcc-cheby": ["4 8", "5 8", "6 8", "7 8"],
    "7pt-jacobi": ["4 8", "5 8"],
    "27pt-cheby": ["4 8", "5 8", "6 8", "7 8"],
    "27pt-gsrb": ["4 8", "5 8", "7 8", "7 64"],
    "fv4-gsrb": ["4 8", "5 8", "6 8", "7 8", "7 64"],
    "fv4-cheby": ["4 8", "5 8"],
    "fv2-cheby": ["4 8", "5 8"],
    "7pt-cheby-periodic": ["4 8", "5 8"],          # -DUSE_PERIODIC_BC: Poisson, the mean is removed (mg.c, solvers.c)
    "7pt-cheby-helm-periodic": ["4 8"],
    "7pt-gsrb-periodic": ["4 8"],
}
def run(variant, args):
    # periodic Poisson subtracts the mean: an order-dependent sum, which the reference only reproduces with one thread
    # (its OpenMP reduction gives 1.572586767996712e-03 / ...719e-03 from run to run with 8 threads); pin the 1-thread order
    env = dict(os.environ, OMP_NUM_THREADS="1" if "periodic" in variant else "8")
    out = subprocess.run([os.path.join(REF, "hpgmg-" + variant)] + args.split(), capture_output=True, text=True, env=env, check=True).stdout
    fc = re.findall(r"f-cycle\s+norm=(\S+)\s+rel=(\S+)", out)
    # the Richardson section solves h, 2h, 4h once each: the last three f-cycle lines
    last3 = fc[-3:]
    # timed section: 20 identical solves per size; make sure they agree with the Richardson ones
    sizes = []
    for n, r in fc:
        if not sizes or sizes[-1] != (n, r): sizes.append((n, r))
    assert [s[0] for s in sizes[:3]] == [s[0] for s in last3], (variant, args, sizes[:4], last3)
    err = re.search(r"\|\|error\|\|=(\S+)", out).group(1)
    order = re.search(r"order=(\S+)", out).group(1)
    eig = re.findall(r"eigenvalue_max<(\S+)", out)
    lam = re.findall(r"lambda_max\.\.\. <(\S+)", out)          # black-box rebuild (27pt, fv2, fv4 with Chebyshev)
    levels = re.findall(r"attempting to create a (\d+)\^3 level from (\d+) x (\d+)\^3 boxes", out)
    return {"norms": [n for n, _ in last3], "rels": [r for _, r in last3], "richardson_error": err, "order": order,
            "eigenvalue_max": eig, "lambda_max": lam, "levels": [[int(a), int(b), int(c)] for a, b, c in levels]}
def main():
    """no arguments: regenerate everything; `--only SUBSTR`: (re)run the cases whose key contains SUBSTR and merge them into the file
    (`--missing`: of those, only the ones the file does not hold yet)"""
    path = os.path.join(ROOT, "tests", "golden", "fcycle_norms.json")
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    gold = {"_generated_by": "tests/golden/make_golden.py from oracle/_ref/hpgmg-* (reference built by oracle/Makefile: gcc -O2 -fopenmp, no MPI)"}
    if only:
        gold = json.load(open(path))
    for v, arglist in CASES.items():
        for a in arglist:
            key = f"{v} {a}"
            if only and only not in key or (only and key in gold and "--missing" in sys.argv):
                continue
            print("running", key, file=sys.stderr)
            gold[key] = run(v, a)
    with open(os.path.join(ROOT, "tests", "golden", "fcycle_norms.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
if __name__ == "__main__":
    main()

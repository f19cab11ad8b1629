#!/usr/bin/env python3
"""Generate tests/golden/ops_golden.json from the REFERENCE's operator layer.

oracle/Makefile links the unmodified reference sources (timers.c level.c operators.<OP>.c mg.c solvers.c) with oracle/op_harness.c -- our
main(), which calls the operators of operators.h one at a time on the 16^3 test problem and dumps every vector and scalar they leave --
into oracle/_ref/opharness-<variant>.  This script runs those executables (only possible where /root/reference exists), condenses each
dumped vector to sha256 digests (whole padded boxes, and interior cells only) plus its max-abs, keeps the scalars to 17 digits, and writes
the JSON the tests compare the CPU restatement and the HIP plugin with (tests/ops_script.py replays the same calls).  The fixtures are
data produced by the reference, not its text.  One thread: dot() and mean() are sums whose OpenMP reduction order is not reproducible.
"""
import json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ops_script import GEOMETRIES, HARNESS_VARIANTS, LARGE_CASES, parse_harness_file  # noqa: E402


def run_harness(variant, boxes_in_i, box_dim, keep_data=False):
    exe = os.path.join(ROOT, "oracle", "_ref", "opharness-" + variant)
    with tempfile.TemporaryDirectory(dir=os.environ.get("HPGMG_HARNESS_TMP")) as tmp:      # a dump of the 256^3 problem is 4.4 GB
        path = os.path.join(tmp, "dump.bin")
        subprocess.run([exe, str(boxes_in_i), str(box_dim), path], check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS="1"))
        config, geoms, records, scalars = parse_harness_file(path, keep_data)
    return {"config": config, "geoms": {str(k): v for k, v in geoms.items()}, "records": records, "scalars": scalars}


def main():
    out = {}
    for variant in HARNESS_VARIANTS:
        for bi, bd in GEOMETRIES:
            out["%s %d %d" % (variant, bi, bd)] = run_harness(variant, bi, bd)
            print(variant, bi, bd, len(out["%s %d %d" % (variant, bi, bd)]["records"]), "vectors")
    for variant, bi, bd in LARGE_CASES:      # the sizes the bandwidth-bound kernels run at (tests/ops_script.py)
        out["%s %d %d" % (variant, bi, bd)] = run_harness(variant, bi, bd)
        print(variant, bi, bd, len(out["%s %d %d" % (variant, bi, bd)]["records"]), "vectors")
    with open(os.path.join(HERE, "ops_golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()

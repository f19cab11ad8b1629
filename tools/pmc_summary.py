#!/usr/bin/env python3
"""Combine two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide prescribes)
into profiles/<tag>_pmc_summary.json: HBM-side bytes per launch of the fine-level kernels next to the bytes a launch
has to move in the form it has (a kernel that does two sweeps in one pass: its own stream count, once -- 80 B per cell for
the Chebyshev sweep pair, 56 for the fv4 and 32 for the 27-point red + black pass).

usage: tools/pmc_summary.py <fetch.db> <write.db> <out.json> [--cells N]
rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> x2; calibrated here on scale_vector
(16 B/cell algorithmic): corrected total / algorithmic must come out ~1.0 (reported as "calibration").
"""
import json, sqlite3, sys

def per_kernel(path):
    db = sqlite3.connect(path); cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
    rows = cur.execute(f"select d.event_id, k.kernel_name, d.grid_size_x*d.grid_size_y*d.grid_size_z from rocpd_kernel_dispatch{suf} d "
                       f"join rocpd_info_kernel_symbol{suf} k on d.kernel_id=k.id").fetchall()
    val = {}
    for ev, v in cur.execute(f"select event_id, value from rocpd_pmc_event{suf}"):
        val[ev] = val.get(ev, 0.0) + v
    out = {}
    for ev, name, grid in rows:
        out.setdefault(name, []).append((grid, val.get(ev, 0.0)))
    return out

def biggest(kern, pattern):
    """average counter value over the launches of the largest grid of the kernels matching pattern.  Several instantiations may share that grid (the sweep-pair
    kernel runs the 256^3 and the 128^3 level of config 2 as 256 workgroups each; the pre-pass has a once-per-rebuild packing form): of those launched at least
    a quarter as often as the most frequent one, the one that moves the most bytes = the fine level's."""
    groups = []
    for name, lst in kern.items():
        if pattern in name:
            g = max(x[0] for x in lst)
            vals = [v for gg, v in lst if gg == g]
            groups.append((g, sum(vals) / len(vals), len(vals), name))
    if not groups: return None
    gmax = max(x[0] for x in groups)
    groups = [x for x in groups if x[0] == gmax]
    often = max(x[2] for x in groups)
    groups = [x for x in groups if 4 * x[2] >= often]
    return max(groups, key=lambda x: x[1])

def main():
    fetch, write, out = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    cells = int(sys.argv[sys.argv.index("--cells") + 1]) if "--cells" in sys.argv else 256 ** 3
    rows = {"cheby_pair_fine": ("cheby_pair_kernel", 80), "cheby_pair_edge_columns": ("cheby_pair_edge_kernel", 0), "cheby_fine": ("stencil7_wide_kernelILi0ELi0", 72), "residual_fine": ("stencil7_wide_kernelILi0ELi3", 56),
            "residual_restrict_zero_fine": ("stencil7_wide_kernelILi0ELi6", 58), "residual_norm_fine": ("stencil7_wide_kernelILi0ELi7", 56),
            "norm_copy_restrict_fine": ("norm_copy_restrict_kernel", 17), "fv4_gsrb_fine": ("fv4_tile_kernelILi5ELi1", 56), "fv4_rb_fine": ("fv4_rb_kernel", 56), "fv4_special_cells": ("fv4_special_kernel", 0), "stencil27_gsrb_fine": ("stencil27_tile_kernelILi1", 32), "stencil27_rb_fine": ("stencil27_rb_kernel", 32),
            "scale_fine": ("elementwise_kernelILi3", 16), "interp_p0_fine": ("interp_blocks_kernelILi0", 17),
            "interp_p1_fine": ("interp_blocks_kernelILi1", 17), "restrict_fine": ("restrict_blocks_kernelILi0", 9), "norm_fine": ("absmax_kernel", 8)}
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), MI355X",
           "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950: 64 B counted per 128-B request); WRITE_SIZE KiB x 1024", "cells": cells, "kernels": {}}
    for key, (pat, bpc) in rows.items():
        f, w = biggest(fetch, pat), biggest(write, pat)
        if not f or not w: continue
        fb, wb = f[1] * 1024 * 2, w[1] * 1024
        alg = bpc * cells if bpc else fb + wb
        res["kernels"][key] = {"kernel": f[3], "launches_averaged": f[2], "FETCH_SIZE_bytes_raw": f[1] * 1024, "FETCH_bytes_corrected_x2": fb,
                               "WRITE_SIZE_bytes": wb, "hbm_bytes_per_launch": fb + wb, "algorithmic_bytes": alg, "ratio": (fb + wb) / alg}
    if "scale_fine" in res["kernels"]:
        if 0.5 < res["kernels"]["scale_fine"]["ratio"] < 2.0: res["calibration_scale_vector_ratio"] = res["kernels"]["scale_fine"]["ratio"]
        else: del res["kernels"]["scale_fine"]      # this workload never runs scale_vector on the fine level: the row would describe a small level
    k = res["kernels"]
    if "cheby_pair_fine" in k:      # the smoother launch bench.py times = edge-column pre-pass + pair kernel (two sweeps)
        res["hbm_bytes_per_launch_cheby_fine"] = k["cheby_pair_fine"]["hbm_bytes_per_launch"] + k.get("cheby_pair_edge_columns", {}).get("hbm_bytes_per_launch", 0.0)
        res["sweeps_per_launch"] = 2
    elif "fv4_rb_fine" in k:            # one launch = both coloured half sweeps in one pass (the pre-pass launches are small and listed separately)
        res["hbm_bytes_per_launch_smoother_fine"] = k["fv4_rb_fine"]["hbm_bytes_per_launch"]; res["sweeps_per_launch"] = 2
    elif "fv4_gsrb_fine" in k:
        res["hbm_bytes_per_launch_smoother_fine"] = k["fv4_gsrb_fine"]["hbm_bytes_per_launch"]; res["sweeps_per_launch"] = 1
    elif "stencil27_rb_fine" in k:      # one launch = both coloured half sweeps
        res["hbm_bytes_per_launch_smoother_fine"] = k["stencil27_rb_fine"]["hbm_bytes_per_launch"]; res["sweeps_per_launch"] = 2
    elif "stencil27_gsrb_fine" in k:
        res["hbm_bytes_per_launch_smoother_fine"] = k["stencil27_gsrb_fine"]["hbm_bytes_per_launch"]; res["sweeps_per_launch"] = 1
    elif "cheby_fine" in k:
        res["hbm_bytes_per_launch_cheby_fine"] = k["cheby_fine"]["hbm_bytes_per_launch"]
        res["sweeps_per_launch"] = 1
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:16s} hbm {v['hbm_bytes_per_launch']/1e6:9.1f} MB  algorithmic {v['algorithmic_bytes']/1e6:9.1f} MB  ratio {v['ratio']:.3f}")

if __name__ == "__main__":
    main()

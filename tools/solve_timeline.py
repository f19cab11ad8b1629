#!/usr/bin/env python3
"""The kernel sequence of the LAST timed solve of a profiled bench.py run, from a rocprofv3 rocpd database.

usage: tools/solve_timeline.py <results.db> [--out FILE] [--marker SUBSTRING]
A solve starts at the last dispatch whose kernel name contains the marker (default: norm_copy_restrict_kernel, the opening pass of FMGSolve,
mg.c:1262-1270) -- or one dispatch earlier when that one is the zero_vector(U) of the bench step -- and ends with the last dispatch of the run.
Per dispatch: start (us since the solve began), duration, idle gap since the previous dispatch ended, grid, short kernel name; then totals per
kernel and the split busy / idle.  This is how the launch structure of a cycle is read (launch-bound levels).
"""
import collections, re, sqlite3, sys


def short(name):
    m = re.match(r"_ZN5hpgmg\d+([A-Za-z0-9_]+?)(I.*)?E", name)
    base = m.group(1) if m else name.split("(")[0]
    targs = re.findall(r"L[ib](\d+)", name.split("Ev")[0]) if m else []
    return base + ("<" + ",".join(targs) + ">" if targs else "")


def main():
    db_path = sys.argv[1]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    marker = sys.argv[sys.argv.index("--marker") + 1] if "--marker" in sys.argv else "norm_copy_restrict_kernel"
    db = sqlite3.connect(db_path); cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    suffix = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
    rows = cur.execute(f"select k.kernel_name, d.start, d.end, d.grid_size_x * d.grid_size_y * d.grid_size_z, d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z "
                       f"from rocpd_kernel_dispatch{suffix} d join rocpd_info_kernel_symbol{suffix} k on d.kernel_id=k.id order by d.start").fetchall()
    starts = [n for n, r in enumerate(rows) if marker in r[0]]
    if not starts:
        raise SystemExit("no dispatch matches the marker " + marker)
    first = starts[-1]
    if first > 0 and "fill_kernel" in rows[first - 1][0]:
        first -= 1
    seq = rows[first:]
    t0, prev_end = seq[0][1], seq[0][1]
    lines, per = [], collections.defaultdict(lambda: [0, 0.0])
    busy = idle = 0.0
    for name, s, e, grid, wg in seq:
        gap = max(0.0, (s - prev_end) / 1e3)
        dur = (e - s) / 1e3
        key = short(name)
        lines.append(f"{(s - t0) / 1e3:10.1f} {dur:9.2f} {gap:7.2f} {grid:10d} {wg:5d}  {key}")
        per[key][0] += 1; per[key][1] += dur
        busy += dur; idle += gap
        prev_end = max(prev_end, e)
    text = [f"last solve of {db_path}: {len(seq)} dispatches, {(prev_end - t0) / 1e3:.1f} us from first start to last end, {busy:.1f} us in kernels, {idle:.1f} us idle between them", "",
            f"{'start us':>10} {'dur us':>9} {'gap us':>7} {'grid':>10} {'wg':>5}  kernel"] + lines + ["", f"{'calls':>6} {'total us':>10}  kernel"]
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        text.append(f"{n:6d} {t:10.1f}  {k}")
    print("\n".join(text[:3] + text[-(len(per) + 2):]))
    if out:
        open(out, "w").write("\n".join(text) + "\n")


if __name__ == "__main__":
    main()

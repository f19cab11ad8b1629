#!/usr/bin/env python3
"""usage: tools/kernel_durations.py <results.db> <substring of the kernel name> [--grids]
every dispatch of the matching kernels at their largest grid, in launch order (us); --grids: calls / mean duration per grid size instead"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
suffix = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
rows = cur.execute(f"select k.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from rocpd_kernel_dispatch{suffix} d join rocpd_info_kernel_symbol{suffix} k on d.kernel_id=k.id order by d.start").fetchall()
sel = [(n, s, e, gx * max(gy, 1)) for n, s, e, gx, gy in rows if sys.argv[2] in n]
for name in sorted(set(n for n, _, _, _ in sel)):
    v = [(e - s, g) for n, s, e, g in sel if n == name]
    if "--grids" in sys.argv:
        by = collections.defaultdict(list)
        for d, g in v: by[g].append(d)
        print(name[:90])
        for g in sorted(by, reverse=True): print(f"   grid {g:10d}  calls {len(by[g]):5d}  mean {sum(by[g]) / len(by[g]) / 1e3:8.1f} us  total {sum(by[g]) / 1e6:8.3f} ms")
    else:
        big = max(g for _, g in v)
        print(name[:90], "grid", big, [round(d / 1e3, 1) for d, g in v if g == big][:24])

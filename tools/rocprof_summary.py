#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace and/or PMC counters) into JSON + text.

usage: tools/rocprof_summary.py <results.db> [--out profiles/NAME]
Per kernel: calls, total/avg/min/max duration; for the largest launches of each kernel also the
grid.  With PMC data: per kernel average counter values (FETCH_SIZE/WRITE_SIZE are in KiB units
in rocprofv3; gfx950 correction for wide streaming reads is applied by the caller).
"""
import collections, json, sqlite3, sys

def main():
    db_path = sys.argv[1]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    db = sqlite3.connect(db_path); cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    suffix = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
    rows = cur.execute(f"select d.id, k.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.workgroup_size_y "
                       f"from rocpd_kernel_dispatch{suffix} d join rocpd_info_kernel_symbol{suffix} k on d.kernel_id=k.id").fetchall()
    agg = collections.defaultdict(list)
    for did, name, s, e, g, wx, wy in rows:
        agg[name].append((e - s, g, did))
    total = sum(d for v in agg.values() for d, _, _ in v)
    summary = {"db": db_path, "dispatches": len(rows), "total_kernel_ms": total / 1e6, "kernels": []}
    for name, v in sorted(agg.items(), key=lambda kv: -sum(d for d, _, _ in kv[1])):
        durs = sorted(d for d, _, _ in v)
        big_grid = max(g for _, g, _ in v)
        big = [d for d, g, _ in v if g == big_grid]
        summary["kernels"].append({"name": name, "calls": len(v), "total_ms": sum(durs) / 1e6, "avg_us": sum(durs) / len(durs) / 1e3,
                                   "min_us": durs[0] / 1e3, "max_us": durs[-1] / 1e3, "pct": 100.0 * sum(durs) / total,
                                   "largest_grid": big_grid, "largest_grid_calls": len(big), "largest_grid_avg_us": sum(big) / len(big) / 1e3})
    # PMC counters, if collected
    pmc_tab = f"rocpd_pmc_event{suffix}"
    if pmc_tab in tables:
        try:
            info = dict(cur.execute(f"select id, name from rocpd_info_pmc{suffix}").fetchall())
            ev2disp = dict(cur.execute(f"select event_id, id from rocpd_kernel_dispatch{suffix}").fetchall())
            disp = {did: (name, g) for did, name, s, e, g, wx, wy in rows}
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for event_id, pmc_id, value in cur.execute(f"select event_id, pmc_id, value from {pmc_tab}"):
                did = ev2disp.get(event_id)
                if did is None: continue
                name, g = disp[did]
                acc[(name, g)][info.get(pmc_id, str(pmc_id))].append(value)
            summary["pmc"] = [{"name": n, "grid": g, "counters": {c: {"avg": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}}
                              for (n, g), cs in sorted(acc.items(), key=lambda kv: -kv[0][1])][:40]
        except Exception as exc:
            summary["pmc_error"] = repr(exc)
    text = [f"rocprofv3 summary of {db_path}: {len(rows)} dispatches, {total/1e6:.3f} ms of kernel time", ""]
    text.append(f"{'total ms':>10} {'calls':>7} {'avg us':>9} {'max us':>9} {'%':>6}  {'big-grid avg us':>15}  kernel")
    for k in summary["kernels"]:
        text.append(f"{k['total_ms']:10.3f} {k['calls']:7d} {k['avg_us']:9.2f} {k['max_us']:9.2f} {k['pct']:6.1f}  {k['largest_grid_avg_us']:15.2f}  {k['name'][:110]}")
    for p in summary.get("pmc", [])[:12]:
        text.append(f"PMC {p['name'][:70]} grid={p['grid']}: " + ", ".join(f"{c}={v['avg']:.1f} (n={v['n']})" for c, v in p["counters"].items()))
    print("\n".join(text))
    if out:
        json.dump(summary, open(out + ".json", "w"), indent=1)
        open(out + ".txt", "w").write("\n".join(text) + "\n")

if __name__ == "__main__":
    main()

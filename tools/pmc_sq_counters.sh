cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; rm -rf gpurun_out/prof_lds
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d gpurun_out/prof_lds/p -o pmc -- python3 bench.py --workload config3-fv4 --no-cpu-baseline --no-also --steps 2 --warmup 1 > /dev/null 2>gpurun_out/lds.err </dev/null
db=$(find gpurun_out/prof_lds -name '*.db' | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
suf = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
info = dict(cur.execute(f"select id, name from rocpd_info_pmc{suf}").fetchall())
rows = cur.execute(f"select d.event_id, k.kernel_name, d.grid_size_x*d.grid_size_y*d.grid_size_z from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} k on d.kernel_id=k.id").fetchall()
ev = {e: (n, g) for e, n, g in rows}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for e, pid, v in cur.execute(f"select event_id, pmc_id, value from rocpd_pmc_event{suf}"):
    if e in ev: acc[ev[e]][info.get(pid, str(pid))].append(v)
for (n, g), cs in sorted(acc.items(), key=lambda kv: -kv[0][1]):
    if 'fv4_rb_kernel' in n or 'fv4_tile_kernelILi5ELi3ELi8' in n or 'stencil27_rb_kernel' in n:
        print(n[:60], g, {c: round(sum(v)/len(v)) for c, v in cs.items()})
PY
rm -rf gpurun_out/prof_lds

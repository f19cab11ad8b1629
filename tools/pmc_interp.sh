cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; rm -rf gpurun_out/prof_ip
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVES -d gpurun_out/prof_ip/p -o pmc -- python3 bench.py --workload ${1:-config3-27pt} --no-cpu-baseline --no-also --steps 2 --warmup 1 > /dev/null 2>gpurun_out/ip.err </dev/null
db=$(find gpurun_out/prof_ip -name '*.db' | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
suf = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
info = dict(cur.execute(f"select id, name from rocpd_info_pmc{suf}").fetchall())
rows = cur.execute(f"select d.event_id, k.kernel_name, d.grid_size_x*d.grid_size_y*d.grid_size_z, d.end-d.start from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} k on d.kernel_id=k.id").fetchall()
ev = {e: (n, g) for e, n, g, t in rows}
dur = collections.defaultdict(list)
for e, n, g, t in rows: dur[(n, g)].append(t / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for e, pid, v in cur.execute(f"select event_id, pmc_id, value from rocpd_pmc_event{suf}"):
    if e in ev: acc[ev[e]][info.get(pid, str(pid))].append(v)
for (n, g), cs in sorted(acc.items(), key=lambda kv: -kv[0][1]):
    if 'interp_tensor' in n or 'norm_copy' in n:
        print(n[8:60], g, 'us', round(max(dur[(n, g)])), {c: round(sum(v)/len(v)) for c, v in cs.items()})
PY
rm -rf gpurun_out/prof_ip

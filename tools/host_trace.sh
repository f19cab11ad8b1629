#!/bin/bash
# usage (GPU box, repo root): bash tools/host_trace.sh <tag> [workload] -- HIP API trace + kernel trace of one bench run: how long the HOST spends per launch
tag=${1:-h}; wl=${2:-config3-fv4}
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/prof_${tag}
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats -d gpurun_out/prof_${tag}/ht -o bench -- python3 bench.py --workload $wl --no-cpu-baseline --no-also --steps 3 --warmup 2 > gpurun_out/${tag}_bench_traced.json 2>gpurun_out/${tag}_ht.err </dev/null
db=$(find gpurun_out/prof_${tag}/ht -name '*.db' | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
print([t for t in tables if 'region' in t or 'api' in t.lower() or 'string' in t][:12])
suffix = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
try:
    rows = cur.execute(f"select s.string, r.start, r.end from rocpd_region{suffix} r join rocpd_string{suffix} s on r.name_id = s.id order by r.start").fetchall()
except Exception as e:
    print("region query failed:", e); rows = []
agg = collections.defaultdict(list)
for n, s, e in rows: agg[n].append(e - s)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort(); print(f"{n:40s} calls {len(v):7d}  total {sum(v)/1e6:9.2f} ms  median {v[len(v)//2]/1e3:7.2f} us  p90 {v[int(len(v)*0.9)]/1e3:7.2f} us")
# launch cadence on the host during the last solve: start-to-start of consecutive hipLaunchKernel-like calls
L = [(s, e) for n, s, e in rows if 'LaunchKernel' in n or 'hipModuleLaunch' in n]
if L:
    tail = L[-640:]
    d = sorted(b[0] - a[0] for a, b in zip(tail, tail[1:]))
    print("host launch cadence over the last", len(tail), "launches: median", d[len(d)//2]/1e3, "us, p25", d[len(d)//4]/1e3, "p75", d[3*len(d)//4]/1e3)
PY
find gpurun_out/prof_${tag} -name '*.db' -delete

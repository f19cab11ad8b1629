#!/bin/bash
# usage (GPU box, repo root): bash tools/quick_timeline.sh <tag> [workload] -- kernel stats + the dispatch sequence of the last solve of one profiled bench run
tag=${1:-q}; wl=${2:-config2}
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/prof_${tag}
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}/kt -o bench -- python3 bench.py --workload $wl --no-cpu-baseline --no-also > gpurun_out/${tag}_bench_profiled.json 2>gpurun_out/${tag}_kt.err </dev/null
db=$(find gpurun_out/prof_${tag}/kt -name '*.db' | head -1)
python3 tools/rocprof_summary.py "$db" --out gpurun_out/${tag}_kernel_stats </dev/null | head -${3:-45} | cut -c1-200
python3 tools/solve_timeline.py "$db" --out gpurun_out/${tag}_last_solve.txt </dev/null | cut -c1-160
find gpurun_out/prof_${tag} -name '*.db' -delete

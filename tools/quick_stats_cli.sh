#!/bin/bash
# usage (GPU box, repo root): bash tools/quick_stats_cli.sh <tag> <hpgmg-fv arguments...> -- kernel stats of this repository's driver (all solves of the run: 3 sizes x (warm-up + timed))
tag=${1:-qc}; shift
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/prof_${tag}
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}/kt -o cli -- hpgmg_amd/bin/hpgmg-fv "$@" > gpurun_out/${tag}_cli.out 2>gpurun_out/${tag}_kt.err </dev/null
db=$(find gpurun_out/prof_${tag}/kt -name '*.db' | head -1)
python3 tools/rocprof_summary.py "$db" --out gpurun_out/${tag}_kernel_stats </dev/null | head -40 | cut -c1-200
grep -E "DOF/s" gpurun_out/${tag}_cli.out | head -3
find gpurun_out/prof_${tag} -name '*.db' -delete

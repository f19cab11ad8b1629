#!/bin/bash
# usage (GPU box, repo root): bash tools/quick_stats_routeb.sh <tag> [binary suffix, default 7pt-cheby-helm] [log2 boxes] -- kernel stats of the reference's driver on the plugin
tag=${1:-qb}; bin=${2:-7pt-cheby-helm}; shift 2; args=${@:-7 8}
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/prof_${tag}
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}/kt -o rb -- oracle/_ref/routeb-$bin $args > gpurun_out/${tag}_routeb.out 2>gpurun_out/${tag}_kt.err </dev/null
db=$(find gpurun_out/prof_${tag}/kt -name '*.db' | head -1)
python3 tools/rocprof_summary.py "$db" --out gpurun_out/${tag}_kernel_stats </dev/null | head -45 | cut -c1-200
grep -E "DOF/s|done \(" gpurun_out/${tag}_routeb.out | tail -3
find gpurun_out/prof_${tag} -name '*.db' -delete

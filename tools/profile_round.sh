#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# bench.py once plainly, once under rocprofv3 --kernel-trace --stats, then the two PMC passes; summaries land in
# gpurun_out/<tag>_* for copying into profiles/.
# optional 2nd argument: bench.py --workload (default config2); 3rd: fine-grid cells for the PMC summary (default 256^3)
tag=${1:-rXX}; wl=${2:-config2}; cells=${3:-16777216}
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python3 bench.py --no-also --workload $wl > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err </dev/null
tail -1 gpurun_out/${tag}_bench.json
rm -rf gpurun_out/prof_${tag}
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}/kt -o bench -- python3 bench.py --workload $wl --no-cpu-baseline --no-also > gpurun_out/${tag}_bench_profiled.json 2>gpurun_out/${tag}_kt.err </dev/null
db=$(find gpurun_out/prof_${tag}/kt -name '*.db' | head -1)
python3 tools/rocprof_summary.py "$db" --out gpurun_out/${tag}_bench_kernel_stats </dev/null | head -12
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${tag}/fetch -o pmc -- python3 bench.py --workload $wl --no-cpu-baseline --no-also --steps 3 --warmup 2 > /dev/null 2>gpurun_out/${tag}_fetch.err </dev/null
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${tag}/write -o pmc -- python3 bench.py --workload $wl --no-cpu-baseline --no-also --steps 3 --warmup 2 > /dev/null 2>gpurun_out/${tag}_write.err </dev/null
f=$(find gpurun_out/prof_${tag}/fetch -name '*.db' | head -1); w=$(find gpurun_out/prof_${tag}/write -name '*.db' | head -1)
python3 tools/rocprof_summary.py "$f" --out gpurun_out/${tag}_pmc_fetch </dev/null > /dev/null
python3 tools/rocprof_summary.py "$w" --out gpurun_out/${tag}_pmc_write </dev/null > /dev/null
python3 tools/pmc_summary.py "$f" "$w" gpurun_out/${tag}_pmc_summary.json --cells $cells </dev/null
# the plain bench line once more, now that this build's counter summary exists: roofline.traffic / traffic_source on the line are then this tag's
# (the copy under profiles/ lives on this box only; the caller copies gpurun_out/${tag}_* into profiles/ of the repository)
cp gpurun_out/${tag}_pmc_summary.json profiles/ 2>/dev/null
timeout 900 python3 bench.py --no-also --workload $wl > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err </dev/null
tail -1 gpurun_out/${tag}_bench.json | cut -c1-200
find gpurun_out/prof_${tag} -name '*.db' -size +30M -delete

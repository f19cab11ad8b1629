#!/bin/bash
# profile sets of a round (TAG=r03e bash tools/profile_all.sh): config 2 (headline), config 3 fv4 and 27-point with counters, the other configurations' bench lines, Route B
cd ${GRAFT_REPO_ROOT:-.}
bash tools/profile_round.sh ${TAG:-rXX} config2 16777216 > gpurun_out/${TAG:-rXX}_log.txt 2>&1; tail -12 gpurun_out/${TAG:-rXX}_log.txt
bash tools/profile_round.sh ${TAG:-rXX}_fv4 config3-fv4 134217728 > gpurun_out/${TAG:-rXX}_fv4_log.txt 2>&1; tail -8 gpurun_out/${TAG:-rXX}_fv4_log.txt
bash tools/profile_round.sh ${TAG:-rXX}_27pt config3-27pt 134217728 > gpurun_out/${TAG:-rXX}_27pt_log.txt 2>&1; tail -6 gpurun_out/${TAG:-rXX}_27pt_log.txt
for w in config1 config4 config5; do timeout 600 python3 bench.py --workload $w --no-cpu-baseline --no-also 2>/dev/null | tail -1 > gpurun_out/${TAG:-rXX}_bench_$w.json; python3 -c "import json; d=json.load(open('gpurun_out/${TAG:-rXX}_bench_$w.json')); print('$w', round(d['ms_per_step'],3), d['value'])"; done
timeout 600 python3 bench.py --route-b 2>/dev/null | tail -1 > gpurun_out/${TAG:-rXX}_bench_routeb.json; cat gpurun_out/${TAG:-rXX}_bench_routeb.json | cut -c1-200

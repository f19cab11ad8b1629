#!/bin/bash
# the three bench lines again (TAG=r03e), once profiles/ holds this build's counter summaries (roofline.traffic / traffic_source)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 900 python3 bench.py --no-also > gpurun_out/${TAG:-rXX}_bench.json 2> gpurun_out/${TAG:-rXX}_bench.err
timeout 900 python3 bench.py --no-also --workload config3-fv4 > gpurun_out/${TAG:-rXX}_fv4_bench.json 2> /dev/null
timeout 900 python3 bench.py --no-also --workload config3-27pt > gpurun_out/${TAG:-rXX}_27pt_bench.json 2> /dev/null
tail -c 600 gpurun_out/${TAG:-rXX}_bench.json

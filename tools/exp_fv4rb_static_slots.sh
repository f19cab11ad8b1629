#!/bin/bash
# usage (from the repo root, here AND on the GPU box): bash tools/exp_fv4rb_static_slots.sh build | run | restore
#   build   : compiles kernels/fv4_rb.hip with -DHPGMG_EXP_STATIC_SLOTS (every ring slot of the march a compile-time constant: the RESULTS ARE VOID, the timing is
#             an upper bound of what compile-time ring slots could give) and relinks libhpgmg_hip.so in place
#   run     : bench.py --workload config3-fv4 on that library (parity_ok will be False by construction)
#   restore : rebuilds the product library
# The measurement of round 6: profiles/r06f_fv4_rb_phase_static.txt.
set -e
cd "$(dirname "$0")/.."
case "$1" in
  build)   make -C hpgmg_amd/csrc FLAGS_fv4_rb="-Xarch_device -mllvm=-misched=gcn-iterative-minreg -DHPGMG_EXP_STATIC_SLOTS" -W kernels/fv4_rb.hip ;;
  run)     python3 bench.py --workload config3-fv4 --no-also --no-cpu-baseline --steps 4 --warmup 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['parity_ok'])" ;;
  restore) touch hpgmg_amd/csrc/kernels/fv4_rb.hip; make -C hpgmg_amd/csrc ;;
  *) echo "usage: $0 build|run|restore"; exit 2 ;;
esac

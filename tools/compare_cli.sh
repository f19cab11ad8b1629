#!/bin/bash
# tools/compare_cli.sh -- run the HIP executable and the CPU oracle executable with the same
# arguments and diff every pinned line (f-cycle norms, eigenvalue bounds, Richardson error/order).
# usage: tools/compare_cli.sh [--helmholtz] [--smoother gsrb] ... log2_box_dim boxes_per_rank
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ext(){ grep -E "f-cycle|v-cycle=|iter=|MGPCG solve|MGPCG dot|Bottom solver iterations|\|\|error\|\||order=|eigenvalue" | sed -E 's/  done \(.*//; s/.*(eigenvalue_max.*)/\1/' | uniq -c; }
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-8} OMP_WAIT_POLICY=passive
timeout 120 "$ROOT/hpgmg_amd/bin/hpgmg-fv" --warmup 1 --solves 2 "$@" > /tmp/hip_full.txt 2>&1 || { echo "HIP run failed"; tail -5 /tmp/hip_full.txt; exit 1; }
timeout 300 "$ROOT/oracle/hpgmg-fv-oracle" --warmup 1 --solves 2 "$@" > /tmp/cpu_full.txt 2>&1 || { echo "oracle run failed"; tail -5 /tmp/cpu_full.txt; exit 1; }
ext < /tmp/hip_full.txt > /tmp/hip.txt; ext < /tmp/cpu_full.txt > /tmp/cpu.txt
if diff /tmp/hip.txt /tmp/cpu.txt > /tmp/cli.diff; then echo "PARITY OK  [$*]  ($(wc -l < /tmp/hip.txt) pinned lines identical)"; else echo "PARITY MISMATCH [$*]"; head -20 /tmp/cli.diff; fi
grep -E "DOF/s" /tmp/hip_full.txt | sed 's/^/   hip: /'

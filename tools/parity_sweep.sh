#!/bin/bash
# tools/parity_sweep.sh -- run on the GPU box: HIP executable vs CPU oracle executable over every plugin / smoother and a
# spread of decompositions (1 box ... 11^3 boxes, power-of-two and odd box counts), diffing every pinned line
# (f-cycle norms at h/2h/4h, eigenvalue bounds, Richardson error and order).  120 cases, under two minutes.
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16} OMP_WAIT_POLICY=passive
here=$(cd "$(dirname "$0")" && pwd)
bad=0; n=0
for op in "" "--smoother gsrb" "--helmholtz" "--const-coeff" "--smoother jacobi" "--op 27pt" "--op 27pt --smoother gsrb" \
          "--op fv4 --smoother gsrb" "--op fv4" "--op fv2" "--periodic" "--periodic --smoother gsrb --helmholtz"; do
  for sz in "4 27" "5 1" "6 1" "4 64" "7 1" "4 125" "4 343" "4 729" "5 216" "6 27"; do
    r=$(timeout 600 bash "$here/compare_cli.sh" $op $sz 2>&1 | grep -E "PARITY|failed" | head -1)
    n=$((n+1)); case "$r" in "PARITY OK"*) ;; *) bad=$((bad+1)); echo "$r";; esac
  done
done
echo "parity sweep: $n cases, $bad mismatches"
exit $bad

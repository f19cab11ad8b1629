#!/bin/bash
# tools/parity_sweep_modes.sh -- run on the GPU box: the HIP executable vs the CPU oracle executable for the OTHER modes of the host layer -- MGSolve
# (--vcycles), the U-cycle ladder (--ucycles), V-cycles after the F-cycle (--unlimit), the CG bottom solver, MGPCG -- over every plugin and a few
# decompositions incl. odd box counts; every pinned line (f-cycle / v-cycle / iter norms, iteration counts, eigenvalue bounds, Richardson estimate).
# One OpenMP thread for the oracle: several of these modes hang on sums over many boxes, which the plugin forms in the reference's one-thread order.
export OMP_NUM_THREADS=1 OMP_WAIT_POLICY=passive
here=$(cd "$(dirname "$0")" && pwd)
bad=0; n=0
for mode in "--vcycles" "--ucycles" "--unlimit" "--bottom-solver cg" "--mgpcg" "--ucycles --vcycles" "--ucycles --bottom-solver cg" "--unlimit --bottom-solver cg"; do
  for op in "" "--helmholtz" "--smoother gsrb" "--op 27pt --smoother gsrb" "--op fv4 --smoother gsrb" "--op fv2" "--periodic"; do
    for sz in "4 8" "4 27" "5 8"; do
      r=$(timeout 900 bash "$here/compare_cli.sh" $mode $op $sz 2>&1 | grep -E "PARITY|failed" | head -1)
      n=$((n+1)); case "$r" in "PARITY OK"*) ;; *) bad=$((bad+1)); echo "$r";; esac
    done
  done
done
echo "parity sweep (modes): $n cases, $bad mismatches"
exit $bad

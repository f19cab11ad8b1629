export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for args in "--helmholtz 7 8" "--smoother gsrb 7 8" "--const-coeff 7 8" "--op 27pt 7 8" "--op fv4 --smoother gsrb 7 8" "--helmholtz 8 8"; do
  echo "[$args]"; timeout 120 hpgmg_amd/bin/hpgmg-fv $args --warmup 3 --solves 20 2>&1 | grep -E "DOF/s" | head -1
done

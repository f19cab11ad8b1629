export OMP_NUM_THREADS=8 OMP_WAIT_POLICY=passive
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for args in "--fp32-smoother 7 8" "--const-coeff --fp32-smoother 7 8"; do
  echo "[$args]"; timeout 300 hpgmg_amd/bin/hpgmg-fv $args --warmup 3 --solves 10 2>&1 | grep -E "DOF/s" | head -1
done

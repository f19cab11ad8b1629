export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for args in "--op 27pt 7 8" "--op fv4 --smoother gsrb 7 8" "--op fv2 7 8"; do
  echo "[$args]"; timeout 300 hpgmg_amd/bin/hpgmg-fv $args --warmup 2 --solves 8 2>&1 | grep -E "DOF/s" | head -1
done

export OMP_NUM_THREADS=8 OMP_WAIT_POLICY=passive
timeout 1500 python -m pytest tests/test_gpu_fcycle_parity.py -m gpu -x -q 2>&1 | tail -12
for args in "--periodic 7 8" "--periodic --helmholtz 7 8"; do
  echo "[$args]"; timeout 300 hpgmg_amd/bin/hpgmg-fv $args --warmup 2 --solves 5 2>&1 | grep -E "DOF/s|f-cycle" | sort | uniq -c | sort -rn | head -4
done

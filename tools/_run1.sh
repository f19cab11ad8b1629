export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for args in "--helmholtz 7 8" "--smoother gsrb 7 8" "--const-coeff 7 8"; do
  echo "[$args]"; timeout 120 hpgmg_amd/bin/hpgmg-fv $args --warmup 3 --solves 20 2>&1 | grep -E "DOF/s|Bottom solver"
done
echo "[bottom off]"; HPGMG_FUSED_BOTTOM=0 timeout 120 hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 3 --solves 20 2>&1 | grep -E "DOF/s|Bottom solver"

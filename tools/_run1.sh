for args in "--helmholtz 7 8" "--smoother gsrb 7 8" "--const-coeff 7 8"; do
  echo "[$args]"; timeout 120 hpgmg_amd/bin/hpgmg-fv $args --warmup 3 --solves 20 2>&1 | grep -E "DOF/s"
done
echo "[tail off]"; HPGMG_FUSED_TAIL=0 timeout 120 hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 3 --solves 20 2>&1 | grep -E "DOF/s"

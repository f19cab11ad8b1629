export OMP_NUM_THREADS=8 OMP_WAIT_POLICY=passive
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 3 --solves 20 2>&1 | grep -E "DOF/s" | head -3

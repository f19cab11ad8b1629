export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5

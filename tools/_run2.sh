export OMP_NUM_THREADS=8 OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | tail -15

export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 1200 python -m pytest tests/test_gpu_operators.py -m gpu -x -q -k fused_chebyshev 2>&1 | tail -15

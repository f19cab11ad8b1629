export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5

for m in 8 16 32 8 32; do
HPGMG_TUNE_27PT_RB_BOX_MAXDIM=$m python bench.py --workload config3-27pt --no-also --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rb_box maxdim', $m, round(d['ms_per_step'],3), d['config']['parity_ok'])"
done

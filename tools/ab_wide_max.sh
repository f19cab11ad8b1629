for w in config3-fv4 config3-27pt; do for mx in 64 32 16; do
HPGMG_TUNE_BRICK_WIDE_MAX=$mx python bench.py --workload $w --no-also --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', $mx, round(d['ms_per_step'],3), d['config']['parity_ok'])"
done; done

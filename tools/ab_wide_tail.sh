# usage (GPU box): bash tools/ab_wide_tail.sh -- config 3 with the one-box levels left to the single-workgroup tail from 8^3 / 4^3 / only the bottom solve (HPGMG_TUNE_BRICK_WIDE_TAIL = 8 / 4 / 2)
for w in config3-fv4 config3-27pt; do for t in 2 4 8 2; do
HPGMG_TUNE_BRICK_WIDE_TAIL=$t python bench.py --workload $w --no-also --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'tail', $t, round(d['ms_per_step'],3), d['config']['parity_ok'])"
done; done

# usage (GPU box): bash tools/ab_ftail.sh -- configs 1 and 2 with the F-cycle tail starting at 16^3 (default) or 8^3 (HPGMG_TUNE_FTAIL_MAX=8: the 16^3 level of the climb as bricks)
for w in config1 config2; do for m in 16 8 16 8; do
HPGMG_TUNE_FTAIL_MAX=$m python bench.py --workload $w --no-also --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'ftail from', $m, round(d['ms_per_step'],4), d['config']['parity_ok'])"
done; done

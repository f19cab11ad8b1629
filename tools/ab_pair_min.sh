# same-box A/B: the smallest level (cells) that takes the sweep-pair kernel -- 4 M (the 256^3 level only) vs 2 M (config 2's 128^3 level too: boxes of 64, two per row)
for rep in 1 2; do for m in 4000000 2000000; do
HPGMG_PAIR_MIN_CELLS=$m python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min cells $m', round(d['ms_per_step'],4), d['config']['parity_ok'])"
done; done

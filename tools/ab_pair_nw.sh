# same-box A/B: waves per workgroup of the sweep-pair kernel (HPGMG_TUNE_PAIR_NW = 10 | 12 | 16, 0 = the cost model's choice), optional k chunk
for v in 0 16 12 10 0 16 12 10; do
for kc in 0 $EXTRA_KC; do
HPGMG_TUNE_PAIR_NW=$v HPGMG_TUNE_PAIR_KC=$kc python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nw $v kc $kc', round(d['ms_per_step'],4), d['roofline']['avg_launch_us'], d['config']['parity_ok'])"
done; done

# same-box A/B: waves per workgroup of the sweep-pair kernel (experiment builds -DHPGMG_EXP_PAIR_NW=n kept under build_variants/), optional k chunk
for v in 16 12 10 8 14 16 12; do
cp build_variants/nw$v.so hpgmg_amd/libhpgmg_hip.so
for kc in 0 $EXTRA_KC; do
HPGMG_TUNE_PAIR_KC=$kc python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nw $v kc $kc', round(d['ms_per_step'],4), d['roofline']['avg_launch_us'], d['config']['parity_ok'])"
done; done

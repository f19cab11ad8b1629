# usage (GPU box): bash tools/ab_pair_kc.sh -- config 2 with the sweep-pair kernel's k chunk forced (HPGMG_TUNE_PAIR_KC): 0 = the launcher's choice (20 at 256^3)
for kc in 0 43 20 43 0; do
HPGMG_TUNE_PAIR_KC=$kc python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KC', $kc, round(d['ms_per_step'],4), d['roofline']['avg_launch_us'], d['config']['parity_ok'])"
done

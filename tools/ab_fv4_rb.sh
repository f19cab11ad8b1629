for cfg in "" "HPGMG_TUNE_FV4_RB_ORDER=0" "HPGMG_TUNE_FV4_RB_KCHUNK=64" "HPGMG_TUNE_FV4_RB_KCHUNK=32"; do
env $cfg python bench.py --workload config3-fv4 --no-also --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', round(d['ms_per_step'],3), d['roofline']['avg_launch_us'], d['config']['parity_ok'])"
done

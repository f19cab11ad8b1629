# usage: bash tools/ab_env.sh "<workloads>" "<env assignment or empty>" ... -- one bench line per (setting, workload), twice, on one box
wls="$1"; shift
for rep in 1 2; do for cfg in "$@"; do for wl in $wls; do
env $cfg python bench.py --workload $wl --no-also --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', '$wl', round(d['ms_per_step'],4), d['config']['parity_ok'])"
done; done; done

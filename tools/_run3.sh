cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
mkdir -p gpurun_out/sweep
for cfg in "8 16" "8 8" "8 32" "4 16" "4 8"; do
  set -- $cfg; export HPGMG_TUNE_WIDE_WJ=$1 HPGMG_TUNE_KCHUNK=$2; tag=w$1k$2
  rm -rf gpurun_out/sweep/kt$tag gpurun_out/sweep/pmc$tag
  echo "[$tag]"; timeout 120 hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 3 --solves 20 2>&1 | grep -E "DOF/s" | head -1
  timeout 200 rocprofv3 --kernel-trace -d gpurun_out/sweep/kt$tag -o r -- hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 1 --solves 3 > /dev/null 2>&1 </dev/null
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/sweep/pmc$tag -o r -- hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 1 --solves 2 > /dev/null 2>&1 </dev/null
done

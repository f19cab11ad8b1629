cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/sweep
for cfg in "16 0" "16 13" "16 43" "8 0"; do
  set -- $cfg; export HPGMG_TUNE_PAIR_NW=$1 HPGMG_TUNE_PAIR_KC=$2; tag=p$1k$2
  rm -rf gpurun_out/sweep/kt$tag gpurun_out/sweep/pmc$tag
  echo "[$tag]"; timeout 120 hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 3 --solves 20 2>&1 | grep -E "DOF/s|f-cycle" | sort | uniq | head -3
  timeout 200 rocprofv3 --kernel-trace -d gpurun_out/sweep/kt$tag -o r -- hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 1 --solves 3 > /dev/null 2>&1 </dev/null
done

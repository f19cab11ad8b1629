cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_c; mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_c -o kt -- hpgmg_amd/bin/hpgmg-fv --helmholtz 7 8 --warmup 2 --solves 10 > gpurun_out/prof_c.log 2>&1 </dev/null

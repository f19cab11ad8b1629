cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "fv4:--op fv4 --smoother gsrb" "27pt:--op 27pt"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rm -rf gpurun_out/prof_$tag
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o kt -- hpgmg_amd/bin/hpgmg-fv $args 7 8 --warmup 1 --solves 3 > /dev/null 2>&1 </dev/null
  db=$(find gpurun_out/prof_$tag -name '*.db' | head -1)
  echo "== $tag"; python3 tools/rocprof_summary.py "$db" </dev/null | head -16
done

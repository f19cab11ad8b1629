# same-box A/B: the shortest k chunk of the tiled fv4 kernel (the 128^3 level of 7 64: boxes of 32^3) -- kernel totals from profiled runs
for kc in 2 8 16 2 8; do
export HPGMG_TUNE_FV4_KCHUNK_MIN=$kc
bash tools/quick_timeline.sh r06k_kc config3-fv4 5 >/dev/null
echo "kc_min=$kc: $(grep -E ' fv4_tile_kernel<5,1,16,32>' gpurun_out/r06k_kc_last_solve.txt | tail -1) | $(grep -E ' fv4_tile_kernel<5,3,16,32>' gpurun_out/r06k_kc_last_solve.txt | tail -1) | solve $(head -1 gpurun_out/r06k_kc_last_solve.txt | grep -oE '[0-9.]+ us from' )"
done

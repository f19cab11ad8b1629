# same-box A/B: R = F of FMGSolve's opening pass written past the caches on levels larger than the infinity cache (kernel time from a profiled run)
for nt in 1 0 1 0; do for w in config4 config3-fv4; do
export HPGMG_TUNE_COPY_NT=$nt
bash tools/quick_timeline.sh r06k_nt $w 5 >/dev/null
echo "nt=$nt $w: $(grep -E ' norm_copy_restrict_kernel' gpurun_out/r06k_nt_last_solve.txt | tail -1)  | solve $(head -1 gpurun_out/r06k_nt_last_solve.txt | grep -oE '[0-9.]+ us from' )"
done; done

cd $GRAFT_REPO_ROOT
for v in 0 1 2 3; do
  if [ $v -eq 0 ]; then BENCH_ARGS="--storage bf16+grads" bash tools/prof_step.sh r03_y_p$v; else BENCH_ARGS="--storage bf16+grads" bash tools/prof_step.sh r03_y_p$v GCMI_FWD_H_PER_CU=$v; fi
  grep "fwd_hd_kernel" gpurun_out/r03_y_p${v}_timeline.txt | awk '{printf "%s ", $5}'; echo
done

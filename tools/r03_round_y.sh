cd $GRAFT_REPO_ROOT
for t in 1024 768 1024 768; do
  bash tools/prof_step.sh r03_y_t$t GCMI_WIN_THREADS_TWO_STAGE=$t
  grep "SumAccMaxBwdOp" gpurun_out/r03_y_t${t}_timeline.txt | head -1 | cut -c1-130
done

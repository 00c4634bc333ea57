cd $GRAFT_REPO_ROOT
for t in 512 0 512 0; do
  if [ $t -eq 0 ]; then bash tools/prof_step.sh r03_y_d$t; else bash tools/prof_step.sh r03_y_d$t GCMI_WIN_THREADS=$t; fi
  grep "MaxOp" gpurun_out/r03_y_d${t}_timeline.txt | head -2 | cut -c1-130
done

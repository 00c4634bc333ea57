cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for env in "X=0" "GCMI_HEAD_WIDE_MIN=1" "GCMI_WIN_THREADS_TWO_STAGE=512"; do
  echo "== $env"
  bash tools/prof_step.sh r03_s $env && grep "head\|SumAccMaxBwd\|step span" gpurun_out/r03_s_timeline.txt | head -8 | cut -c1-130
done
GCMI_HEAD_WIDE_MIN=1 timeout -k 10 300 python -m pytest tests/test_gpu_fused_bwd.py tests/test_gpu_scale.py -q -x -k "not million" 2>&1 | tail -2

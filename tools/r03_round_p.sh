cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_head_wide.py tests/test_gpu_scale.py -q -x -k "pcba or wide_head" 2>&1 | tail -2 || exit 1
for wgs in 1024 512 256; do
  echo "GCMI_HEAD_WGRAD_WGS=$wgs"
  BENCH_ARGS="--batch 8192 --tasks 128" bash tools/prof_step.sh r03_p_pcba_$wgs GCMI_HEAD_WGRAD_WGS=$wgs && grep "head_bwd_wide\|head_wgrad" gpurun_out/r03_p_pcba_${wgs}_timeline.txt | head -2 | cut -c1-120
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py tests/test_gpu_model.py tests/test_gpu_fused_bwd.py tests/test_gpu_scale.py tests/test_gpu_bf16_stream.py tests/test_gpu_small.py -q -x > gpurun_out/r03_v_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r03_v_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_v_tests.log | head -20 | cut -c1-300; exit 1; fi
bash tools/prof_step.sh r03_v_f32 && grep "bn_finalize\|bn_bwd_params\|step span" gpurun_out/r03_v_f32_timeline.txt | cut -c1-120
BENCH_ARGS="--batch 8192 --tasks 128" bash tools/prof_step.sh r03_v_pcba

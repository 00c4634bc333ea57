cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_stream.py tests/test_gpu_kernels.py -q -x > gpurun_out/r03_z_tests.log 2>&1
rc=$?
tail -2 gpurun_out/r03_z_tests.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_z_tests.log | head -10 | cut -c1-250; exit 1; fi
BENCH_ARGS="--storage bf16+grads" bash tools/prof_step.sh r03_z_bf16g && grep "SumAccMaxBwdOp\|step span" gpurun_out/r03_z_bf16g_timeline.txt | cut -c1-130
BENCH_ARGS="--storage bf16+grads" bash tools/prof_step.sh r03_z_bf16gb && grep "SumAccMaxBwdOp\|step span" gpurun_out/r03_z_bf16gb_timeline.txt | cut -c1-130

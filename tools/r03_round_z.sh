cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fused_bwd.py tests/test_gpu_scale.py tests/test_gpu_model.py -q -x > gpurun_out/r03_z_tests.log 2>&1
rc=$?
tail -2 gpurun_out/r03_z_tests.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_z_tests.log | head -10 | cut -c1-250; exit 1; fi
bash tools/prof_step.sh r03_z_f32 && grep "fused_bwd_kernel<128\|step span" gpurun_out/r03_z_f32_timeline.txt | cut -c1-130
bash tools/prof_step.sh r03_z_f32b && grep "fused_bwd_kernel<128\|step span" gpurun_out/r03_z_f32b_timeline.txt | cut -c1-130

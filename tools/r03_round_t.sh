cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py tests/test_gpu_scale.py tests/test_gpu_bf16_stream.py tests/test_gpu_fused_bwd.py tests/test_gpu_model.py -q -x > gpurun_out/r03_t_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r03_t_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_t_tests.log | head -20 | cut -c1-300; exit 1; fi
bash tools/prof_step.sh r03_t_f32 && grep "readout\|step span" gpurun_out/r03_t_f32_timeline.txt | cut -c1-140
BENCH_ARGS="--storage bf16+grads" bash tools/prof_step.sh r03_t_bf16g && grep "readout\|step span" gpurun_out/r03_t_bf16g_timeline.txt | cut -c1-140
GCMI_READOUT_WINDOWS=0 timeout -k 10 300 python bench.py --profile-only --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('windows off', d['value'], d['ms_per_step'])"

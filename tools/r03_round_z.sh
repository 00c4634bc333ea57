cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_scale.py tests/test_gpu_bf16_stream.py tests/test_gpu_fused_bwd.py -q -x > gpurun_out/r03_z_tests.log 2>&1
rc=$?
tail -2 gpurun_out/r03_z_tests.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_z_tests.log | head -10 | cut -c1-250; exit 1; fi
for a in "" "--storage bf16" "--storage bf16+grads"; do
  timeout -k 10 300 python bench.py --profile-only --steps 20 --warmup 3 $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])" || exit 1
done

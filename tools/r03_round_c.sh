cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16_stream.py -q -s > gpurun_out/r03_c_tests.log 2>&1
rc=$?
grep -n "passed\|failed\|FAILED\|^E  .*Error\|^E  .*assert\|bf16 storage vs" gpurun_out/r03_c_tests.log | cut -c1-330
if [ $rc -ne 0 ]; then echo "tests failed or were killed: stopping"; exit 1; fi
for dma in 1 0; do
  GCMI_FWD_H_DMA=$dma timeout -k 10 300 python bench.py --profile-only --storage bf16 --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_c_bf16_dma$dma.json || exit 1
  python - $dma <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_c_bf16_dma%s.json"%sys.argv[1]).read())
print("dma",sys.argv[1],d["value"],d["ms_per_step"],d["kernel_ms_per_step"])
PY
done
BENCH_ARGS="--storage bf16" bash tools/prof_step.sh r03_c_bf16

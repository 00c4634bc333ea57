cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_bf16_stream.py -q -s > gpurun_out/r03_h_tests.log 2>&1
rc=$?
grep -n "passed\|failed\|FAILED\|^E  .*Error\|^E  .*assert\|| vs the\|fit losses" gpurun_out/r03_h_tests.log | cut -c1-330
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests were killed: stopping"; exit 1; fi
for st in bf16 bf16+grads; do
  timeout -k 10 300 python bench.py --profile-only --storage $st --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_h_$st.json || exit 1
  python - $st <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_h_%s.json"%sys.argv[1]).read())
print(sys.argv[1],d["value"],d["ms_per_step"],d["kernel_ms_per_step"])
PY
done

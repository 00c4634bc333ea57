cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16_stream.py tests/test_gpu_dist.py::test_bench_gpus_2_runs_two_ranks_or_refuses -q -s > gpurun_out/r03_f_tests.log 2>&1
rc=$?
grep -n "passed\|failed\|FAILED\|^E  .*Error\|^E  .*assert\|bf16 storage vs" gpurun_out/r03_f_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "tests failed or were killed: stopping"; exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for fps in 1 0; do
  GCMI_FUSED_POOL_SUM=$fps timeout -k 10 300 python bench.py --profile-only --storage bf16 --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_f_bf16_fps$fps.json || exit 1
  python - $fps <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_f_bf16_fps%s.json"%sys.argv[1]).read())
print("bf16 fused pool+sum",sys.argv[1],d["value"],d["ms_per_step"],d["kernel_ms_per_step"])
PY
done
timeout -k 10 400 python tools/debug/ref_seeds.py 2>&1 | grep "^seed"

# round 3, first measurement pass: changed GPU tests, the default bench line (with the bf16-storage item), kernel
# statistics + timeline + PMC traffic of the large-batch step in both storages.  Steps run only while none was killed.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_stream.py tests/test_gpu_scale.py tests/test_gpu_dist.py tests/test_gpu_small.py -q -s > gpurun_out/r03_a_tests.log 2>&1
rc=$?
grep -n "passed\|failed\|FAILED\|bf16 storage vs\|fp32 storage vs\|eval logits\|fit losses\|streaming kernels:\|outputs:\|gradients:\|GPU-f64 " gpurun_out/r03_a_tests.log | cut -c1-330
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests were killed: stopping"; exit 1; fi
timeout -k 10 600 python bench.py > gpurun_out/r03_a_bench.json 2> gpurun_out/r03_a_bench.err || { echo "bench failed"; tail -5 gpurun_out/r03_a_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_a_bench.json").read().strip().splitlines()[-1])
print("f32", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernel_ms_per_step"])
b=d.get("bf16_storage")
print("bf16", b and (b["value"], b["ms_per_step"], b["kernel_ms_per_step"]))
PY
bash tools/prof_step.sh r03_a_f32 && BENCH_ARGS="--storage bf16" bash tools/prof_step.sh r03_a_bf16 && \
PMC_OUT=gpurun_out/r03_a_pmc_traffic.json bash tools/pmc_passes.sh && \
BENCH_ARGS="--storage bf16" PMC_OUT=gpurun_out/r03_a_bf16_pmc_traffic.json bash tools/pmc_passes.sh

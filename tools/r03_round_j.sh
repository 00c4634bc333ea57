# round 3, final validation + artefacts: the whole GPU suite, smoke, the default bench line, kernel statistics of the three
# storages, PMC passes of the three storages
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r03_j_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r03_j_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_j_tests.log | head -20 | cut -c1-300; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests were killed: stopping"; exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/r03_j_bench.json 2> gpurun_out/r03_j_bench.err || { echo "bench failed"; tail -5 gpurun_out/r03_j_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_j_bench.json").read().strip().splitlines()[-1])
print("f32", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_real"))
for k in ("bf16_storage", "bf16_storage_and_gradient_streams"):
    b=d.get(k); print(k, b and (b["value"], b["ms_per_step"], b.get("step_traffic")))
print({k:v for k,v in d["config"].items() if "molecules_per_s" in k or "exact" in k})
print("tox21", {k:v for k,v in d["config"].get("tox21_real",{}).items() if "fit_" in k})
print("cpu", d.get("cpu_baseline"), d.get("cpu_baseline_large_batch"))
PY
bash tools/prof_step.sh r03_j_f32 && BENCH_ARGS="--storage bf16" bash tools/prof_step.sh r03_j_bf16 && BENCH_ARGS="--storage bf16+grads" bash tools/prof_step.sh r03_j_bf16g && \
PMC_OUT=gpurun_out/r03_j_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_j_pmc_f32.txt && \
BENCH_ARGS="--storage bf16" PMC_OUT=gpurun_out/r03_j_bf16_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_j_pmc_bf16.txt && \
BENCH_ARGS="--storage bf16+grads" PMC_OUT=gpurun_out/r03_j_bf16g_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_j_pmc_bf16g.txt && \
python - <<'PY'
import json
for f in ("gpurun_out/r03_j_pmc_traffic.json", "gpurun_out/r03_j_bf16_pmc_traffic.json", "gpurun_out/r03_j_bf16g_pmc_traffic.json"):
    k = json.load(open(f))["kernels"]
    steps = max(r.get("fetch_launches", 0) for n, r in k.items() if "readout_fwd" in n)
    tot = sum(r.get("hbm_bytes_per_launch", 0) * max(r.get("fetch_launches", 0), r.get("write_launches", 0)) for r in k.values()) / steps
    print(f, "bytes per step %.3f GB" % (tot / 1e9))
PY

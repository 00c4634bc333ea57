# round 3 artefacts: the default bench line, kernel statistics + timeline and the two PMC passes of the large-batch step
# in both storages
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/prof_step.sh r03_g_f32 && BENCH_ARGS="--storage bf16" bash tools/prof_step.sh r03_g_bf16 && \
PMC_OUT=gpurun_out/r03_g_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_g_pmc_f32.txt && \
BENCH_ARGS="--storage bf16" PMC_OUT=gpurun_out/r03_g_bf16_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_g_pmc_bf16.txt && \
python - <<'PY'
import json
for f in ("gpurun_out/r03_g_pmc_traffic.json", "gpurun_out/r03_g_bf16_pmc_traffic.json"):
    k = json.load(open(f))["kernels"]
    steps = max(r.get("fetch_launches", 0) for n, r in k.items() if "readout_fwd" in n)
    tot = sum(r.get("hbm_bytes_per_launch", 0) * max(r.get("fetch_launches", 0), r.get("write_launches", 0)) for r in k.values()) / steps
    print(f, "bytes per step %.3f GB" % (tot / 1e9))
PY

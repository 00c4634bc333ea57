cd $GRAFT_REPO_ROOT
PMC_OUT=gpurun_out/r03_f_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_f_pmc_f32.txt && \
BENCH_ARGS="--storage bf16" PMC_OUT=gpurun_out/r03_f_bf16_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_f_pmc_bf16.txt && \
BENCH_ARGS="--storage bf16+grads" PMC_OUT=gpurun_out/r03_f_bf16g_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_f_pmc_bf16g.txt && \
python - <<'PY'
import json
for f in ("gpurun_out/r03_f_pmc_traffic.json", "gpurun_out/r03_f_bf16_pmc_traffic.json", "gpurun_out/r03_f_bf16g_pmc_traffic.json"):
    k = json.load(open(f))["kernels"]
    steps = max(r.get("fetch_launches", 0) for n, r in k.items() if "readout_fwd" in n)
    tot = sum(r.get("hbm_bytes_per_launch", 0) * max(r.get("fetch_launches", 0), r.get("write_launches", 0)) for r in k.values()) / steps
    print(f, "bytes per step %.3f GB" % (tot / 1e9))
    for n, r in sorted(k.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0))[:3]:
        print("   ", n[-60:], r.get("hbm_bytes_per_launch"))
PY

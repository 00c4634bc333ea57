cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for wt in 1024 512; do
  GCMI_WIN_THREADS_TWO_STAGE=$wt timeout -k 10 300 python bench.py --profile-only --storage bf16+grads --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_i_wt$wt.json || exit 1
  python - $wt <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_i_wt%s.json"%sys.argv[1]).read())
print("two-stage threads",sys.argv[1],d["value"],d["ms_per_step"],d["kernel_ms_per_step"])
PY
done
BENCH_ARGS="--storage bf16+grads" bash tools/prof_step.sh r03_i_bf16g && \
BENCH_ARGS="--storage bf16+grads" PMC_OUT=gpurun_out/r03_i_bf16g_pmc_traffic.json bash tools/pmc_passes.sh > gpurun_out/r03_i_pmc_bf16g.txt && \
python - <<'PY'
import json
f="gpurun_out/r03_i_bf16g_pmc_traffic.json"
k = json.load(open(f))["kernels"]
steps = max(r.get("fetch_launches", 0) for n, r in k.items() if "readout_fwd" in n)
tot = sum(r.get("hbm_bytes_per_launch", 0) * max(r.get("fetch_launches", 0), r.get("write_launches", 0)) for r in k.values()) / steps
print(f, "bytes per step %.3f GB" % (tot / 1e9))
PY
head -16 gpurun_out/r03_i_bf16g_kernel_stats.csv | cut -d, -f1,2,4 | cut -c1-110

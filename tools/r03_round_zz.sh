# round 3, final validation + artefacts (second pass, after the wide head / MPNN / launch merges)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r03_zz_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r03_zz_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_zz_tests.log | head -20 | cut -c1-300; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests were killed: stopping"; exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/r03_zz_bench.json 2> gpurun_out/r03_zz_bench.err || { echo "bench failed"; tail -5 gpurun_out/r03_zz_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_zz_bench.json").read().strip().splitlines()[-1])
print("f32", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_real"))
for k in ("bf16_storage", "bf16_storage_and_gradient_streams"):
    b=d.get(k); print(k, b and (b["value"], b["ms_per_step"]))
c=d["config"]
print({k:v for k,v in c.items() if "molecules_per_s" in k or "exact" in k})
print("pcba", c.get("pcba_shape")); print("widths", c.get("widths_128_128_dense_256"))
print("tox21", {k:v for k,v in c.get("tox21_real",{}).items() if "fit_" in k})
print("head_gemm", d.get("head_gemm",{}).get("pcba"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_large_batch",{}).get("value"))
PY
bash tools/prof_step.sh r03_zz_f32 && BENCH_ARGS="--storage bf16" bash tools/prof_step.sh r03_zz_bf16 && BENCH_ARGS="--storage bf16+grads" bash tools/prof_step.sh r03_zz_bf16g && \
BENCH_ARGS="--batch 8192 --tasks 128" bash tools/prof_step.sh r03_zz_pcba
for b in 1024 4096; do
  timeout -k 10 300 python tools/kbench_mpnn.py --mols $b --steps 5 --cpu-mols 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mpnn', {k:d[k] for k in ('n_mols','train_step_ms','train_molecules_per_s')})"
done
timeout -k 10 300 python tools/kbench_weave.py 2>/dev/null | tail -1 | cut -c1-400

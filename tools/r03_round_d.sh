cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r03_d_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r03_d_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_d_tests.log | head -20 | cut -c1-300; echo "tests failed or were killed: stopping"; exit 1; fi
for pc in 3 2; do
  GCMI_FWD_H_PIECES=$pc timeout -k 10 300 python bench.py --profile-only --storage bf16 --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_d_bf16_p$pc.json || exit 1
  python - $pc <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_d_bf16_p%s.json"%sys.argv[1]).read())
print("bf16 pieces",sys.argv[1],d["value"],d["ms_per_step"],d["kernel_ms_per_step"])
PY
done
timeout -k 10 300 python bench.py --profile-only --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_d_f32.json || exit 1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_d_f32.json").read())
print("f32",d["value"],d["ms_per_step"],d["kernel_ms_per_step"])
PY

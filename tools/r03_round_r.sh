cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r03_r_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r03_r_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_r_tests.log | head -20 | cut -c1-300; exit 1; fi
for a in "" "--storage bf16+grads" "--batch 8192 --tasks 128"; do
  timeout -k 10 300 python bench.py --profile-only --steps 20 --warmup 3 $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])" || exit 1
done

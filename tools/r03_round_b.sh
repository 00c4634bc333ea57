# round 3, second pass: the tests that failed in pass A, then the bf16 step under the forward-product switches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_stream.py "tests/test_gpu_scale.py::test_one_pass_step_meets_the_oracle_at_16k_molecules" "tests/test_gpu_dist.py::test_two_ranks_on_the_small_batch_engine_reproduce_the_single_process_run" tests/test_gpu_small.py::test_bf16_storage_is_refused_only_where_no_kernel_has_it -q -s > gpurun_out/r03_b_tests.log 2>&1
rc=$?
grep -n "passed\|failed\|FAILED\|^E  .*Error\|^E  .*assert" gpurun_out/r03_b_tests.log | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests were killed: stopping"; exit 1; fi
for cfg in "3 2" "3 3" "2 3" "2 4" "3 1"; do
  set -- $cfg
  GCMI_FWD_H_PIECES=$1 GCMI_FWD_H_PER_CU=$2 timeout -k 10 300 python bench.py --profile-only --storage bf16 --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_b_bf16_p$1_c$2.json || exit 1
  python - $1 $2 <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_b_bf16_p%s_c%s.json"%(sys.argv[1],sys.argv[2])).read())
print("pieces",sys.argv[1],"per_cu",sys.argv[2],d["value"],d["ms_per_step"],d["kernel_ms_per_step"])
PY
done

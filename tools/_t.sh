for env in "GCMI_FUSED_PSUMS=1" "GCMI_FUSED_PSUMS=0" "GCMI_FUSED_BWD=0"; do
echo "== $env"; env $env timeout -k 10 300 python -m pytest tests/test_gpu_model.py -q -k "pipeline_equals_python or fast_path_equals" 2>&1 | grep -E "AssertionError: |passed|failed|np.float32" | head -6
done
for i in 1 2 3; do timeout -k 10 300 python -m pytest tests/test_gpu_small.py -q -k "bf16_storage_follows" 2>&1 | grep -E "assert \(np|passed|failed" | head -3; done

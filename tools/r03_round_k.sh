# PCBA-shaped head on the matrix cores: parity tests, the step timeline, the head product alone
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_head_wide.py tests/test_gpu_scale.py tests/test_gpu_model.py -q -x -k "pcba or wide_head" 2>&1 | tail -3 || exit 1
BENCH_ARGS="--batch 8192 --tasks 128" bash tools/prof_step.sh r03_k_pcba $@ && grep -n "head\|loss\|wgrad\|seg_gemm4\|step span" gpurun_out/r03_k_pcba_timeline.txt | cut -c1-150

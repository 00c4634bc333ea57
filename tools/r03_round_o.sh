cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py -q -x -k "regression_preset" -s > gpurun_out/r03_o_widths.log 2>&1
grep -n "outputs:\|gradients:\|median entry\|passed\|failed\|Error" gpurun_out/r03_o_widths.log | head -20

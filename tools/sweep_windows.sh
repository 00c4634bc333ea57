set -e
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "window or gather" > gpurun_out/t8.log 2>&1 || { tail -n 30 gpurun_out/t8.log; exit 1; }
tail -n 3 gpurun_out/t8.log
for cap in 48 64 96 128; do for wt in 256 512; do
 echo "== cap $cap wt $wt"
 GCMI_WIN_THREADS=$wt timeout -k 10 120 python tools/kbench.py --only gather --win-cap $cap 2>&1 | grep -E "windows|gather_.*F(64|76)"
done; done > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log

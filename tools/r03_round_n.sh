# MPNN: kernel statistics of the step at 4 096 molecules (+ the gemm / kernel tests that cover the chunked wgrad)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py -q -x 2>&1 | tail -2 || exit 1
rm -rf gpurun_out/prof_mpnn
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mpnn -- python3 tools/kbench_mpnn.py --mols 4096 --steps 5 --cpu-mols 2 > gpurun_out/prof_mpnn.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_mpnn > gpurun_out/r03_n_mpnn_kernel_stats.csv
head -24 gpurun_out/r03_n_mpnn_kernel_stats.csv | cut -c1-150
wc -l gpurun_out/r03_n_mpnn_kernel_stats.csv
rm -rf gpurun_out/prof_mpnn

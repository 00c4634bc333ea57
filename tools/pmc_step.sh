# One counter pass over a bench step: per-kernel averages of the counters named on the command line.
# Keep to at most two counters of one hardware block (TA_*, TCP_*, TCC_*) per pass: three TA counters plus two TCP
# counters made rocprofv3 replay/hang until the timeout on this pool.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_s
timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_s -- python3 bench.py --steps 2 --warmup 1 --profile-only > gpurun_out/pmc_s.log 2>&1
python3 tools/pmc_kernels.py gpurun_out/pmc_s "" | grep -E "seg_gemm4|wgrad3|col_sums|win_kernel|bn_bwd_dx|readout_fwd" 
rm -rf gpurun_out/pmc_s

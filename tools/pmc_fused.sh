# Counter passes over the one-pass block kernels inside the bench step (kernel trace + PMC only)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { # name, counters...
  n=$1; shift
  rm -rf gpurun_out/pmc_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$n -- python3 bench.py --profile-only --steps 2 --warmup 1 > gpurun_out/pmc_$n.log 2>&1
  python3 tools/pmc_kernels.py gpurun_out/pmc_$n fused >> gpurun_out/pmc_fused_summary.txt
  rm -rf gpurun_out/pmc_$n
}
: > gpurun_out/pmc_fused_summary.txt
run a SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run b SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY
run d SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
cat gpurun_out/pmc_fused_summary.txt

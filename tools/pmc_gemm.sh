set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { # name, counters...
  n=$1; shift
  rm -rf gpurun_out/pmc_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$n -- python3 tools/kbench.py --only ${KB_ONLY:-seg_gemm} --iters 2 > gpurun_out/pmc_$n.log 2>&1
  python3 tools/pmc_kernels.py gpurun_out/pmc_$n ${KB_FILTER:-seg_gemm} >> gpurun_out/pmc_summary.txt
  rm -rf gpurun_out/pmc_$n
}
: > gpurun_out/pmc_summary.txt
run a SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run b SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
run c TCP_PENDING_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TCC_EA0_WRREQ_STALL TCP_TCR_TCP_STALL_CYCLES TCC_TAG_STALL
run d SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM
cat gpurun_out/pmc_summary.txt

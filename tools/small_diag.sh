#!/bin/bash
# Timing experiments on the small-batch engine: the same fit under rocprofv3 with parts of the kernels switched
# off (GCMI_SMALL_DIAG bits: 1 BatchNorm-statistics atomics, 2 readout atomics, 4 dense weight-gradient atomics,
# 8 matrix products, 16 neighbour gathers of the forward).  Results of those runs are wrong on purpose.
# The switches are compiled only into a diagnostic build: rebuild smallstep.hip with -DGCMI_SMALL_DIAG_BUILD first
# (GCMI_EXTRA_HIPCC_FLAGS=-DGCMI_SMALL_DIAG_BUILD python -m deepchem_amd._build --force); the shipped library ignores
# GCMI_SMALL_DIAG.
cd /tmp && export TMPDIR=/tmp
for d in ${DIAGS:-0 1 2 4 8 16}; do
  export GCMI_SMALL_DIAG=$d
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/diag_$d -o small -- python3 $GRAFT_REPO_ROOT/tools/small_profile.py ${BATCH:-64} ${MODE:-reference} 2 > $GRAFT_REPO_ROOT/gpurun_out/diag_$d.log 2>&1
  echo "== GCMI_SMALL_DIAG=$d"
  grep fit_molecules $GRAFT_REPO_ROOT/gpurun_out/diag_$d.log
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/diag_$d/small_results.db small_ | grep "small_"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/diag_$d
done

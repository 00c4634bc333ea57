# Phase clocks and phase switches of fwd_hd_kernel (csrc/fwd_bf16.hip): a DIAGNOSTIC BUILD of the library on the GPU box
# (-DGCMI_FWD_H_DIAG_BUILD; the shipped library has neither the switches nor the clocks), then the bf16 bench step under
# GCMI_FWD_H_DIAG = 0 (clocks only), 1 (no products), 2 (no global stores), 4 (no operand loads).  Results with a switch
# on are wrong on purpose.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GCMI_EXTRA_HIPCC_FLAGS=-DGCMI_FWD_H_DIAG_BUILD python -m deepchem_amd._build --force > gpurun_out/diag_build.log 2>&1 || { tail -5 gpurun_out/diag_build.log; exit 1; }
for f in ${FLAGS:-0 1 2 4 7}; do
  echo "== GCMI_FWD_H_DIAG=$f"
  GCMI_FWD_H_DIAG=$f timeout -k 10 300 python bench.py --profile-only --storage bf16 --steps 4 --warmup 1 2> gpurun_out/fwd_h_diag_$f.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step']['seg_gemm'])"
  grep "fwd_hd<" gpurun_out/fwd_h_diag_$f.err | sort | uniq -c | sort -rn | head -6
done

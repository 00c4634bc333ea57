cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in bf16 fp32; do
  GCMI_FUSED_DIAG=1 timeout -k 10 300 python bench.py --profile-only --storage $st --steps 3 --warmup 1 2> gpurun_out/r03_e_fused_diag_$st.err | tail -1 > /dev/null
  echo "== $st"; grep "fused_bwd<" gpurun_out/r03_e_fused_diag_$st.err | tail -6
done

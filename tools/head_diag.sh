# Phase clocks of the wide head kernels (csrc/head_bwd.hip): a DIAGNOSTIC BUILD of the library on the GPU box
# (-DGCMI_HEAD_DIAG_BUILD; the shipped library has no clocks), then a few PCBA-shaped steps.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GCMI_EXTRA_HIPCC_FLAGS=-DGCMI_HEAD_DIAG_BUILD python -m deepchem_amd._build --force > gpurun_out/diag_build.log 2>&1 || { tail -5 gpurun_out/diag_build.log; exit 1; }
timeout -k 10 300 python bench.py --profile-only --batch 8192 --tasks 128 --steps 3 --warmup 1 2> gpurun_out/head_diag.err | tail -1 | cut -c1-120
grep head_diag gpurun_out/head_diag.err | tail -9

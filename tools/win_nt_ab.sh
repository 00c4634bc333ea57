# A/B: the window kernels' fp32 row stores as ordinary or as non-temporal stores (-DGCMI_WIN_NT=1): bench step + gather-sum time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nt in 0 1 2 0 1 2; do
  GCMI_EXTRA_HIPCC_FLAGS=-DGCMI_WIN_NT=$nt python -m deepchem_amd._build --force > gpurun_out/nt_build.log 2>&1 || { tail -5 gpurun_out/nt_build.log; exit 1; }
  timeout -k 10 300 python bench.py --profile-only --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('nt=$nt', d['ms_per_step'], 'gather_sum', k.get('gather_sum'), 'gather_max', k.get('gather_max'), 'max_bwd', k.get('gather_max_bwd'), 'roofline', d['roofline']['achieved'])" || exit 1
done
python -m deepchem_amd._build --force > gpurun_out/nt_build.log 2>&1

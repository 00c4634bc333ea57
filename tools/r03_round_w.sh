cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mpnn_model.py tests/test_gpu_mpnn.py -q -x 2>&1 | tail -3 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_dist.py -q -x -k "family" 2>&1 | tail -2 || exit 1
for b in 1024 4096 8192; do
  timeout -k 10 300 python tools/kbench_mpnn.py --mols $b --steps 5 --cpu-mols 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('n_mols','train_step_ms','train_molecules_per_s','forward_ms','edge_moments_forward','edge_moments_backward','edge_moments_forward_per_atom_kernel','edge_moments_kernels_max_rel_diff')})" || exit 1
done

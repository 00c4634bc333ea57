# MPNN after the flat step: model tests, two-rank test, step rate against batch size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mpnn_model.py tests/test_gpu_mpnn.py tests/test_gpu_weave.py -q -x 2>&1 | tail -4 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_dist.py -q -x -k "family or mpnn or weave" 2>&1 | tail -3 || exit 1
for b in 1024 4096 8192; do
  timeout -k 10 300 python tools/kbench_mpnn.py --mols $b --steps 5 --cpu-mols 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('n_mols','n_atoms','n_pairs','train_step_ms','train_molecules_per_s','forward_ms')})" || exit 1
done

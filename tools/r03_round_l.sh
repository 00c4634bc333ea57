# MPNN: step rate against batch size, kernel statistics of one step shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 1024 4096 8192; do
  timeout -k 10 300 python tools/kbench_mpnn.py --mols $b --steps 5 --cpu-mols 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('n_mols','n_atoms','n_pairs','train_step_ms','train_molecules_per_s','forward_ms')})" || exit 1
done
rm -rf gpurun_out/prof_mpnn
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mpnn -- python3 tools/kbench_mpnn.py --mols 4096 --steps 3 --cpu-mols 2 > gpurun_out/prof_mpnn.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_mpnn > gpurun_out/r03_l_mpnn_kernel_stats.csv
head -30 gpurun_out/r03_l_mpnn_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_mpnn

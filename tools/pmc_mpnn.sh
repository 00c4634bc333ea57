# HBM traffic per launch of the MPNN step's kernels (tools/kbench_mpnn.py at 4 096 molecules): two counter passes
# (FETCH_SIZE, WRITE_SIZE), kernel trace only -> gpurun_out/r03_mpnn_pmc_traffic.json
OUT=${PMC_OUT:-gpurun_out/r03_mpnn_pmc_traffic.json}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f -- python3 tools/kbench_mpnn.py --mols 4096 --steps 2 --cpu-mols 2 > gpurun_out/pmc_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w -- python3 tools/kbench_mpnn.py --mols 4096 --steps 2 --cpu-mols 2 > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w $OUT
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
python3 - $OUT <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))["kernels"]
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get("hbm_bytes_per_launch",0))[:14]:
    print(k[-70:], v.get("fetch_bytes_per_launch"), v.get("write_bytes_per_launch"), v.get("hbm_bytes_per_launch"))
PY

# HBM traffic per launch of every kernel of one bench step: two counter passes (FETCH_SIZE, WRITE_SIZE), kernel trace only
# BENCH_ARGS: extra bench.py arguments (e.g. "--storage bf16"); PMC_OUT: output file (default gpurun_out/pmc_traffic.json)
OUT=${PMC_OUT:-gpurun_out/pmc_traffic.json}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --profile-only ${BENCH_ARGS} > gpurun_out/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --profile-only ${BENCH_ARGS} > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w $OUT
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
python3 - $OUT <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))["kernels"]
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get("hbm_bytes_per_launch",0))[:16]:
    print(k[-70:], v.get("fetch_bytes_per_launch"), v.get("write_bytes_per_launch"), v.get("hbm_bytes_per_launch"))
PY

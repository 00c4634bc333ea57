# rehearsal of the driver's N=2 bench launch on the one-GPU box (gloo backend, both ranks on GPU 0: numbers mean nothing,
# the code path -- parameter broadcast, flat all-reduce, barriers, every collective leg -- is the real one)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GCMI_BENCH_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03_q_bench_2ranks.json 2> gpurun_out/r03_q_bench_2ranks.err
rc=$?
echo "rc=$rc"; tail -3 gpurun_out/r03_q_bench_2ranks.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_q_bench_2ranks.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d["config"].get("pcba_shape"), {k:v for k,v in d["config"].get("tox21_real",{}).items() if "fit_" in k})
PY

# Round-end evidence, short form (one pass of the GPU suite): suite, smoke, bench, kernel statistics and a one-step
# timeline of the same bench command.  Outputs under gpurun_out/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -q -m gpu > gpurun_out/final_tests.log 2>&1 || { tail -30 gpurun_out/final_tests.log; exit 1; }
tail -1 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench.json
rm -rf gpurun_out/prof_final
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -- python3 bench.py --profile-only --steps 5 --warmup 2 > gpurun_out/prof_final.log 2>&1
python tools/step_timeline.py gpurun_out/prof_final > gpurun_out/final_timeline.txt 2>&1 || true
python tools/kernel_stats.py gpurun_out/prof_final > gpurun_out/final_kernel_stats.csv
rm -rf gpurun_out/prof_final
head -c 300 gpurun_out/final_bench.json; echo
python -c "
import json;d=json.load(open('gpurun_out/final_bench.json'));print(json.dumps(d['head_gemm']))"

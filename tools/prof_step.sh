# Kernel statistics + one-step timeline of the large-batch bench step: tools/prof_step.sh <tag> [ENV=VAL ...]
# (BENCH_ARGS="--storage bf16" etc. are passed to bench.py)
# -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_timeline.txt
set -e
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --profile-only --steps 20 --warmup 3 ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/${tag}_bench.json
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python3 bench.py --profile-only --steps 5 --warmup 2 ${BENCH_ARGS} > gpurun_out/prof_$tag.log 2>&1
python tools/step_timeline.py gpurun_out/prof_$tag > gpurun_out/${tag}_timeline.txt 2>&1 || true
python tools/kernel_stats.py gpurun_out/prof_$tag > gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
python -c "
import json,sys
d=json.loads(open('gpurun_out/${tag}_bench.json').read())
print('$tag', d['value'], d['ms_per_step'])"

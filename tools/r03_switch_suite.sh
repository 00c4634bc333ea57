# (bf16 activation storage refuses these switches by design -- "never another arithmetic" -- so its tests are left out)
# The GPU suite once under each of the switches given as arguments ("NAME=VALUE" ...): the non-default paths after this round's changes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for kv in "$@"; do
  name=$(echo $kv | tr '=' '_')
  env $kv timeout -k 10 380 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_dist.py --deselect tests/test_gpu_bf16_stream.py -k "not bf16" > gpurun_out/r03_switch_$name.log 2>&1
  rc=$?
  echo "$kv rc=$rc: $(tail -1 gpurun_out/r03_switch_$name.log | cut -c1-200)"
  if [ $rc -ne 0 ]; then grep -n "^E  \|FAILED" gpurun_out/r03_switch_$name.log | head -8 | cut -c1-250; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
done

# SURVEY.md 8d configuration R1: the gather / readout kernels alone on N = 2^22 atoms (Tox21 degree mix, ~230 k
# molecules, E ~ 8.7 M directed edges), F in {64, 75 (stored as 76), 128}: HIP-event timings with algorithmic GB/s,
# rocprofv3 kernel statistics of the same command, and the two HBM counter passes (FETCH_SIZE, WRITE_SIZE; separate
# runs, kernel trace only).  Usage on the GPU box:  bash tools/r1_roofline.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MOLS=${MOLS:-230456}
ONLY=gather_sum,gather_max,readout
python3 tools/kbench.py --mols $MOLS --only $ONLY --iters 10 > gpurun_out/r1_kbench.log 2>&1
tail -1 gpurun_out/r1_kbench.log > gpurun_out/r1_kbench.json
rm -rf gpurun_out/r1_stats gpurun_out/r1_f gpurun_out/r1_w
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r1_stats -o r1 -- python3 tools/kbench.py --mols $MOLS --only $ONLY --iters 5 > gpurun_out/r1_stats.log 2>&1
python3 tools/rocpd_summary.py $(ls gpurun_out/r1_stats/*.db | head -1) "" > gpurun_out/r1_kernel_stats.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r1_f -o f -- python3 tools/kbench.py --mols $MOLS --only $ONLY --iters 2 > gpurun_out/r1_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r1_w -o w -- python3 tools/kbench.py --mols $MOLS --only $ONLY --iters 2 > gpurun_out/r1_w.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/r1_f gpurun_out/r1_w gpurun_out/r1_pmc_traffic.json
rm -rf gpurun_out/r1_stats gpurun_out/r1_f gpurun_out/r1_w
cat gpurun_out/r1_kbench.json
head -14 gpurun_out/r1_kernel_stats.txt

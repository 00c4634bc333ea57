#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer build of the HOST C++ of libgcmi.so (core.cpp, collate.cpp,
# featurize.cpp: index arithmetic over caller-supplied arrays, a backtracking kekuliser fed arbitrary SMILES) and
# the host test files run against it: the featurizer's fuzz test, the collation tests, the C-ABI host tests.
# CPU only (the GPU pool refuses sanitizer runs): the host side is instrumented (-fno-gpu-sanitize leaves the few
# device functions of collate.cpp alone); the other kernel files are not in this library and nothing here touches a GPU.
#   bash tools/asan_host.sh [pytest arguments]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/deepchem_amd/csrc/build/libgcmi_host_asan.so
mkdir -p $(dirname $OUT)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -fno-gpu-sanitize -x hip -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined \
  -fno-omit-frame-pointer -shared $ROOT/deepchem_amd/csrc/core.cpp $ROOT/deepchem_amd/csrc/collate.cpp \
  $ROOT/deepchem_amd/csrc/featurize.cpp -o $OUT -lpthread
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $ROOT
TESTS=${@:-tests/test_featurizer.py tests/test_collate_plans.py tests/test_mol_graphs.py tests/test_resident.py tests/test_atom_codes.py}
GCMI_HOST_ONLY_LIB=$OUT LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest $TESTS -x -q -m "not gpu" -p no:cacheprovider

"""The host C++ of libgcmi.so (collation, SMILES featurizer) under AddressSanitizer + UndefinedBehaviorSanitizer:
tools/asan_host.sh builds core.cpp / collate.cpp / featurize.cpp with -fsanitize=address,undefined (CPU build; the
GPU pool does not run sanitizers) and runs the featurizer fuzz test and the collation tests against that library."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_host_code_is_clean_under_asan_and_ubsan():
    if os.environ.get("GCMI_HOST_ONLY_LIB"):
        pytest.skip("already running inside the sanitizer build")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here: the sanitizer build needs the compiler")
    done = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh")], capture_output=True, text=True,
                          timeout=580)
    tail = (done.stdout + done.stderr)[-3000:]
    assert done.returncode == 0, tail
    assert "passed" in done.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail

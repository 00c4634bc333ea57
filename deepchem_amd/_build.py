"""Build libgcmi.so (HIP kernels + C ABI) in-tree for gfx950.

``hipcc`` cross-compiles without a GPU, so this runs in the build container
(``__graft_entry__.build()``) and the resulting ``deepchem_amd/csrc/libgcmi.so``
travels to the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libgcmi.so")
SOURCES = ["core.cpp", "collate.cpp", "featurize.cpp", "gather.hip", "gather_lds.hip", "readout.hip", "bn.hip", "gemm.hip", "gemm_split.hip", "bwd_fused.hip", "fwd_fused.hip", "fwd_bf16.hip", "head_bwd.hip", "loss.hip", "weave.hip", "mpnn.hip", "model.hip", "smallstep.hip"]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    deps.append(os.path.join(os.path.dirname(CSRC), "..", "include", "gcmi.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libgcmi.so cannot be built here")
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(CSRC, "build", src.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-x", "hip"] + \
            os.environ.get("GCMI_EXTRA_HIPCC_FLAGS", "").split() + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out.strip():
            sys.stderr.write(out.decode(errors="replace"))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs + ["-lpthread"]
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)

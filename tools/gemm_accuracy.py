"""Error of the segmented GEMM kernels against an fp64 product (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepchem_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, K, n_out = 200000, 64, 64
a1 = torch.randn(N, K, generator=g).to(dev)
a2 = torch.randn(N, K, generator=g).to(dev)
w = (torch.randn(2 * K * n_out, generator=g) * 0.2).to(dev)
bias = torch.randn(n_out, generator=g).to(dev)
out = ops.seg_gemm([0], [N], a1, w, [0], a2, w, [K * n_out], bias, [0], n_out, False, True, N, K, K)
ref = torch.relu(a1.double() @ w[:K * n_out].view(K, n_out).double() + a2.double() @ w[K * n_out:].view(K, n_out).double()
                 + bias.double())
err = (out.double() - ref).abs()
print("kernel", "v3" if os.environ.get("GCMI_GEMM_V3", "1") != "0" else "v2", "max abs err", float(err.max()),
      "rel to max", float(err.max() / ref.abs().max()), "mean abs err", float(err.mean()))
ref32 = torch.relu(a1 @ w[:K * n_out].view(K, n_out) + a2 @ w[K * n_out:].view(K, n_out) + bias)
print("torch fp32 matmul max abs err", float((ref32.double() - ref).abs().max()), "mean", float((ref32.double() - ref).abs().mean()))

"""Gradient error of the one-pass backward and of the separate kernels against the oracle (torch-CPU fp32 autograd of
the reference's layers) on the same batch: max |g - g_ref| / max |g_ref| per parameter tensor, worst six of each."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_gpu_fused_bwd import _step
from tests.util import oracle_batch, oracle_convmols
from oracle import graphconv_oracle as O
from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules

n, tasks = 400, 4
packed = synthetic_molecules(n, seed=21, max_atoms=35)
y, w = synthetic_labels(n, tasks, "classification", 21, pos_rate=0.4)
cfg = O.ModelConfig(tasks, batch_size=n)
state = O.init_state(cfg, 21)
tr = O.OracleTrainer(cfg, state, grad_mode="full")
inputs, labels, weights = oracle_batch(cfg, oracle_convmols(packed), y, w, np.arange(n), n, True)
ref, _ = tr.loss(inputs, labels, weights)
ref.backward()
ref_grads = tr.grads()
for fused in (True, False):
    loss, grads, slices, rng, _ = _step(packed, y, w, tasks, "full", fused, state=state)
    errs = []
    for name, (off, cnt) in slices:
        if ref_grads.get(name) is None:
            continue
        a = grads[off:off + cnt].double().cpu().numpy()
        b = np.asarray(ref_grads[name], np.float64).reshape(-1)
        scale = max(np.abs(b).max(), 1e-12)
        errs.append((np.abs(a - b).max() / scale, name))
    errs.sort(reverse=True)
    print("fused" if fused else "separate", "loss err %.2e" % abs(loss - float(ref)), ["%s %.2e" % (k, e) for e, k in errs[:6]])

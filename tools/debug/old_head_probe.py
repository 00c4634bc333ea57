"""Which shapes does the separate-launch head path (GCMI_FUSED_HEAD=0 / exact mode / GCMI_FUSED_BWD=0) get wrong?
Worst gradient tensor (|GPU - float64 oracle| / scale) for a grid of (tasks, molecules)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_gpu_scale import _native_step, _oracle_step
from oracle import graphconv_oracle as O
from deepchem_amd.utils.synthetic import synthetic_molecules

for tasks, n in ((12, 40), (12, 75), (20, 40), (20, 64), (20, 65), (20, 75), (20, 130), (100, 75), (16, 75)):
    packed = synthetic_molecules(n, seed=tasks + n, max_atoms=40)
    rng = np.random.RandomState(tasks)
    y = rng.randint(0, 2, size=(n, tasks)).astype(np.float64)
    w = (rng.rand(n, tasks) < 0.9).astype(np.float64) * (0.5 + rng.rand(n, tasks))
    cfg = O.ModelConfig(tasks, batch_size=n)
    state = O.init_state(cfg, 9)
    nat = _native_step(packed, y, w, tasks, "full", state)
    o64 = _oracle_step(packed, y, w, tasks, "full", state, double=True)
    loss, logits, fp, grads, slices, rng_, stats, _ = nat
    worst = []
    for name, (off, cnt) in slices:
        b = o64[2].get(name)
        if b is None:
            continue
        a = grads[off:off + cnt].numpy().astype(np.float64).reshape(-1)
        b = np.asarray(b, np.float64).reshape(-1)
        worst.append((np.abs(a - b).max() / max(np.abs(b).max(), 1e-6), name))
    worst.sort(reverse=True)
    print("tasks %3d mols %3d  loss err %.1e  worst: %s" % (tasks, n, abs(loss - o64[0]), ", ".join("%s %.1e" % (k, v) for v, k in worst[:3])))

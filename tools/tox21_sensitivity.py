"""How far do two CPU runs of the SAME Tox21 training drift apart when the initial weights differ by one
part in 10^7?  (Checker-side tool: runs the oracle, never the product.)

    python tools/tox21_sensitivity.py [b100|b64] [relative perturbation]

Trains the oracle (CPU restatement of the reference, reference gradient semantics) on the real Tox21 split
of tests/test_gpu_round2.py with the shuffles of np.random.seed(123), once from init_state(123) and once
from that state times (1 + eps * N(0,1)), and prints per-task valid ROC-AUC of both next to the fixture
the reference itself produced (tests/golden/tox21_ref.npz)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import graphconv_oracle as O  # noqa: E402
from oracle import mol_graphs_oracle as MO  # noqa: E402
from tests.test_gpu_round2 import tox21_splits  # noqa: E402
from tests.util import load_golden, oracle_batch, oracle_predict  # noqa: E402
from deepchem_amd.metrics import roc_auc_per_task  # noqa: E402


def train(run, eps, seed_pert=0):
    g = load_golden("tox21_ref.npz")
    B, epochs, seed = (int(v) for v in g[run + "_cfg"])
    train_ds, valid = tox21_splits()
    cfg = O.ModelConfig(12, batch_size=B)
    state = O.init_state(cfg, 123)
    if eps:
        rng = np.random.RandomState(seed_pert)
        state = {k: (v * (1.0 + eps * torch.from_numpy(rng.randn(*v.shape)).to(v.dtype))
                     if v.dtype.is_floating_point else v) for k, v in state.items()}
    tr = O.OracleTrainer(cfg, state, grad_mode="reference", learning_rate=float(g[run + "_lr"]), faithful=False)
    mols = [MO.conv_mol(*train_ds.packed.molecule(m)) for m in range(len(train_ds))]
    vmols = [MO.conv_mol(*valid.packed.molecule(m)) for m in range(len(valid))]
    np.random.seed(seed)
    losses = []
    t0 = time.time()
    for idx, n_real in train_ds.iter_index_batches(B, epochs, deterministic=False, pad_batches=True):
        inputs, labels, weights = oracle_batch(cfg, mols, train_ds.y, train_ds.w, idx, n_real, True)
        losses.append(tr.train_step(inputs, labels, weights))
    probs = oracle_predict(tr, cfg, vmols, 0)
    auc = roc_auc_per_task(valid.y, probs, valid.w)
    print("eps", eps, "wall %.0f s" % (time.time() - t0), "last-100 mean loss %.5f" % np.mean(losses[-100:]), flush=True)
    return probs, auc, g


if __name__ == "__main__":
    import json
    run = sys.argv[1] if len(sys.argv) > 1 else "b100"
    eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-7
    n_pert = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    torch.set_num_threads(os.cpu_count() or 1)
    p0, a0, g = train(run, 0.0)
    ref_p, ref_a = g[run + "_valid_probs"], g[run + "_valid_auc"]
    print("oracle vs reference fixture: max |dprob| %.3g  max |dAUC| %.4f" % (np.abs(p0 - ref_p).max(),
                                                                               np.nanmax(np.abs(a0 - ref_a))))
    task_dev, mean_dev, prob_dev = [], [], []
    for k in range(n_pert):
        p1, a1, _ = train(run, eps, seed_pert=k)
        task_dev.append(float(np.nanmax(np.abs(a1 - a0))))
        mean_dev.append(float(abs(np.nanmean(a1) - np.nanmean(a0))))
        prob_dev.append(float(np.abs(p1 - p0).max()))
        print("perturbed (%.0e, seed %d) vs unperturbed oracle: max |dprob| %.3g  max per-task |dAUC| %.4f  "
              "|d mean AUC| %.4f" % (eps, k, prob_dev[-1], task_dev[-1], mean_dev[-1]), flush=True)
    print("AUC reference ", np.round(ref_a, 4))
    print("AUC oracle    ", np.round(a0, 4))
    out = {"run": run, "eps": eps, "oracle_equals_reference_bitwise": bool(np.array_equal(p0, ref_p)),
           "max_per_task_dauc": task_dev, "d_mean_auc": mean_dev, "max_dprob": prob_dev}
    path = os.path.join(ROOT, "tests", "golden", "tox21_envelope_%s.json" % run)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)

"""MPNNModel (BASELINE.json config 4, QM9-like) on one GPU: training-step rate and the message kernel alone.
    python tools/kbench_mpnn.py [--mols 1024] [--steps 10] [--cpu-mols 16]
Synthetic molecules shaped like QM9 with hydrogens (18 atoms on average, at most 29; 70 atom features, 8 pair
features, all n x n ordered pairs), model n_hidden 100, T 5, M 10 (the reference's defaults,
deepchem/models/graph_models.py:1066-1076).  Reported: molecules/s of the optimizer step (forward, loss, backward,
Adam), the per-atom moment kernel of EdgeNetwork forward / backward with algorithmic GB/s (per launch: pair
features P*K*4 + gathered states P*d*4 read, moments N*(K+1)*d*4 written), and the oracle (torch-CPU autograd
restatement of the Keras model) on a smaller sample."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Mol:
    def __init__(self, nodes, pairs):
        self.nodes, self.pairs = nodes, pairs

    def get_num_atoms(self):
        return self.nodes.shape[0]

    def get_atom_features(self):
        return self.nodes

    def get_pair_features(self):
        return self.pairs


def qm9_like(n_mols, seed, fa=70, fp=8):
    rng = np.random.RandomState(seed)
    sizes = np.clip(np.round(rng.normal(18.0, 3.0, n_mols)), 3, 29).astype(int)
    mols = np.empty(n_mols, dtype=object)
    for i, n in enumerate(sizes):
        mols[i] = Mol((rng.rand(n, fa) < 0.1).astype(np.float32), (rng.rand(n, n, fp) < 0.25).astype(np.float32))
    return mols


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-mols", type=int, default=16)
    args = ap.parse_args()
    import deepchem_amd as dc
    from deepchem_amd import ops
    from deepchem_amd.models.torch_models.mpnn import MPNNModel, PairPlan
    dev = torch.device("cuda:0")
    B = args.mols
    model = MPNNModel(12, n_atom_feat=70, n_pair_feat=8, n_hidden=100, T=5, M=10, mode="regression", batch_size=B,
                      device=dev, learning_rate=1e-3, log_frequency=10**9)
    mols = qm9_like(B, 0)
    rng = np.random.RandomState(1)
    ds = dc.data.NumpyDataset(mols, rng.randn(B, 12), np.ones((B, 12)))
    batch = next(iter(model.default_generator(ds, pad_batches=True)))
    model._ensure_built()
    model.model.train()
    inputs, labels, weights = model._prepare_batch(batch)
    n_atoms, n_pairs = inputs[0].shape[0], inputs[1].shape[0]
    res = {"n_mols": B, "n_atoms": int(n_atoms), "n_pairs": int(n_pairs), "n_hidden": 100, "T": 5, "M": 10}

    def step():
        return model._train_step(inputs, labels, weights, model._loss_fn, model._pytorch_optimizer)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    res["train_step_ms"] = round(dt * 1e3, 3)
    res["train_molecules_per_s"] = round(B / dt, 1)
    with torch.no_grad():
        model.model.eval()
        model.model(inputs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.model(inputs)
        torch.cuda.synchronize()
        res["forward_ms"] = round((time.perf_counter() - t0) / args.steps * 1e3, 3)
    # the message kernel alone, both directions
    d, K = 100, 8
    pf = inputs[1]
    from deepchem_amd.models.torch_models.weave_layers import _csr_from_sorted
    split = inputs[2].cpu().numpy() if torch.is_tensor(inputs[2]) else np.asarray(inputs[2])
    mol_ptr = torch.from_numpy(_csr_from_sorted(np.asarray(split, np.int64), B, "atom_split")).to(dev)
    plan = PairPlan(inputs[3], pf, n_atoms, dev, mol_ptr, int(np.bincount(np.asarray(split, np.int64)).max()))
    h = torch.randn(n_atoms, d, device=dev) * 0.3
    alg = n_pairs * (K * 4 + d * 4) + n_atoms * (K + 1) * d * 4

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    us = timed(lambda: ops.edge_network_moments(h, plan.pf, plan.dst_ptr, plan.src, plan.mol_ptr, plan.max_mol_atoms))
    res["edge_moments_forward"] = {"us": round(us, 1), "algorithmic_GBps": round(alg / us / 1e3, 1)}
    us = timed(lambda: ops.edge_network_moments(h, plan.pf_t, plan.src_ptr, plan.dst_of_sorted, plan.mol_ptr, plan.max_mol_atoms))
    res["edge_moments_backward"] = {"us": round(us, 1), "algorithmic_GBps": round(alg / us / 1e3, 1)}
    us = timed(lambda: ops.edge_network_moments(h, plan.pf, plan.dst_ptr, plan.src))
    res["edge_moments_forward_per_atom_kernel"] = {"us": round(us, 1)}
    a_new = ops.edge_network_moments(h, plan.pf, plan.dst_ptr, plan.src, plan.mol_ptr, plan.max_mol_atoms)
    a_old = ops.edge_network_moments(h, plan.pf, plan.dst_ptr, plan.src)
    res["edge_moments_kernels_max_rel_diff"] = float((a_new - a_old).abs().max() / a_old.abs().max())
    res["reference_pair_matrix_GB_per_round"] = round(n_pairs * d * d * 4 / 1e9, 2)
    # CPU oracle on a smaller sample
    from oracle.mpnn_oracle import MPNNOracle
    cb = args.cpu_mols
    cm = qm9_like(cb, 2)
    cds = dc.data.NumpyDataset(cm, rng.randn(cb, 12), np.ones((cb, 12)))
    small = MPNNModel(12, n_atom_feat=70, n_pair_feat=8, n_hidden=100, T=5, M=10, mode="regression", batch_size=cb,
                      device=dev)
    cin, clab, cw = next(iter(small.default_generator(cds, pad_batches=True)))
    oracle = MPNNOracle({k: v.detach().cpu() for k, v in small.model.state_dict().items()}, 70, 100, 5, 10, cb,
                        "regression", 12)
    opt = torch.optim.Adam(list(oracle.p.values()), lr=1e-3)
    t0 = time.perf_counter()
    opt.zero_grad()
    l = oracle.loss(oracle.forward(*cin), clab[0], cw[0])
    l.backward()
    opt.step()
    ct = time.perf_counter() - t0
    res["cpu_oracle"] = {"mols": cb, "threads": torch.get_num_threads(), "train_step_s": round(ct, 3),
                         "molecules_per_s": round(cb / ct, 2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

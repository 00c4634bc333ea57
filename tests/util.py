"""Shared helpers for the tests: fixtures -> oracle runs.  (The oracle is the
checker; nothing here is product code.)"""
import os

import numpy as np
import torch

from deepchem_amd.utils.synthetic import PackedMols
from oracle import graphconv_oracle as O
from oracle import mol_graphs_oracle as MO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def packed_from(npz, prefix="in_"):
    return PackedMols(npz[prefix + "atom_features"], npz[prefix + "atom_ptr"],
                      npz[prefix + "adj_ptr"], npz[prefix + "adj_idx"])


def oracle_convmols(packed):
    return [MO.conv_mol(*packed.molecule(m)) for m in range(packed.n_mols)]


def batch_indices(n, batch_size, pad):
    """NumpyDataset.iterbatches(deterministic=True) + pad_batch (data/datasets.py:843-898, :142-218)."""
    for s in range(0, n, batch_size):
        idx = np.arange(s, min(n, s + batch_size))
        n_real = idx.shape[0]
        if pad and n_real < batch_size:
            idx = idx[np.arange(batch_size) % n_real]
        yield idx, n_real


def cfg_from(npz):
    return O.ModelConfig(int(npz["cfg_n_tasks"]), dense_layer_size=int(npz["cfg_dense"]),
                         mode="classification" if int(npz["cfg_mode"]) == 0 else "regression",
                         batch_normalize=bool(int(npz["cfg_batch_normalize"])),
                         uncertainty=bool(int(npz["cfg_uncertainty"])),
                         batch_size=int(npz["cfg_batch_size"]))


def oracle_batch(cfg, mols, y, w, idx, n_real, pad, predict=False):
    multi = MO.agglomerate([mols[i] for i in idx])
    y_b = None if y is None else y[idx]
    w_b = None
    if w is not None:
        w_b = w[idx].copy()
        if pad and n_real < idx.shape[0]:
            w_b[n_real:] = 0
    return O.batch_tensors(multi, idx.shape[0], y_b, w_b, cfg, predict=predict)


def oracle_fit(cfg, state, mols, y, w, epochs, grad_mode, faithful=True):
    tr = O.OracleTrainer(cfg, state, grad_mode=grad_mode, faithful=faithful)
    losses = []
    for _ in range(epochs):
        for idx, n_real in batch_indices(len(mols), cfg.batch_size, True):
            inputs, labels, weights = oracle_batch(cfg, mols, y, w, idx, n_real, True)
            losses.append(tr.train_step(inputs, labels, weights))
    return tr, losses


def oracle_predict(tr, cfg, mols, which):
    outs = []
    for idx, n_real in batch_indices(len(mols), cfg.batch_size, False):
        inputs, _, _ = oracle_batch(cfg, mols, None, None, idx, n_real, False, predict=True)
        outs.append(tr.predict(inputs)[which].numpy())
    return np.concatenate(outs, 0)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

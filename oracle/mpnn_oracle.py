"""CPU restatement of the reference's message-passing sub-layers (TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this).

* ``edge_network`` -- EdgeNetwork.forward, deepchem/models/torch_models/layers.py:4060-4088
* ``gru``          -- GatedRecurrentUnit.forward, layers.py:2903-2913
* ``set_gather``   -- SetGather.forward + _LSTMStep (set2set, M steps), layers.py:3024-3102; the
  ``torch_geometric.utils.scatter`` it calls (not installed here) is a per-molecule sum
* ``ccc_pair_features`` -- WeaveFeaturizer atom/pair features of 'CCC' derived by hand from
  feat/graph_features.py:531-651 (bond features :400-470, ``find_distance`` :654-694)

Pinned by tests/test_oracle_mpnn.py: the reference's assets ``edgenetwork_{weights,result}.npy``,
``{atom_feat,weights,result}_SetGather*.npy`` and outputs of the reference's EdgeNetwork /
GatedRecurrentUnit run in the build container (tests/golden/mpnn_layers.npz).  The torch
``MPNNModel`` itself delegates to dgllife (not under /root/reference, not installed): model-level
parity is unpinned.
"""
from typing import Tuple

import numpy as np
import torch

from oracle.weave_oracle import carbon_atom_features


def edge_network(pair_features, atom_features, atom_to_pair, W, b) -> torch.Tensor:
    pf = torch.as_tensor(pair_features).float()
    h = torch.as_tensor(atom_features).float()
    a2p = torch.as_tensor(np.asarray(atom_to_pair)).long()
    d = h.shape[1]
    A = (pf @ W + b).reshape(-1, d, d)
    out = torch.matmul(A, h[a2p[:, 1]].unsqueeze(2)).squeeze(2)
    ind = a2p[:, 0]
    n_seg = int(ind.max()) + 1 if ind.numel() else 0
    res = torch.zeros((n_seg, d))
    res.index_add_(0, ind, out)  # segment_sum over ascending ids
    return res


def gru(h_tm1, x, p) -> torch.Tensor:
    h_tm1, x = torch.as_tensor(h_tm1).float(), torch.as_tensor(x).float()
    z = torch.sigmoid(x @ p["Wz"] + h_tm1 @ p["Uz"] + p["bz"])
    r = torch.sigmoid(x @ p["Wr"] + h_tm1 @ p["Ur"] + p["br"])
    return (1 - z) * torch.tanh(x @ p["Wh"] + (h_tm1 * r) @ p["Uh"] + p["bh"]) + z * x


def lstm_step(h, c, U, b, n_hidden):
    z = h.float() @ U.float() + b
    i = torch.sigmoid(z[:, :n_hidden])
    f = torch.sigmoid(z[:, n_hidden:2 * n_hidden])
    o = torch.sigmoid(z[:, 2 * n_hidden:3 * n_hidden])
    c_out = f * c + i * torch.tanh(z[:, 3 * n_hidden:])
    return o * torch.tanh(c_out), c_out


def set_gather(atom_features, atom_split, M: int, batch_size: int, U, b) -> torch.Tensor:
    x = torch.as_tensor(np.asarray(atom_features))
    split = np.asarray(atom_split)
    n_hidden = x.shape[1]
    c = torch.zeros((batch_size, n_hidden))
    h = torch.zeros((batch_size, n_hidden))
    idx = torch.from_numpy(split).long()
    q_star = None
    for _ in range(M):
        e = (x * h[idx]).sum(dim=-1)
        a = torch.zeros_like(e)
        for m in range(batch_size):
            mask = torch.from_numpy(split == m)
            if mask.any():
                a[mask] = torch.softmax(e[mask], dim=0)
        r = torch.zeros((batch_size, n_hidden), dtype=x.dtype)
        r.index_add_(0, idx, a.reshape(-1, 1) * x)
        q_star = torch.cat([h, r], dim=1)
        h, c = lstm_step(q_star, c, U, b, n_hidden)
    return q_star


def ccc_pair_features(center_last: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(atom_features (3,75), pair_features (9,14) in meshgrid order, atom_to_pair (9,2)) of 'CCC' as
    the reference's EdgeNetwork test builds them (models/tests/test_layers.py:1144-1190)."""
    center = 2 if center_last else 1
    atoms = np.stack([carbon_atom_features(2, 2) if a == center else carbon_atom_features(1, 3) for a in range(3)])
    adj = {a: ([x for x in range(3) if x != center] if a == center else [center]) for a in range(3)}
    n = 3
    pairs = np.zeros((n, n, 14), np.float32)
    for a1 in range(n):
        for a2 in adj[a1]:
            pairs[a1, a2, :6] = [1, 0, 0, 0, 0, 0]  # single, not conjugated, not in a ring
        # graph-distance one-hot (find_distance): distance d -> column 7 + (d - 1); self: zeros
        for a2 in range(n):
            if a2 == a1:
                continue
            dist = 1 if a2 in adj[a1] else 2
            pairs[a1, a2, 7 + dist - 1] = 1
    C0, C1 = np.meshgrid(np.arange(n), np.arange(n))
    a2p = np.transpose(np.array([C1.flatten(), C0.flatten()]))
    return atoms, pairs.reshape(n * n, 14), a2p

"""CPU restatement of the reference's Weave layers (TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this).

Op for op on torch-CPU tensors, explicit weights instead of module state:

* ``weave_layer``  -- WeaveLayer.forward, deepchem/models/torch_models/layers.py:4327-4429
* ``weave_gather`` -- WeaveGather.forward + gaussian_histogram, layers.py:4566-4648
* ``weave_batch``  -- WeaveModel.compute_features_on_batch, torch_models/weavemodel_pytorch.py:516-578

Pinned by tests/test_oracle_weave.py against (a) the reference's own assets
``weavegather_results_{with,without}_compression.npy`` and (b) outputs of the reference layers run in
the build container on seeded inputs (oracle/gen_golden_weave.py -> tests/golden/weave_layers.npz).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

GAUSSIAN_MEMBERSHIPS = [(-1.645, 0.283), (-1.080, 0.170), (-0.739, 0.134), (-0.468, 0.118), (-0.228, 0.114),
                        (0., 0.114), (0.228, 0.114), (0.468, 0.118), (0.739, 0.134), (1.080, 0.170), (1.645, 0.283)]


def bn_eval(x: torch.Tensor, bn: Optional[Dict[str, torch.Tensor]], eps: float = 1e-3) -> torch.Tensor:
    """nn.BatchNorm1d in eval mode (the reference calls ``.eval()`` before every use)."""
    if bn is None:
        return x
    return (x - bn["running_mean"]) / torch.sqrt(bn["running_var"] + eps) * bn["weight"] + bn["bias"]


def segment_sum_in_order(x: torch.Tensor, ids: Sequence[int]) -> torch.Tensor:
    """The reference's dict loop (layers.py:4374-4386, :4585-4596): rows with equal id are added up,
    result rows in order of first appearance."""
    groups: Dict[int, torch.Tensor] = {}
    for r, s in enumerate(ids):
        s = int(s)
        groups[s] = groups[s] + x[r] if s in groups else x[r]
    return torch.stack(list(groups.values()))


def weave_layer(atom_features, pair_features, pair_split, atom_to_pair, p: Dict[str, torch.Tensor],
                bns: Optional[Dict[str, Dict[str, torch.Tensor]]], update_pair: bool = True):
    """p: W_AA,b_AA,W_PA,b_PA,W_A,b_A[,W_AP,b_AP,W_PP,b_PP,W_P,b_P]; bns: AA,PA,A[,AP,PP,P] or None."""
    A = torch.as_tensor(atom_features).float()
    Pf = torch.as_tensor(pair_features).float()
    a2p = torch.as_tensor(np.asarray(atom_to_pair)).long()
    bn = (lambda x, k: bn_eval(x, bns[k])) if bns is not None else (lambda x, k: x)
    AA = torch.relu(bn(A @ p["W_AA"] + p["b_AA"], "AA"))
    PA = torch.relu(bn(Pf @ p["W_PA"] + p["b_PA"], "PA"))
    PA = segment_sum_in_order(PA, np.asarray(pair_split))
    A_out = torch.relu(bn(torch.cat([AA, PA], 1) @ p["W_A"] + p["b_A"], "A"))
    if not update_pair:
        return A_out, Pf
    n_in = A.shape[1]
    AP_ij = torch.relu(bn(A[a2p].reshape(-1, 2 * n_in) @ p["W_AP"] + p["b_AP"], "AP"))
    AP_ji = torch.relu(bn(A[torch.flip(a2p, [1])].reshape(-1, 2 * n_in) @ p["W_AP"] + p["b_AP"], "AP"))
    PP = torch.relu(bn(Pf @ p["W_PP"] + p["b_PP"], "PP"))
    P_out = torch.relu(bn(torch.cat([AP_ij + AP_ji, PP], 1) @ p["W_P"] + p["b_P"], "P"))
    return A_out, P_out


def gaussian_histogram(x: torch.Tensor) -> torch.Tensor:
    import torch.distributions as dist
    ds = [dist.Normal(torch.tensor(m), torch.tensor(s)) for m, s in GAUSSIAN_MEMBERSHIPS]
    peak = [ds[i].log_prob(torch.tensor(GAUSSIAN_MEMBERSHIPS[i][0])).exp() for i in range(11)]
    out = torch.stack([ds[i].log_prob(x).exp() / peak[i] for i in range(11)], dim=2)
    out = out / torch.sum(out, dim=2, keepdim=True)
    return out.reshape(-1, x.shape[1] * 11)


def weave_gather(atom_features, atom_split, gaussian_expand: bool = True, W: Optional[torch.Tensor] = None,
                 b: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = torch.as_tensor(atom_features).float()
    if gaussian_expand:
        x = gaussian_histogram(x)
    out = segment_sum_in_order(x, np.asarray(atom_split))
    if W is not None:
        out = torch.tanh(out @ W + (b if b is not None else 0.))
    return out


def weave_batch(mols: List[Tuple[np.ndarray, np.ndarray, np.ndarray]]):
    """mols: (nodes (n,Fa), pairs (n_pairs,Fp), pair_edges (2,n_pairs)) per molecule ->
    (atom_feat, pair_feat, pair_split, atom_split, atom_to_pair)."""
    atom_feat, pair_feat, atom_split, atom_to_pair, pair_split = [], [], [], [], []
    start = 0
    for im, (nodes, pairs, pair_edges) in enumerate(mols):
        n_atoms = nodes.shape[0]
        atom_split.extend([im] * n_atoms)
        atom_to_pair.append(pair_edges.T + start)
        pair_split.extend(pair_edges.T[:, 0] + start)
        start += n_atoms
        atom_feat.append(nodes)
        pair_feat.append(pairs)
    return (np.concatenate(atom_feat, axis=0), np.concatenate(pair_feat, axis=0), np.array(pair_split),
            np.array(atom_split), np.concatenate(atom_to_pair, axis=0))


def carbon_atom_features(degree: int, n_h: int) -> np.ndarray:
    """The 75-vector of an sp3 carbon (feat/graph_features.py:322-381: 44 symbol | 11 degree |
    7 implicit valence | charge | radicals | 5 hybridisation | aromatic | 5 total H)."""
    v = np.zeros(75, np.float32)
    v[[0, 44 + degree, 55 + n_h, 66, 70 + n_h]] = 1.0
    return v


def ccc_and_c_atoms() -> Tuple[np.ndarray, np.ndarray]:
    """Atom features / atom_split of the reference's ['CCC', 'C'] WeaveGather tests
    (models/tests/test_weave_gather.py:14-93); the order of the atoms inside a molecule is irrelevant
    to a per-molecule sum."""
    feats = np.stack([carbon_atom_features(1, 3), carbon_atom_features(1, 3), carbon_atom_features(2, 2),
                      carbon_atom_features(0, 4)])
    return feats, np.array([0, 0, 0, 1])

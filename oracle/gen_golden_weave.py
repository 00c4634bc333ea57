"""Generate tests/golden/weave_layers.npz by running THE REFERENCE's torch Weave layers in the build
container (run here once, commit the output; /root/reference does not travel):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_weave.py

* the reference's own assets for WeaveGather (models/tests/assets/weavegather_*.npy), as plain arrays;
* WeaveLayer / WeaveGather (models/torch_models/layers.py:4135-4648) on seeded random batches with
  seeded weights and NON-trivial BatchNorm running statistics / affine parameters, with and without
  pair update, batch normalisation, Gaussian expansion and compression;
* ``WeaveModel.compute_features_on_batch`` (torch_models/weavemodel_pytorch.py:516-578) on the same
  molecules.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import OUT, REF, import_reference  # noqa: E402


def random_mols(seed, n_mols=7, max_atoms=9, fa=75, fp=14, max_dist_pairs=False):
    """(nodes, pairs, pair_edges) per molecule: self pairs always, every other ordered pair either
    always (complete) or with probability 0.5 symmetric (a finite max_pair_distance)."""
    rng = np.random.RandomState(seed)
    mols = []
    for m in range(n_mols):
        n = 1 if m == 2 else rng.randint(2, max_atoms + 1)
        keep = np.ones((n, n), bool)
        if max_dist_pairs:
            up = rng.rand(n, n) < 0.5
            keep = np.triu(up, 1)
            keep = keep | keep.T | np.eye(n, dtype=bool)
        src, dst = np.nonzero(keep)            # row-major: grouped by source atom, ascending
        edges = np.stack([src, dst]).astype(np.int64)
        nodes = rng.standard_normal((n, fa)).astype(np.float32)
        pairs = rng.standard_normal((edges.shape[1], fp)).astype(np.float32)
        mols.append((nodes, pairs, edges))
    return mols


def main():
    dc = import_reference()
    import torch
    import deepchem.models.torch_models.layers as L
    from deepchem.feat.mol_graphs import WeaveMol
    from deepchem.models.torch_models import WeaveModel
    out = {}
    A = os.path.join(REF, "deepchem/models/tests/assets")
    for name in ("weavegather_results_with_compression", "weavegather_results_without_compression",
                 "weavegather_weights"):
        out["asset_" + name] = np.load(os.path.join(A, name + ".npy"))

    wm = WeaveModel(1, batch_size=4, mode="classification", fully_connected_layer_sizes=[20, 10])
    for case, (seed, partial) in enumerate(((0, False), (1, True))):
        mols = random_mols(seed, max_dist_pairs=partial)
        X = [WeaveMol(n, p, e) for n, p, e in mols]
        atom_feat, pair_feat, pair_split, atom_split, atom_to_pair = wm.compute_features_on_batch(X)
        pre = "c%d_" % case
        out[pre + "n_mols"] = np.array(len(mols))
        for i, (n, p, e) in enumerate(mols):
            out[pre + "mol%d_nodes" % i], out[pre + "mol%d_pairs" % i], out[pre + "mol%d_edges" % i] = n, p, e
        out[pre + "atom_feat"], out[pre + "pair_feat"] = atom_feat, pair_feat
        out[pre + "pair_split"], out[pre + "atom_split"], out[pre + "atom_to_pair"] = pair_split, atom_split, atom_to_pair
        for bn_on in (True, False):
            for update_pair in (True, False):
                torch.manual_seed(100 + case)
                layer = L.WeaveLayer(n_atom_output_feat=40, n_pair_output_feat=30, n_hidden_AA=50, n_hidden_PA=34,
                                     n_hidden_AP=26, n_hidden_PP=50, update_pair=update_pair, batch_normalize=bn_on)
                rng = np.random.RandomState(5 + case)
                tag = pre + "bn%d_up%d_" % (bn_on, update_pair)
                for name in ("W_AA", "b_AA", "W_PA", "b_PA", "W_A", "b_A", "W_AP", "b_AP", "W_PP", "b_PP", "W_P", "b_P"):
                    if not hasattr(layer, name):
                        continue
                    t = getattr(layer, name)
                    v = (rng.standard_normal(tuple(t.shape)) * (0.2 if name.startswith("W") else 0.5)).astype(np.float32)
                    setattr(layer, name, torch.from_numpy(v))
                    out[tag + name] = v
                for name in ("AA_bn", "PA_bn", "A_bn", "AP_bn", "PP_bn", "P_bn"):
                    if not hasattr(layer, name):
                        continue
                    bn = getattr(layer, name)
                    n = bn.num_features
                    vals = {"running_mean": rng.standard_normal(n) * 0.3, "running_var": rng.rand(n) + 0.3,
                            "weight": rng.rand(n) + 0.5, "bias": rng.standard_normal(n) * 0.2}
                    with torch.no_grad():
                        bn.running_mean.copy_(torch.from_numpy(vals["running_mean"].astype(np.float32)))
                        bn.running_var.copy_(torch.from_numpy(vals["running_var"].astype(np.float32)))
                        bn.weight.copy_(torch.from_numpy(vals["weight"].astype(np.float32)))
                        bn.bias.copy_(torch.from_numpy(vals["bias"].astype(np.float32)))
                    for k, v in vals.items():
                        out[tag + name + "_" + k] = v.astype(np.float32)
                with torch.no_grad():
                    Ao, Po = layer([atom_feat, pair_feat, pair_split, atom_to_pair])
                out[tag + "A_out"] = Ao.numpy()
                out[tag + "P_out"] = Po.numpy() if torch.is_tensor(Po) else np.asarray(Po)
        # gather on (scaled) random atom rows
        rng = np.random.RandomState(9 + case)
        x = (rng.standard_normal((atom_feat.shape[0], 24)) * 0.8).astype(np.float32)
        out[pre + "gather_x"] = x
        for expand in (True, False):
            g = L.WeaveGather(batch_size=len(mols), n_input=24, gaussian_expand=expand)
            with torch.no_grad():
                out[pre + "gather_e%d" % expand] = g([x, atom_split]).numpy()
        g = L.WeaveGather(batch_size=len(mols), n_input=24, gaussian_expand=True, compress_post_gaussian_expansion=True)
        W = (rng.standard_normal((24 * 11, 24)) * 0.1).astype(np.float32)
        b = (rng.standard_normal(24) * 0.3).astype(np.float32)
        g.W, g.b = torch.from_numpy(W), torch.from_numpy(b)
        out[pre + "gather_W"], out[pre + "gather_b"] = W, b
        with torch.no_grad():
            out[pre + "gather_compressed"] = g([x, atom_split]).numpy()
            out[pre + "gather_hist"] = g.gaussian_histogram(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(OUT, "weave_layers.npz"), **out)
    print("wrote weave_layers.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()

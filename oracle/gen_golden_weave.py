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


if __name__ == "__main__" and "--model" not in sys.argv:
    main()


def model_fixture():
    """tests/golden/weave_model.npz: the reference WeaveModel (default seed-22 initialisation) on a
    seeded set of WeaveMol objects: initial parameters, predictions, per-batch training losses of
    ``fit_on_batch`` and the parameters afterwards, classification and regression."""
    dc = import_reference()
    import torch
    from deepchem.feat.mol_graphs import WeaveMol
    from deepchem.models.torch_models import WeaveModel
    out = {}
    mols = random_mols(11, n_mols=10, max_atoms=7)
    rng = np.random.RandomState(4)
    mols = [((n * 0.5).astype(np.float32), (p * 0.5).astype(np.float32), e) for n, p, e in mols]
    out["n_mols"] = np.array(len(mols))
    for i, (n, p, e) in enumerate(mols):
        out["mol%d_nodes" % i], out["mol%d_pairs" % i], out["mol%d_edges" % i] = n, p, e
    X = np.empty(len(mols), dtype=object)
    for i, (n, p, e) in enumerate(mols):
        X[i] = WeaveMol(n, p, e)
    for mode in ("classification", "regression"):
        y = (rng.rand(len(mols), 2) < 0.5).astype(np.float64) if mode == "classification" else rng.standard_normal((len(mols), 2))
        w = (rng.rand(len(mols), 2) < 0.9).astype(np.float64)
        out[mode + "_y"], out[mode + "_w"] = y, w
        model = WeaveModel(2, fully_connected_layer_sizes=[40, 20], batch_size=4, mode=mode, learning_rate=1e-3,
                           device=torch.device("cpu"))
        ds = dc.data.NumpyDataset(X, y, w)
        sd = {k: v.detach().cpu().numpy().copy() for k, v in model.model.state_dict().items()}
        for k, v in sd.items():
            out[mode + "_init_" + k] = v
        for li, layer in enumerate(model.model.layers):
            for name in ("W_AA", "W_PA", "W_A", "W_AP", "W_PP", "W_P"):
                if hasattr(layer, name):
                    out[mode + "_init_layers.%d.%s" % (li, name)] = getattr(layer, name).detach().numpy().copy()
        out[mode + "_pred0"] = model.predict(ds)
        losses = []
        for epoch in range(3):
            for s in range(0, len(mols), 4):
                losses.append(model.fit_on_batch(X[s:s + 4], y[s:s + 4], w[s:s + 4]))
        out[mode + "_losses"] = np.array(losses, np.float64)
        out[mode + "_pred1"] = model.predict(ds)
        for k, v in model.model.state_dict().items():
            out[mode + "_trained_" + k] = v.detach().cpu().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "weave_model.npz"), **out)
    print("wrote weave_model.npz with", len(out), "arrays")


if __name__ == "__main__" and "--model" in sys.argv:
    model_fixture()

"""Generate tests/golden/mpnn_layers.npz by running THE REFERENCE's EdgeNetwork and GatedRecurrentUnit
(models/torch_models/layers.py:4006-4088, :2884-2913) in the build container on seeded inputs, and copy
the reference's assets for EdgeNetwork / SetGather as plain arrays.  (SetGather itself imports
torch_geometric, which is not installed: it is pinned by its assets only.)

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_mpnn.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import OUT, REF, import_reference  # noqa: E402
from oracle.gen_golden_weave import random_mols  # noqa: E402
from oracle.weave_oracle import weave_batch  # noqa: E402


def main():
    import_reference()
    import torch
    import deepchem.models.torch_models.layers as L
    out = {}
    A = os.path.join(REF, "deepchem/models/tests/assets")
    for name in ("edgenetwork_weights", "edgenetwork_result", "atom_feat_SetGather", "weights_SetGather_tf",
                 "result_SetGather_tf"):
        out["asset_" + name] = np.load(os.path.join(A, name + ".npy"))
    for case, (d, K) in enumerate(((20, 14), (75, 8))):
        mols = random_mols(20 + case, n_mols=6, max_atoms=8, fa=d, fp=K)
        atom_feat, pair_feat, pair_split, atom_split, atom_to_pair = weave_batch(mols)
        rng = np.random.RandomState(case)
        W = (rng.standard_normal((K, d * d)) * 0.1).astype(np.float32)
        b = (rng.standard_normal(d * d) * 0.1).astype(np.float32)
        layer = L.EdgeNetwork(K, d)
        layer.W, layer.b = torch.from_numpy(W), torch.from_numpy(b)
        pre = "c%d_" % case
        out[pre + "atom_feat"], out[pre + "pair_feat"], out[pre + "atom_to_pair"] = atom_feat, pair_feat, atom_to_pair
        out[pre + "atom_split"] = atom_split
        out[pre + "W"], out[pre + "b"] = W, b
        with torch.no_grad():
            msg = layer([torch.from_numpy(pair_feat), torch.from_numpy(atom_feat), torch.from_numpy(atom_to_pair)])
        out[pre + "edge_out"] = msg.numpy()
        gru = L.GatedRecurrentUnit(d)
        for name in ("Wz", "Wr", "Wh", "Uz", "Ur", "Uh", "bz", "br", "bh"):
            v = (rng.standard_normal(tuple(getattr(gru, name).shape)) * 0.3).astype(np.float32)
            setattr(gru, name, torch.from_numpy(v))
            out[pre + "gru_" + name] = v
        with torch.no_grad():
            out[pre + "gru_out"] = gru([torch.from_numpy(atom_feat), msg]).numpy()
    np.savez_compressed(os.path.join(OUT, "mpnn_layers.npz"), **out)
    print("wrote mpnn_layers.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()

"""cProfile of one fit() call (one epoch of 65 536-molecule batches) after a warm-up call: where the host time of a
call goes outside the steps."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepchem_amd as dc  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=262144)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--sort", default="cumulative")
    args = ap.parse_args()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "tests", "golden", "smiles_sample.txt")) as f:
        smiles = [l.strip() for l in f if l.strip() and not l.startswith("#")]
    base, _ = dc.feat.ConvMolFeaturizer().featurize_packed(smiles)
    n = args.mols
    packed = base.select(np.arange(n) % base.n_mols)
    y = (np.random.RandomState(0).rand(n, 12) < 0.1).astype(np.float64)
    w = np.ones_like(y)
    model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=args.batch,
                                                  grad_mode="full", log_frequency=10**9)
    ds = dc.data.PackedDataset(packed, y, w)
    model.fit(ds, nb_epoch=1, checkpoint_interval=0)
    model.fit(ds, nb_epoch=1, checkpoint_interval=0)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    model.fit(ds, nb_epoch=1, checkpoint_interval=0)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats(args.sort).print_stats(32)


if __name__ == "__main__":
    main()

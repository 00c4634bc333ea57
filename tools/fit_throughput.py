"""End-to-end fit() throughput (collation + H2D + step) on featurized real molecules: the committed 400-SMILES
sample tiled to N molecules, kept as 8-byte atom codes or as float rows."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepchem_amd as dc  # noqa: E402
from deepchem_amd.utils.synthetic import PackedMols  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=262144)
    ap.add_argument("--batches", default="4096,65536")
    ap.add_argument("--epochs", type=int, default=2, help="timed epochs (the pipeline's fill is part of the time)")
    args = ap.parse_args()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "tests", "golden", "smiles_sample.txt")) as f:
        smiles = [l.strip() for l in f if l.strip() and not l.startswith("#")]
    t0 = time.perf_counter()
    base, _ = dc.feat.ConvMolFeaturizer().featurize_packed(smiles)
    t_feat = time.perf_counter() - t0
    idx = np.arange(args.mols) % base.n_mols
    coded = base.select(idx)
    flt = PackedMols(coded.atom_features, coded.atom_ptr, coded.adj_ptr, coded.adj_idx)
    rng = np.random.RandomState(0)
    y = (rng.rand(args.mols, 12) < 0.1).astype(np.float64)
    w = np.ones_like(y)
    res = {"epochs": args.epochs, "resident_set": os.environ.get("GCMI_RESIDENT_SET", "1") != "0", "n_mols": args.mols, "n_atoms": coded.n_atoms, "featurize_400_smiles_ms": round(t_feat * 1e3, 2)}
    for B in [int(b) for b in args.batches.split(",")]:
        for name, packed in (("codes", coded), ("floats", flt)):
            model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=B,
                                                          grad_mode="full", log_frequency=10**9)
            ds = dc.data.PackedDataset(packed, y, w)
            model.fit(ds, nb_epoch=1, checkpoint_interval=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.fit(ds, nb_epoch=args.epochs, checkpoint_interval=0)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["fit_molecules_per_s_batch_%d_%s" % (B, name)] = round(args.epochs * args.mols / dt, 1)
            print(B, name, res["fit_molecules_per_s_batch_%d_%s" % (B, name)], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

"""Rate of the batch pipeline alone (no training step): shuffled batches of the tiled 400-SMILES sample, collated
on the host or by the GPU from the resident set, handed to a consumer that only waits for them."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepchem_amd as dc  # noqa: E402
from deepchem_amd.data.packed_dataset import DeviceBatchPipeline, PackedDataset  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=262144)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--epochs", type=int, default=8)
    args = ap.parse_args()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "tests", "golden", "smiles_sample.txt")) as f:
        smiles = [l.strip() for l in f if l.strip() and not l.startswith("#")]
    base, _ = dc.feat.ConvMolFeaturizer().featurize_packed(smiles)
    packed = base.select(np.arange(args.mols) % base.n_mols)
    y = (np.random.RandomState(0).rand(args.mols, 12) < 0.1).astype(np.float64)
    w = np.ones_like(y)
    dev = torch.device("cuda:0")
    res = {"n_mols": args.mols, "batch": args.batch, "epochs": args.epochs}
    for resident in (False, True):
        for workers in (1, 2, 3):
            helper = PackedDataset(packed, y, w)
            rates = []
            for rep in range(2):
                idx = helper.iter_index_batches(args.batch, args.epochs, False, True)
                pipe = DeviceBatchPipeline(packed, y, w, idx, dev, resident=resident, workers=workers)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 0
                for batch, y_t, w_t in pipe:
                    n += batch.n_samples
                torch.cuda.synchronize()
                rates.append(n / (time.perf_counter() - t0))
            key = "%s_workers_%d" % ("resident" if resident else "host", workers)
            res[key] = round(rates[-1], 1)
            print(key, res[key], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

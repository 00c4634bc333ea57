"""One configuration of the small-batch engine for rocprofv3 passes:
    python tools/small_profile.py [batch] [grad_mode] [epochs] [engine 0/1]
fit() on the real Tox21 train split (tests/golden/tox21.csv.gz) + one predict() of the valid split; prints rates."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import deepchem_amd as dc
    from deepchem_amd.data.data_loader import convert_df_to_numpy, load_csv_files
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    gm = sys.argv[2] if len(sys.argv) > 2 else "reference"
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    engine = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    df = next(iter(load_csv_files([os.path.join(ROOT, "tests", "golden", "tox21.csv.gz")], shard_size=8192)))
    packed, keep = dc.feat.ConvMolFeaturizer().featurize_packed(df["smiles"].tolist())
    y, w = convert_df_to_numpy(df, bench.TOX21_TASKS)
    y, w = y[keep], w[keep]
    n = packed.n_mols
    a, b = int(0.8 * n), int(0.9 * n)
    train = dc.data.PackedDataset(packed.select(np.arange(a)), y[:a], w[:a])
    valid = dc.data.PackedDataset(packed.select(np.arange(a, b)), y[a:b], w[a:b])
    model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=B, grad_mode=gm,
                                                  device=dev, log_frequency=10**9)
    model.small_batch_engine = engine
    model.fit(train, nb_epoch=1, checkpoint_interval=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit(train, nb_epoch=epochs, checkpoint_interval=0)
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    model.predict(valid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.predict(valid)
    pred_s = time.perf_counter() - t0
    steps = epochs * ((a + B - 1) // B)
    print(json.dumps({"batch": B, "grad_mode": gm, "engine": engine, "fit_molecules_per_s": round(epochs * a / fit_s, 1),
                      "us_per_step": round(fit_s / steps * 1e6, 2), "predict_molecules_per_s": round(len(valid) / pred_s, 1)}))


if __name__ == "__main__":
    main()

"""Host and device time of one small-batch optimizer step, without a profiler attached.
    python tools/small_steptime.py [batch] [grad_mode]
One epoch of the real Tox21 train split is collated once into a chunk; ``engine.fit`` then runs it several
times.  Reported per step: host time of the C call (enqueue of all launches; perf_counter around the call) and
device time (HIP events on the stream around the same call; equals the kernels' time when the host enqueues
faster than the GPU executes, else the host time)."""
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
    from deepchem_amd.small import ChunkCollator
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    gm = sys.argv[2] if len(sys.argv) > 2 else "reference"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    df = next(iter(load_csv_files([os.path.join(ROOT, "tests", "golden", "tox21.csv.gz")], shard_size=8192)))
    packed, keep = dc.feat.ConvMolFeaturizer().featurize_packed(df["smiles"].tolist())
    y, w = convert_df_to_numpy(df, bench.TOX21_TASKS)
    y, w = y[keep], w[keep]
    a = int(0.8 * packed.n_mols)
    packed = packed.select(np.arange(a))
    model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=B, grad_mode=gm,
                                                  device=dev, log_frequency=10**9)
    model._ensure_built()
    model.model.train()
    engine = model._small_engine()
    y_dev, stride, w_dev = model._labels_for_small(packed, y[:a], w[:a], True)
    coll = ChunkCollator(packed, dev, B)
    n_b = a // B
    idx = [np.arange(i * B, (i + 1) * B) for i in range(n_b)]
    t0 = time.perf_counter()
    ch = coll.collate(idx, [B] * n_b)
    torch.cuda.synchronize()
    collate_s = time.perf_counter() - t0
    sel = torch.from_numpy(np.concatenate(idx)).to(dev)
    coll.bind(ch, [B] * n_b, labels=y_dev.index_select(0, sel), label_stride=stride, weights=w_dev.index_select(0, sel),
              weight_stride=12)
    engine.fit(ch.descs, model._pytorch_optimizer, ch.max_atoms, B)
    torch.cuda.synchronize()
    host, devt = [], []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        engine.fit(ch.descs, model._pytorch_optimizer, ch.max_atoms, B)
        host.append(time.perf_counter() - t0)
        e1.record()
        torch.cuda.synchronize()
        devt.append(e0.elapsed_time(e1) * 1e-3)
    print(json.dumps({"batch": B, "grad_mode": gm, "steps_per_call": n_b, "diag": os.environ.get("GCMI_SMALL_DIAG", "0"),
                      "host_us_per_step": round(min(host) / n_b * 1e6, 2), "device_us_per_step": round(min(devt) / n_b * 1e6, 2),
                      "collate_us_per_batch": round(collate_s / n_b * 1e6, 2),
                      "molecules_per_s_device": round(B * n_b / min(devt), 1)}))


if __name__ == "__main__":
    main()

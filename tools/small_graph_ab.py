"""Eager launches against a captured hipGraph for ONE small-batch optimizer step (SURVEY.md 7 called graph capture
"mandatory" for small batches; this is the measurement).
    python tools/small_graph_ab.py [batch] [grad_mode]
The same collated batch of real Tox21 molecules is stepped 200 times (a) by gcmi_small_fit's own loop, 200 steps
per C call, (b) by 200 C calls of one step each, (c) by 200 replays of a hipGraph captured from one such call.
Reported per step: host time to enqueue and device time (HIP events)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    t0 = time.perf_counter()
    fn()
    host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    return round(host / n * 1e6, 2), round(e0.elapsed_time(e1) * 1e3 / n, 2)


def main():
    import deepchem_amd as dc
    from deepchem_amd.data.data_loader import convert_df_to_numpy, load_csv_files
    from deepchem_amd.small import ChunkCollator
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    gm = sys.argv[2] if len(sys.argv) > 2 else "reference"
    n = 200
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    df = next(iter(load_csv_files([os.path.join(ROOT, "tests", "golden", "tox21.csv.gz")], shard_size=8192)))
    packed, keep = dc.feat.ConvMolFeaturizer().featurize_packed(df["smiles"].tolist())
    y, w = convert_df_to_numpy(df, bench.TOX21_TASKS)
    y, w = y[keep], w[keep]
    model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=B, grad_mode=gm,
                                                  device=dev, log_frequency=10**9)
    model._ensure_built()
    model.model.train()
    engine = model._small_engine()
    y_dev, stride, w_dev = model._labels_for_small(packed, y, w, True)
    coll = ChunkCollator(packed, dev, B)
    idx = [np.arange(B)] * n
    ch = coll.collate(idx, [B] * n)
    sel = torch.from_numpy(np.concatenate(idx)).to(dev)
    coll.bind(ch, [B] * n, labels=y_dev.index_select(0, sel), label_stride=stride, weights=w_dev.index_select(0, sel),
              weight_stride=12)
    opt = model._pytorch_optimizer
    one = type(ch.descs)._type_ * 1
    first = one(ch.descs[0])
    engine.fit(ch.descs, opt, ch.max_atoms, B)
    out = {"batch": B, "grad_mode": gm, "steps": n, "atoms_per_batch": ch.max_atoms}
    out["loop_in_library_host_us"], out["loop_in_library_device_us"] = timed(
        lambda: engine.fit(ch.descs, opt, ch.max_atoms, B), n)

    def eager_calls():
        for _ in range(n):
            engine.fit(first, opt, ch.max_atoms, B)
    eager_calls()
    out["one_call_per_step_host_us"], out["one_call_per_step_device_us"] = timed(eager_calls, n)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        engine.fit(first, opt, ch.max_atoms, B)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            engine.fit(first, opt, ch.max_atoms, B)
    torch.cuda.synchronize()

    def replays():
        for _ in range(n):
            graph.replay()
    replays()
    out["hipgraph_replay_host_us"], out["hipgraph_replay_device_us"] = timed(replays, n)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Where does the small-batch engine stop winning?  One optimizer step on Tox21-like synthetic batches of growing
size, engine (gcmi_small_fit) against the streaming kernels (gcmi_model_forward / _loss_backward + Adam), device
time per step by HIP events.   python tools/small_crossover.py [grad_mode]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import deepchem_amd as dc
    from deepchem_amd.data.collate import collate_to_device
    from deepchem_amd.metrics import to_one_hot
    from deepchem_amd.small import SmallBatchEngine
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    gm = sys.argv[1] if len(sys.argv) > 1 else "full"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for B in (256, 1024, 4096, 16384, 65536):
        packed = synthetic_molecules(B, seed=B)
        y, w = synthetic_labels(B, 12, "classification", seed=B)
        batch = collate_to_device(packed, None, dev)
        batch.graph.ensure_rev_pos()
        labels = torch.as_tensor(to_one_hot(y.flatten(), 2).reshape(-1, 12, 2).astype(np.float32), device=dev)
        weights = torch.as_tensor(w.astype(np.float32), device=dev)
        rec = {"batch": B, "atoms": batch.n_atoms, "grad_mode": gm}
        for which in ("engine", "streaming"):
            model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=B, grad_mode=gm,
                                                          device=dev, log_frequency=10**9)
            model._ensure_built()
            model.model.train()
            n = 20 if B <= 4096 else 6
            if which == "engine":
                eng = SmallBatchEngine(model.model._native_net())
                descs = [eng.describe(batch, labels, weights, B)] * n
                run = lambda: eng.fit(descs, model._pytorch_optimizer, batch.n_atoms, B)
            else:
                def run():
                    for _ in range(n):
                        model._train_step(batch, [labels], [weights], model._loss_fn, model._pytorch_optimizer)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            rec[which + "_us_per_step"] = round(us, 1)
            rec[which + "_molecules_per_s"] = round(B / us * 1e6, 0)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

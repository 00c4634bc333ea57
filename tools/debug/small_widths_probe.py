"""Why do the losses of tests/test_gpu_small.py::test_small_engine_other_widths drift apart after the first step?  Prints the
engine's losses beside the per-batch path's in both product modes: if fast-vs-exact of the SAME path differs as much as
engine-vs-exact, the drift is Adam's sign(noise) on rounding-level gradients, not the engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import deepchem_amd as dc
from tests.test_gpu_small import _setup, _device_batches, DEV
from deepchem_amd.small import SmallBatchEngine

packed, y, w, cfg, state, model = _setup("regression", 1, 12, 44, True, "full", seed=11, widths=(128, 128), dense=256)
model._ensure_built(); model.model.train()
engine = SmallBatchEngine(model.model._native_net())
batches = _device_batches(model, packed, y, w, cfg, 12)
descs = [engine.describe(b, l, ww, 12) for b, l, ww, _, _ in batches]
losses = engine.fit(descs, model._pytorch_optimizer, max(b.n_atoms for b, *_ in batches), 12).cpu().tolist()
print("engine      ", losses)
for mode in ("exact", "fast", "exact"):
    dc.set_gemm_mode(mode)
    m = dc.models.torch_models.GraphConvModel(1, number_input_features=[75, 128], graph_conv_layers=[128, 128],
                                              dense_layer_size=256, mode="regression", batch_size=12,
                                              grad_mode="full", device=torch.device(DEV), learning_rate=1e-3)
    m.model.load_state_dict({k: v.clone() for k, v in state.items()})
    m._ensure_built(); m.model.train()
    ref = [float(m._train_step(b, [l], [ww], m._loss_fn, m._pytorch_optimizer)) for b, l, ww, _, _ in batches]
    print("per-batch %-5s" % mode, ref)
dc.set_gemm_mode("fast")

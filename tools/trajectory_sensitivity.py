"""How far do two fp32-accurate runs of the SAME training drift apart?  Fits the cls_bn fixture model
(tests/golden/model_cls_bn.npz, 2 epochs of Adam) and prints the distance of its predictions to the
reference's, for the current GEMM kernels and for inputs perturbed in the last bit (development
tool; needs a GPU)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import build_model, dataset_from  # noqa: E402
from tests.util import load_golden  # noqa: E402


def run(perturb):
    g = load_golden("model_cls_bn.npz")
    model, cfg, state = build_model(g, "full")
    ds, _ = dataset_from(g)
    if perturb:
        for m in ds.X:
            f = m.get_atom_features()
            f *= np.float32(1.0 + perturb)
    losses = []
    model.fit(ds, nb_epoch=2, deterministic=True, checkpoint_interval=0,
              callbacks=[lambda m, s, iteration_loss=None: losses.append(float(iteration_loss))])
    pred = model.predict(ds)
    return np.array(losses), pred, g


base_l, base_p, g = run(0.0)
print("kernels: GEMM_V4=%s" % os.environ.get("GCMI_GEMM_V4", "1"))
print("vs reference: max |pred diff| %.4g, losses rel %.3g" % (np.abs(base_p - g["full_predict"]).max(),
      np.abs(base_l / g["full_fit_losses"] - 1).max()))
for eps in (1e-7, 1e-6):
    l, p, _ = run(eps)
    print("inputs * (1+%g): max |pred diff| to unperturbed run %.4g, losses rel %.3g" % (
        eps, np.abs(p - base_p).max(), np.abs(l / base_l - 1).max()))

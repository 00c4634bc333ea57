"""cProfile of fit() at the reference's default batch size (development tool)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepchem_amd as dc  # noqa: E402
from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules  # noqa: E402

p = synthetic_molecules(8192, seed=5)
y, w = synthetic_labels(8192, 12, "classification", 5)
ds = dc.data.PackedDataset(p, y, w)
m = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=100,
                                          grad_mode="full", log_frequency=10**9)
m.fit(ds, nb_epoch=1, checkpoint_interval=0)
pr = cProfile.Profile()
pr.enable()
m.fit(ds, nb_epoch=1, checkpoint_interval=0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)

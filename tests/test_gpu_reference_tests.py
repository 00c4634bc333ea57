"""The reference's own model-level tests for GraphConvModel (deepchem/models/tests/test_graph_models.py:50-142), run
against the drop-in classes: same constructor arguments, epochs and bars.  Their data (20 molecules of the BACE /
Delaney sets through the rdkit featurizer) is replaced by 20 molecules of the committed Delaney sample through the
native featurizer; labels are random, as there.  Default gradient mode (``reference``): what a DeepChem user gets."""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def get_dataset(mode, num_tasks=2, n=20, seed=0):
    import deepchem_amd as dc
    import pandas as pd
    smiles = pd.read_csv(os.path.join(GOLDEN, "delaney_sample.csv"))["smiles"].tolist()[:n]
    X = dc.feat.ConvMolFeaturizer().featurize(smiles)
    rng = np.random.RandomState(seed)
    y = rng.randint(0, 2, size=(n, num_tasks)).astype(float) if mode == "classification" else rng.normal(size=(n, num_tasks))
    return dc.data.NumpyDataset(X, y, np.ones((n, num_tasks)), np.array(smiles, dtype=object))


def seeded(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)


def test_graph_conv_model():
    """:50-63 (@flaky there): batch 10, no BatchNorm, 20 epochs, mean ROC-AUC >= 0.9 on the training set."""
    from deepchem_amd.metrics import roc_auc_per_task
    from deepchem_amd.models.torch_models import GraphConvModel
    ds = get_dataset("classification")
    seen = []
    for seed in range(3):
        seeded(seed)
        model = GraphConvModel(2, number_input_features=[75, 64], batch_size=10, batch_normalize=False,
                               mode='classification', device=DEV)
        model.fit(ds, nb_epoch=20)
        scores = model.evaluate(ds, [lambda y, p, w: roc_auc_per_task(y, p, w)], [])
        seen.append(list(scores.values())[0])
        if seen[-1] >= 0.9:
            break
    assert max(seen) >= 0.9, seen


def test_neural_fingerprint_retrieval():
    """:66-81: dense_layer_size 3, batch 50, one epoch; the embedding is (n, 2 * 3) after the caller's trim."""
    from deepchem_amd.models.torch_models import GraphConvModel
    ds = get_dataset("classification")
    model = GraphConvModel(2, number_input_features=[75, 64], batch_size=50, dense_layer_size=3, mode='classification',
                           device=DEV)
    model.fit(ds, nb_epoch=1)
    fp = np.array(model.predict_embedding(ds))[:len(ds)]
    assert fp.shape == (len(ds), 6)


def test_graph_conv_regression_model():
    """:84-98 (@flaky there): batch 10, no BatchNorm, 100 epochs, mean absolute error < 0.1."""
    from deepchem_amd.models.torch_models import GraphConvModel
    ds = get_dataset("regression")
    seen = []
    for seed in range(3):
        seeded(seed)
        model = GraphConvModel(2, number_input_features=[75, 64], batch_size=10, batch_normalize=False, mode='regression',
                               device=DEV)
        model.fit(ds, nb_epoch=100)
        seen.append(float(np.abs(model.predict(ds) - ds.y).mean()))
        if seen[-1] < 0.1:
            break
    assert min(seen) < 0.1, seen


def test_graph_conv_regression_uncertainty():
    """:101-122: dropout 0.1 + uncertainty head, 100 epochs; the error / predicted-deviation relations."""
    from deepchem_amd.models.torch_models import GraphConvModel
    ds = get_dataset("regression")
    seen = []
    for seed in range(3):
        seeded(seed)
        model = GraphConvModel(2, number_input_features=[75, 64], batch_size=10, batch_normalize=False, mode='regression',
                               dropout=0.1, uncertainty=True, device=DEV)
        model.fit(ds, nb_epoch=100)
        pred, std = model.predict_uncertainty(ds, masks=5)
        mean_error, mean_value, mean_std = np.mean(np.abs(ds.y - pred)), np.mean(np.abs(ds.y)), np.mean(std)
        seen.append((float(mean_error), float(mean_std), float(mean_value)))
        if mean_error < 0.5 * mean_value and mean_std > 0.5 * mean_error and mean_std < mean_value:
            break
    else:
        raise AssertionError(seen)


def test_graph_conv_model_no_task(tmp_path):
    """:125-141: predict() on a CSV featurized with tasks=[] (no labels at all)."""
    import deepchem_amd as dc
    from deepchem_amd.models.torch_models import GraphConvModel
    ds = get_dataset("classification")
    model = GraphConvModel(2, number_input_features=[75, 64], batch_size=10, batch_normalize=False, mode='classification',
                           device=DEV)
    model.fit(ds, nb_epoch=2)
    loader = dc.data.CSVLoader(tasks=[], feature_field="smiles", featurizer=dc.feat.ConvMolFeaturizer())
    td = loader.create_dataset(os.path.join(GOLDEN, "delaney_sample.csv"), data_dir=str(tmp_path))
    pred = model.predict(td)
    assert pred.shape == (len(td), 2, 2) and np.isfinite(pred).all()

"""GPU parity of the Weave layers (deepchem_amd.models.torch_models.layers.WeaveLayer / WeaveGather,
kernels in csrc/weave.hip + the segmented GEMM) against outputs of the reference layers
(tests/golden/weave_layers.npz), the reference's own WeaveGather assets and the oracle.
fp32 tolerance 1e-4 relative (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import weave_oracle as WO
from tests.test_oracle_weave import layer_params, rel
from tests.util import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def G():
    return load_golden("weave_layers.npz")


def build_layer(G, tag, bn_on, update_pair):
    from deepchem_amd.models.torch_models.layers import WeaveLayer
    layer = WeaveLayer(n_atom_output_feat=40, n_pair_output_feat=30, n_hidden_AA=50, n_hidden_PA=34, n_hidden_AP=26,
                       n_hidden_PP=50, update_pair=update_pair, batch_normalize=bn_on)
    p, bns = layer_params(G, tag)
    for k, v in p.items():
        assert tuple(getattr(layer, k).shape) == tuple(v.shape), k
        setattr(layer, k, v)  # plain CPU tensors, assigned the way the reference's tests do
    for name, vals in bns.items():
        bn = getattr(layer, name + "_bn")
        with torch.no_grad():
            bn.running_mean.copy_(vals["running_mean"])
            bn.running_var.copy_(vals["running_var"])
            bn.weight.copy_(vals["weight"])
            bn.bias.copy_(vals["bias"])
    return layer


@pytest.mark.parametrize("case", [0, 1])
@pytest.mark.parametrize("bn_on", [True, False])
@pytest.mark.parametrize("update_pair", [True, False])
def test_weave_layer_matches_reference(G, case, bn_on, update_pair):
    pre = "c%d_" % case
    tag = pre + "bn%d_up%d_" % (bn_on, update_pair)
    layer = build_layer(G, tag, bn_on, update_pair)
    A, P = layer([G[pre + "atom_feat"], G[pre + "pair_feat"], G[pre + "pair_split"], G[pre + "atom_to_pair"]])
    assert A.is_cuda and P.is_cuda
    assert rel(A.cpu().numpy(), G[tag + "A_out"]) < TOL
    assert rel(P.cpu().numpy(), G[tag + "P_out"]) < TOL
    # stacked layers take the previous layer's GPU outputs
    if update_pair:
        layer2 = build_layer(G, tag, bn_on, True)
        layer2.n_atom_input_feat = 40
        with pytest.raises(Exception):
            layer2([A, P, G[pre + "pair_split"], G[pre + "atom_to_pair"]])  # widths do not match W_AA


@pytest.mark.parametrize("case", [0, 1])
def test_weave_gather_matches_reference(G, case):
    from deepchem_amd.models.torch_models.layers import WeaveGather
    pre = "c%d_" % case
    x, split = G[pre + "gather_x"], G[pre + "atom_split"]
    n_mols = int(G[pre + "n_mols"])
    g = WeaveGather(batch_size=n_mols, n_input=24, gaussian_expand=True)
    assert rel(g([x, split]).cpu().numpy(), G[pre + "gather_e1"]) < TOL
    assert rel(g.gaussian_histogram(x).cpu().numpy(), G[pre + "gather_hist"]) < TOL
    g = WeaveGather(batch_size=n_mols, n_input=24, gaussian_expand=False)
    assert rel(g([torch.from_numpy(x).cuda(), split]).cpu().numpy(), G[pre + "gather_e0"]) < TOL
    g = WeaveGather(batch_size=n_mols, n_input=24, gaussian_expand=True, compress_post_gaussian_expansion=True)
    g.W, g.b = torch.from_numpy(G[pre + "gather_W"]), torch.from_numpy(G[pre + "gather_b"])
    assert rel(g([x, split]).cpu().numpy(), G[pre + "gather_compressed"]) < TOL


def test_weave_gather_reference_assets(G):
    """The reference's own known answers for ['CCC', 'C'] (models/tests/test_weave_gather.py)."""
    from deepchem_amd.models.torch_models.layers import WeaveGather
    feats, split = WO.ccc_and_c_atoms()
    g = WeaveGather(batch_size=2, n_input=75, gaussian_expand=True)
    out = g([feats, split])
    assert tuple(out.shape) == (2, 11 * 75)
    assert np.allclose(out.cpu().numpy(), G["asset_weavegather_results_without_compression"], atol=1e-4)
    g = WeaveGather(batch_size=2, n_input=75, gaussian_expand=True, compress_post_gaussian_expansion=True)
    g.W = torch.from_numpy(G["asset_weavegather_weights"])
    out = g([feats, split])
    assert tuple(out.shape) == (2, 75)
    assert np.allclose(out.cpu().numpy(), G["asset_weavegather_results_with_compression"], atol=1e-4)


def test_large_batch_against_oracle():
    """A batch of ~3k atoms / ~60k pairs: the fused kernels against the oracle's op-by-op graph."""
    from deepchem_amd.models.torch_models.layers import WeaveGather, WeaveLayer
    from oracle.gen_golden_weave import random_mols
    mols = random_mols(3, n_mols=120, max_atoms=40)
    atom_feat, pair_feat, pair_split, atom_split, atom_to_pair = WO.weave_batch(mols)
    layer = WeaveLayer()
    p = {k: getattr(layer, k).cpu() for k in ("W_AA", "b_AA", "W_PA", "b_PA", "W_A", "b_A", "W_AP", "b_AP", "W_PP",
                                              "b_PP", "W_P", "b_P")}
    bns = {n: {"running_mean": torch.zeros(getattr(layer, n + "_bn").num_features),
               "running_var": torch.ones(getattr(layer, n + "_bn").num_features),
               "weight": torch.ones(getattr(layer, n + "_bn").num_features),
               "bias": torch.zeros(getattr(layer, n + "_bn").num_features)} for n in ("AA", "PA", "A", "AP", "PP", "P")}
    A, P = layer([atom_feat, pair_feat, pair_split, atom_to_pair])
    Ar, Pr = WO.weave_layer(atom_feat, pair_feat, pair_split, atom_to_pair, p, bns, True)
    assert rel(A.cpu().numpy(), Ar.numpy()) < TOL and rel(P.cpu().numpy(), Pr.numpy()) < TOL
    g = WeaveGather(batch_size=len(mols), n_input=50)
    # inside the model the gather sees tanh outputs; far outside the bins all eleven Gaussians
    # underflow and the reference's normalisation yields 0/0 = NaN: the kernel reproduces that too
    small = (Ar * 0.1).numpy()
    assert rel(g([small, atom_split]).cpu().numpy(), WO.weave_gather(small, atom_split, True).numpy()) < TOL
    # (exactly where the underflow sets in depends on denormal handling of expf, so the two NaN
    # sets are compared in size, the values where both are finite)
    ours, ref = g([A, atom_split]).cpu().numpy(), WO.weave_gather(Ar, atom_split, True).numpy()
    assert np.isnan(ref).any() and abs(np.isnan(ours).mean() - np.isnan(ref).mean()) < 0.02
    ok = ~np.isnan(ref) & ~np.isnan(ours)
    assert np.abs(ours[ok] - ref[ok]).max() <= TOL * np.abs(ref[ok]).max()


def test_weave_layers_refuse_bad_input():
    from deepchem_amd.models.torch_models.layers import WeaveLayer
    layer = WeaveLayer(n_atom_input_feat=4, n_pair_input_feat=3)
    A = np.zeros((3, 4), np.float32)
    Pf = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError):
        layer([A, Pf, np.array([0, 2, 1]), np.array([[0, 0], [2, 2], [1, 1]])])  # pair_split not ascending
    with pytest.raises(ValueError):
        layer([A, Pf, np.array([0, 1, 1]), np.array([[0, 0], [1, 1], [1, 5]])])  # atom index out of range


@pytest.mark.parametrize("mode", ["classification", "regression"])
def test_weave_model_matches_reference(mode):
    """WeaveModel under the TorchModel loop against the reference model (tests/golden/weave_model.npz):
    identical seed-22 initialisation, predictions, per-batch losses of fit_on_batch over 3 epochs,
    predictions and trainable parameters afterwards."""
    import deepchem_amd as dc
    from deepchem_amd.models.torch_models import WeaveModel, WeaveMol
    M = load_golden("weave_model.npz")
    n = int(M["n_mols"])
    X = np.empty(n, dtype=object)
    for i in range(n):
        X[i] = WeaveMol(M["mol%d_nodes" % i], M["mol%d_pairs" % i], M["mol%d_edges" % i])
    y, w = M[mode + "_y"], M[mode + "_w"]
    model = WeaveModel(2, fully_connected_layer_sizes=[40, 20], batch_size=4, mode=mode, learning_rate=1e-3)
    sd = model.model.state_dict()
    init = {}
    for k, v in sd.items():  # same construction order, same torch RNG stream (trunc_normal_'s erfinv may
        init[k] = v.detach().cpu().numpy().copy()  # differ in the last bit between host CPUs)
        assert np.allclose(init[k], M[mode + "_init_" + k], rtol=0, atol=1e-7), k
    for li, layer in enumerate(model.model.layers):
        for name in ("W_AA", "W_PA", "W_A", "W_AP", "W_PP", "W_P"):
            if hasattr(layer, name):
                assert np.allclose(getattr(layer, name).cpu().numpy(), M[mode + "_init_layers.%d.%s" % (li, name)],
                                   rtol=0, atol=1e-7)
    ds = dc.data.NumpyDataset(X, y, w)
    pred0 = model.predict(ds)
    assert pred0.shape == M[mode + "_pred0"].shape
    assert rel(pred0, M[mode + "_pred0"]) < TOL
    losses = []
    for epoch in range(3):
        for s in range(0, n, 4):
            losses.append(model.fit_on_batch(X[s:s + 4], y[s:s + 4], w[s:s + 4]))
    assert np.allclose(losses, M[mode + "_losses"], rtol=2e-4, atol=1e-6), (losses, M[mode + "_losses"])
    assert rel(model.predict(ds), M[mode + "_pred1"]) < 2e-3
    for k, v in model.model.state_dict().items():
        e = M[mode + "_trained_" + k]
        if k.startswith(("layers2", "layer_2")) and v.dtype.is_floating_point:
            assert np.abs(v.cpu().numpy() - e).max() <= 1e-2 * max(np.abs(e).max(), 1e-3), k
        elif v.dtype.is_floating_point:
            assert np.array_equal(v.cpu().numpy(), init[k]), k  # nothing in front of the gather trains
            assert np.allclose(v.cpu().numpy(), e, rtol=0, atol=1e-7), k


def test_weave_model_on_real_smiles_from_the_native_featurizer():
    """SMILES -> WeaveFeaturizer (native reader) -> WeaveModel.fit / predict on 64 rows of the Delaney file:
    the pieces of rows f-2 and f-4 together.  Only the post-gather stack trains in this model (reference
    quirk), so the bar is a falling loss and finite predictions of the right shape."""
    import os
    import pandas as pd
    import deepchem_amd as dc
    from deepchem_amd.models.torch_models import WeaveModel
    here = os.path.dirname(os.path.abspath(__file__))
    df = pd.read_csv(os.path.join(here, "golden", "delaney_sample.csv")).iloc[:64]
    X = dc.feat.WeaveFeaturizer().featurize(df["smiles"].tolist())
    assert all(m.get_num_features() == 75 and m.get_pair_features().shape[1] == 14 for m in X)
    y = df["measured log solubility in mols per litre"].to_numpy().reshape(-1, 1)
    y = (y - y.mean()) / y.std()
    ds = dc.data.NumpyDataset(X, y, np.ones_like(y))
    model = WeaveModel(1, batch_size=16, mode="regression", learning_rate=3e-3)
    first = model.fit(ds, nb_epoch=1)
    last = model.fit(ds, nb_epoch=30)
    pred = model.predict(ds)
    assert pred.shape[0] == 64 and np.isfinite(pred).all()
    assert last < 0.8 * first, (first, last)

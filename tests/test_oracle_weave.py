"""The Weave oracle (oracle/weave_oracle.py) against the reference: its own WeaveGather assets and
the outputs of the reference layers on seeded batches (tests/golden/weave_layers.npz, written by
oracle/gen_golden_weave.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import weave_oracle as WO
from tests.util import load_golden

TOL = 2e-5


@pytest.fixture(scope="module")
def G():
    return load_golden("weave_layers.npz")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def layer_params(G, tag):
    p = {k: torch.from_numpy(G[tag + k]) for k in ("W_AA", "b_AA", "W_PA", "b_PA", "W_A", "b_A", "W_AP", "b_AP", "W_PP",
                                                   "b_PP", "W_P", "b_P") if tag + k in G.files}
    bns = {}
    for name in ("AA", "PA", "A", "AP", "PP", "P"):
        if tag + name + "_bn_running_mean" in G.files:
            bns[name] = {k: torch.from_numpy(G[tag + name + "_bn_" + k]) for k in ("running_mean", "running_var",
                                                                                 "weight", "bias")}
    return p, bns


def test_reference_assets_for_ccc_and_c(G):
    """models/tests/test_weave_gather.py: ['CCC', 'C'], n_input 75, Gaussian expansion."""
    feats, split = WO.ccc_and_c_atoms()
    out = WO.weave_gather(feats, split, True)
    assert out.shape == (2, 11 * 75)
    assert np.allclose(out.numpy(), G["asset_weavegather_results_without_compression"], atol=1e-4)
    out = WO.weave_gather(feats, split, True, torch.from_numpy(G["asset_weavegather_weights"]), torch.zeros(75))
    assert np.allclose(out.numpy(), G["asset_weavegather_results_with_compression"], atol=1e-4)


@pytest.mark.parametrize("case", [0, 1])
def test_batch_construction(G, case):
    pre = "c%d_" % case
    mols = [(G[pre + "mol%d_nodes" % i], G[pre + "mol%d_pairs" % i], G[pre + "mol%d_edges" % i])
            for i in range(int(G[pre + "n_mols"]))]
    got = WO.weave_batch(mols)
    for arr, name in zip(got, ("atom_feat", "pair_feat", "pair_split", "atom_split", "atom_to_pair")):
        assert np.array_equal(arr, G[pre + name]), name


@pytest.mark.parametrize("case", [0, 1])
@pytest.mark.parametrize("bn_on", [True, False])
@pytest.mark.parametrize("update_pair", [True, False])
def test_weave_layer(G, case, bn_on, update_pair):
    pre = "c%d_" % case
    tag = pre + "bn%d_up%d_" % (bn_on, update_pair)
    p, bns = layer_params(G, tag)
    A, P = WO.weave_layer(G[pre + "atom_feat"], G[pre + "pair_feat"], G[pre + "pair_split"], G[pre + "atom_to_pair"], p,
                          bns if bn_on else None, update_pair)
    assert rel(A.numpy(), G[tag + "A_out"]) < TOL
    assert rel(P.numpy(), G[tag + "P_out"]) < TOL


@pytest.mark.parametrize("case", [0, 1])
def test_weave_gather(G, case):
    pre = "c%d_" % case
    x, split = G[pre + "gather_x"], G[pre + "atom_split"]
    assert rel(WO.weave_gather(x, split, True).numpy(), G[pre + "gather_e1"]) < TOL
    assert rel(WO.weave_gather(x, split, False).numpy(), G[pre + "gather_e0"]) < TOL
    assert rel(WO.gaussian_histogram(torch.from_numpy(x)).numpy(), G[pre + "gather_hist"]) < TOL
    out = WO.weave_gather(x, split, True, torch.from_numpy(G[pre + "gather_W"]), torch.from_numpy(G[pre + "gather_b"]))
    assert rel(out.numpy(), G[pre + "gather_compressed"]) < TOL

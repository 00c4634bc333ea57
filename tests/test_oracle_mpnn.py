"""The message-passing oracle (oracle/mpnn_oracle.py) against the reference: its EdgeNetwork and
SetGather assets and outputs of the reference's EdgeNetwork / GatedRecurrentUnit on seeded batches
(tests/golden/mpnn_layers.npz, oracle/gen_golden_mpnn.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import mpnn_oracle as MO
from tests.util import load_golden

TOL = 2e-5


@pytest.fixture(scope="module")
def G():
    return load_golden("mpnn_layers.npz")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def gru_params(G, pre):
    return {k: torch.from_numpy(G[pre + "gru_" + k]) for k in ("Wz", "Wr", "Wh", "Uz", "Ur", "Uh", "bz", "br", "bh")}


def test_edge_network_reference_asset(G):
    """models/tests/test_layers.py:1144-1208 ('CCC', WeaveFeaturizer features derived by hand)."""
    atoms, pairs, a2p = MO.ccc_pair_features()
    out = MO.edge_network(pairs, atoms, a2p, torch.from_numpy(G["asset_edgenetwork_weights"]), torch.zeros(75 * 75))
    assert np.allclose(out.numpy(), G["asset_edgenetwork_result"], atol=1e-4)


def test_set_gather_reference_asset(G):
    """models/tests/test_layers.py:996-1016."""
    b = torch.cat((torch.zeros(4), torch.ones(4), torch.zeros(4), torch.zeros(4)))
    out = MO.set_gather(G["asset_atom_feat_SetGather"], np.array([0, 0, 1, 1]), 2, 2,
                        torch.from_numpy(G["asset_weights_SetGather_tf"]), b)
    assert np.allclose(out.numpy(), G["asset_result_SetGather_tf"], atol=1e-4)


@pytest.mark.parametrize("case", [0, 1])
def test_edge_network_and_gru(G, case):
    pre = "c%d_" % case
    msg = MO.edge_network(G[pre + "pair_feat"], G[pre + "atom_feat"], G[pre + "atom_to_pair"], torch.from_numpy(G[pre + "W"]),
                          torch.from_numpy(G[pre + "b"]))
    assert rel(msg.numpy(), G[pre + "edge_out"]) < TOL
    assert rel(MO.gru(G[pre + "atom_feat"], G[pre + "edge_out"], gru_params(G, pre)).numpy(), G[pre + "gru_out"]) < TOL

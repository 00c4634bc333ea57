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


def test_message_passing_round_follows_the_keras_gru():
    """One round of MessagePassing by hand (models/layers.py:3692-3709, EdgeNetwork :3740-3752, GatedRecurrentUnit.call
    :3786-3799): the update carries z * inputs[0] = z * (previous state) -- NOT z * message, which is what the torch
    port of the stand-alone layer does (torch_models/layers.py:2912).  Two atoms, two ordered pairs, n_hidden 2."""
    d = 2
    rng = np.random.RandomState(3)
    P = {k: rng.uniform(-0.8, 0.8, (d, d)) for k in ("gru_Wz", "gru_Wr", "gru_Wh", "gru_Uz", "gru_Ur", "gru_Uh")}
    P.update({k: rng.uniform(-0.3, 0.3, (d,)) for k in ("gru_bz", "gru_br", "gru_bh")})
    P["edge_W"] = rng.uniform(-0.5, 0.5, (3, d * d))
    P["edge_b"] = rng.uniform(-0.2, 0.2, (d * d,))
    x = np.array([[0.5, -1.0], [2.0, 0.25]])
    pf = np.array([[1.0, 0.0, 0.5], [0.0, 1.0, -0.5]])
    a2p = np.array([[0, 1], [1, 0]])  # (destination, source): pair 0 brings atom 1 to atom 0, pair 1 the reverse
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    msg = np.zeros((2, d))
    for k in range(2):
        A = (pf[k] @ P["edge_W"] + P["edge_b"]).reshape(d, d)
        msg[a2p[k, 0]] += A @ x[a2p[k, 1]]
    z = sig(msg @ P["gru_Wz"] + x @ P["gru_Uz"] + P["gru_bz"])
    r = sig(msg @ P["gru_Wr"] + x @ P["gru_Ur"] + P["gru_br"])
    want = (1 - z) * np.tanh(msg @ P["gru_Wh"] + (x * r) @ P["gru_Uh"] + P["gru_bh"]) + z * x
    wrong = (1 - z) * np.tanh(msg @ P["gru_Wh"] + (x * r) @ P["gru_Uh"] + P["gru_bh"]) + z * msg
    assert np.abs(want - wrong).max() > 0.05  # the two definitions are far apart on this input
    o = MO.MPNNOracle({k: torch.from_numpy(v).float() for k, v in P.items()}, d, d, 1, 1, 2)
    got = o.message_passing(x, pf, a2p).detach().numpy()
    assert np.abs(got - want).max() < 1e-6

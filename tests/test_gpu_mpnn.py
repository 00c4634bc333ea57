"""GPU parity of the message-passing sub-layers (EdgeNetwork, GatedRecurrentUnit, SetGather; kernels in
csrc/mpnn.hip + the segmented GEMM) against the reference's assets, outputs of the reference layers
(tests/golden/mpnn_layers.npz) and the oracle.  fp32 tolerance 1e-4 relative."""
import numpy as np
import pytest
import torch

from oracle import mpnn_oracle as MO
from tests.test_oracle_mpnn import gru_params, rel
from tests.util import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def G():
    return load_golden("mpnn_layers.npz")


def test_edge_network_reference_asset(G):
    from deepchem_amd.models.torch_models.layers import EdgeNetwork
    atoms, pairs, a2p = MO.ccc_pair_features()
    layer = EdgeNetwork(14, 75)
    layer.W = torch.from_numpy(G["asset_edgenetwork_weights"])
    out = layer([torch.Tensor(pairs), torch.Tensor(atoms), torch.from_numpy(a2p)])
    assert tuple(out.shape) == (3, 75)
    assert np.allclose(out.cpu().numpy(), G["asset_edgenetwork_result"], atol=1e-4)


def test_set_gather_reference_asset(G):
    from deepchem_amd.models.torch_models.layers import SetGather
    layer = SetGather(2, 2, 4)
    layer.U = torch.nn.Parameter(torch.from_numpy(G["asset_weights_SetGather_tf"]))
    out = layer([G["asset_atom_feat_SetGather"], np.array([0, 0, 1, 1], dtype=np.int32)])
    assert tuple(out.shape) == (2, 8)
    assert np.allclose(out.cpu().numpy(), G["asset_result_SetGather_tf"], atol=1e-4)


@pytest.mark.parametrize("case", [0, 1])
def test_edge_network_and_gru_match_reference(G, case):
    from deepchem_amd.models.torch_models.layers import EdgeNetwork, GatedRecurrentUnit
    pre = "c%d_" % case
    d, K = G[pre + "atom_feat"].shape[1], G[pre + "pair_feat"].shape[1]
    layer = EdgeNetwork(K, d)
    layer.W, layer.b = torch.from_numpy(G[pre + "W"]), torch.from_numpy(G[pre + "b"])
    msg = layer([G[pre + "pair_feat"], G[pre + "atom_feat"], G[pre + "atom_to_pair"]])
    assert rel(msg.cpu().numpy(), G[pre + "edge_out"]) < TOL
    gru = GatedRecurrentUnit(d)
    for k, v in gru_params(G, pre).items():
        setattr(gru, k, v)
    h = gru([G[pre + "atom_feat"], msg])
    assert rel(h.cpu().numpy(), G[pre + "gru_out"]) < TOL


def test_message_passing_steps_and_set2set_against_oracle(G):
    """T = 3 rounds of EdgeNetwork + GRU, then M = 4 rounds of set2set on a larger batch."""
    from deepchem_amd.models.torch_models.layers import EdgeNetwork, GatedRecurrentUnit, SetGather
    from oracle.gen_golden_weave import random_mols
    from oracle.weave_oracle import weave_batch
    d, K = 32, 14
    mols = random_mols(5, n_mols=40, max_atoms=25, fa=d, fp=K)
    atom_feat, pair_feat, _, atom_split, a2p = weave_batch(mols)
    atom_feat = atom_feat * 0.3
    edge, gru, gather = EdgeNetwork(K, d), GatedRecurrentUnit(d), SetGather(4, len(mols), d)
    p = {k: getattr(gru, k).cpu() for k in ("Wz", "Wr", "Wh", "Uz", "Ur", "Uh", "bz", "br", "bh")}
    h_gpu, h_ref = torch.from_numpy(atom_feat).cuda(), torch.from_numpy(atom_feat)
    for _ in range(3):
        h_gpu = gru([h_gpu, edge([pair_feat, h_gpu, a2p])])
        h_ref = MO.gru(h_ref, MO.edge_network(pair_feat, h_ref, a2p, edge.W.cpu(), edge.b.cpu()), p)
    assert rel(h_gpu.cpu().numpy(), h_ref.numpy()) < TOL
    q = gather([h_gpu, atom_split])
    q_ref = MO.set_gather(h_ref.numpy(), atom_split, 4, len(mols), gather.U.detach(), gather.b.detach())
    assert rel(q.cpu().numpy(), q_ref.numpy()) < TOL


def test_edge_moments_molecule_kernel_equals_the_per_atom_kernel(monkeypatch):
    """gcmi_edge_network_moments_mol (a molecule's state rows staged in LDS) against gcmi_edge_network_moments on the
    same pair lists: ragged molecules, one larger than the LDS budget (rows from memory), pairs that leave their
    molecule (must still be right), both pair-feature widths' instantiations."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from deepchem_amd import ops
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
for K, d in ((8, 100), (14, 64), (3, 128)):
    sizes = [1, 5, 29, 2, 17, 150 if d == 100 else 40, 9]
    n = sum(sizes)
    mol_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    dst, src = [], []
    for m, s in enumerate(sizes):
        a0 = mol_ptr[m]
        for i in range(s):
            for j in range(s):
                if rng.rand() < 0.8:
                    dst.append(a0 + i); src.append(a0 + j)
    # a few pairs that leave their molecule
    for _ in range(20):
        dst.append(int(rng.randint(n))); src.append(int(rng.randint(n)))
    order = np.argsort(np.asarray(dst), kind="stable")
    dst, src = np.asarray(dst)[order], np.asarray(src)[order]
    dst_ptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).astype(np.int32)
    h = torch.randn(n, d, device=dev)
    pf = torch.rand(len(dst), K, device=dev)
    a = ops.edge_network_moments(h, pf, torch.from_numpy(dst_ptr).to(dev), torch.from_numpy(src.astype(np.int32)).to(dev))
    for biggest in (0, max(sizes)):
        b = ops.edge_network_moments(h, pf, torch.from_numpy(dst_ptr).to(dev), torch.from_numpy(src.astype(np.int32)).to(dev),
                                     torch.from_numpy(mol_ptr).to(dev), biggest)
        err = float((a - b).abs().max() / a.abs().max())
        assert err <= 1e-6, (K, d, biggest, err)
print("ok")
'''
    env = dict(os.environ, GCMI_EDGE_MOMENTS_MOL="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]

"""MPNNModel (BASELINE.json config 4) on the GPU against ``oracle.mpnn_oracle.MPNNOracle`` -- a torch-CPU autograd
restatement of the Keras MPNNModel (deepchem/models/graph_models.py:1045-1247, models/layers.py:3648-3887).
MODEL-LEVEL PARITY IS UNPINNED: the reference's Keras model needs TensorFlow and its torch MPNNModel is dgllife's
(neither is available); the sub-layers composed here are pinned by the reference's own assets in
tests/test_gpu_mpnn.py / tests/test_oracle_mpnn.py.  Tolerance 1e-4 relative (north_star)."""
import numpy as np
import pytest
import torch

from oracle.mpnn_oracle import MPNNOracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Mol:
    """The three accessors MPNNModel.default_generator uses (a WeaveMol with all n x n pairs)."""

    def __init__(self, nodes, pairs):
        self.nodes, self.pairs = nodes, pairs

    def get_num_atoms(self):
        return self.nodes.shape[0]

    def get_atom_features(self):
        return self.nodes

    def get_pair_features(self):
        return self.pairs


def qm9_like(n_mols, n_atom_feat, n_pair_feat, seed, max_atoms=12):
    rng = np.random.RandomState(seed)
    mols = np.empty(n_mols, dtype=object)
    for i in range(n_mols):
        n = int(rng.randint(1, max_atoms + 1))
        nodes = (rng.rand(n, n_atom_feat) < 0.15).astype(np.float64)
        pairs = (rng.rand(n, n, n_pair_feat) < 0.3).astype(np.float64) * rng.rand(n, n, n_pair_feat)
        mols[i] = Mol(nodes, pairs)
    return mols


def build(mode, T, M, d, B, n_tasks, seed):
    import deepchem_amd as dc
    from deepchem_amd.models.torch_models.mpnn import MPNNModel
    torch.manual_seed(seed)
    model = MPNNModel(n_tasks, n_atom_feat=20, n_pair_feat=6, n_hidden=d, T=T, M=M, mode=mode, batch_size=B,
                      device=torch.device(DEV), learning_rate=1e-3)
    # biases away from zero so that their gradients are exercised too
    with torch.no_grad():
        for name, p in model.model.named_parameters():
            if name.endswith("bias") or "_b" in name:
                p.add_(0.05 * torch.randn_like(p))
    return model


@pytest.mark.parametrize("mode,n_tasks", [("regression", 3), ("classification", 2)])
def test_mpnn_forward_and_gradients_match_the_oracle(mode, n_tasks):
    import deepchem_amd as dc
    dc.set_gemm_mode("exact")
    try:
        B, d, T, M = 6, 32, 3, 4
        model = build(mode, T, M, d, B, n_tasks, seed=1)
        mols = qm9_like(B, 20, 6, seed=2)
        rng = np.random.RandomState(3)
        y = rng.randn(B, n_tasks) if mode == "regression" else (rng.rand(B, n_tasks) < 0.5).astype(float)
        w = (rng.rand(B, n_tasks) < 0.8).astype(float)
        ds = dc.data.NumpyDataset(mols, y, w)
        (inputs, labels, weights), = list(model.default_generator(ds, pad_batches=True))
        state = {k: v.detach().cpu() for k, v in model.model.state_dict().items()}
        oracle = MPNNOracle(state, 20, d, T, M, B, mode, n_tasks)
        ref_out = oracle.forward(*inputs)
        ref_loss = oracle.loss(ref_out, labels[0], weights[0])
        ref_loss.backward()
        model._ensure_built()
        model.model.train()
        prepared, lab, wts = model._prepare_batch((inputs, labels, weights))
        outs = model.model(prepared)
        for a, b in zip(outs, ref_out):
            assert np.abs(a.detach().cpu().numpy() - b.detach().numpy()).max() <= 1e-4 * max(1.0, float(b.abs().max()))
        loss = model._loss_fn([outs[i] for i in model._loss_outputs], lab, wts)
        assert abs(float(loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
        loss.backward()
        for name, p in model.model.named_parameters():
            g = p.grad.detach().cpu().numpy()
            r = oracle.p[name].grad.numpy()
            scale = max(np.abs(r).max(), 1e-6)
            assert np.abs(g - r).max() <= 1e-4 * scale, (name, float(np.abs(g - r).max()), scale)
    finally:
        dc.set_gemm_mode("fast")


def test_mpnn_fit_loss_trajectory_and_predict():
    """Three optimizer steps through ``fit`` against the oracle stepped by torch.optim.Adam on the same batches; then
    ``predict`` of a set whose size is not a multiple of the batch size."""
    import deepchem_amd as dc
    dc.set_gemm_mode("exact")
    try:
        B, d, T, M, n_tasks = 5, 32, 2, 3, 2
        model = build("regression", T, M, d, B, n_tasks, seed=4)
        mols = qm9_like(13, 20, 6, seed=5)
        rng = np.random.RandomState(6)
        y, w = rng.randn(13, n_tasks), np.ones((13, n_tasks))
        ds = dc.data.NumpyDataset(mols, y, w)
        state = {k: v.detach().cpu().clone() for k, v in model.model.state_dict().items()}
        oracle = MPNNOracle(state, 20, d, T, M, B, "regression", n_tasks)
        opt = torch.optim.Adam(list(oracle.p.values()), lr=1e-3)
        ref_losses = []
        for inputs, labels, weights in model.default_generator(ds, deterministic=True, pad_batches=True):
            opt.zero_grad()
            l = oracle.loss(oracle.forward(*inputs), labels[0], weights[0])
            l.backward()
            opt.step()
            ref_losses.append(float(l))
        losses = []
        model.fit(ds, nb_epoch=1, deterministic=True, checkpoint_interval=0,
                  callbacks=[lambda m, s, iteration_loss=None: losses.append(float(iteration_loss))])
        assert len(losses) == 3 and np.allclose(losses, ref_losses, rtol=2e-4), (losses, ref_losses)
        pred = model.predict(ds)
        assert pred.shape == (13, n_tasks)
        with torch.no_grad():
            ref = np.concatenate([oracle.forward(*inp)[0].numpy()
                                  for inp, _, _ in model.default_generator(ds, mode='predict', pad_batches=False)])
        assert np.abs(pred - ref).max() <= 5e-3 * max(1.0, np.abs(ref).max())  # three Adam steps apart
    finally:
        dc.set_gemm_mode("fast")

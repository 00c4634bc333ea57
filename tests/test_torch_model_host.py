"""Host logic of ``TorchModel`` (fit bookkeeping, prediction sink, transformers, checkpoints) on a
plain torch module on the CPU.  The contract checked is the reference's
(deepchem/models/torch_models/torch_model.py:345-496, :547-652, :996-1090;
deepchem/models/models.py:162-223; deepchem/utils/evaluate.py:197-307)."""
import os

import numpy as np
import pytest
import torch

import deepchem_amd as dc
from deepchem_amd.models.losses import L2Loss
from deepchem_amd.models.torch_models.torch_model import TorchModel

CPU = torch.device("cpu")


def _regressor(tmp_path, **kw):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))
    return TorchModel(net, L2Loss(), batch_size=10, device=CPU, model_dir=str(tmp_path), learning_rate=1e-2, **kw)


def _data(n=37, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.randn(n, 4)
    y = np.stack([X.sum(1) * 3.0 + 10.0, X[:, 0] - 5.0], 1)
    return dc.data.NumpyDataset(X, y, np.ones_like(y))


def test_predict_undoes_y_transformers_like_manual_unnormalisation(tmp_path):
    raw = _data()
    norm = dc.trans.NormalizationTransformer(transform_y=True, dataset=raw)
    ds = norm.transform(raw)
    model = _regressor(tmp_path)
    model.fit(ds, nb_epoch=3, checkpoint_interval=0)
    plain = model.predict(ds)
    undone = model.predict(ds, [norm])
    assert plain.shape == (37, 2)
    assert np.allclose(undone, plain * norm.y_stds + norm.y_means, rtol=1e-6, atol=1e-6)
    # evaluate: labels and predictions both go back to raw units (utils/evaluate.py:304-307)
    def mae(y_true, y_pred, w):
        return np.abs(y_true - y_pred).mean(0)
    scores = model.evaluate(ds, [mae], [norm], per_task_metrics=True)
    assert np.allclose(scores["mae"], np.abs(raw.y - undone).mean(0), rtol=1e-6)
    assert isinstance(model.evaluate(ds, [mae], [norm])["mae"], float)


def test_predict_with_transformers_rejects_several_outputs(tmp_path):
    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = torch.nn.Linear(4, 2)

        def forward(self, x):
            y = self.l(x)
            return [y, y * 2]

    ds = _data()
    norm = dc.trans.NormalizationTransformer(transform_y=True, dataset=ds)
    model = TorchModel(Two(), L2Loss(), output_types=["prediction", "prediction"], batch_size=10, device=CPU,
                       model_dir=str(tmp_path))
    both = model.predict(ds)
    assert isinstance(both, list) and len(both) == 2 and np.allclose(both[1], 2 * both[0])
    with pytest.raises(ValueError, match="does not support Transformers for models with multiple outputs"):
        model.predict(ds, [norm])


def test_prediction_sink_moves_to_the_host_in_pieces_when_asked(tmp_path, monkeypatch):
    from deepchem_amd.models.torch_models import torch_model as tm
    model = _regressor(tmp_path)
    ds = _data(95)
    whole = model.predict(ds)
    monkeypatch.setattr(tm._OutputSink, "FLUSH_BYTES", 100)  # flush after every batch or two
    assert np.array_equal(model.predict(ds), whole)


def test_output_roles_and_errors(tmp_path):
    model = _regressor(tmp_path)
    assert model._prediction_outputs is None and model._loss_outputs is None
    with pytest.raises(ValueError, match="cannot compute uncertainties"):
        list(model._predict(model.default_generator(_data()), [], True, None))
    with pytest.raises(ValueError, match="no other output_types were specified"):
        model._predict(model.default_generator(_data()), [], False, ["embedding"])
    with pytest.raises(ValueError, match="simultaneously"):
        model._predict(model.default_generator(_data()), [], True, ["embedding"])
    typed = TorchModel(torch.nn.Linear(4, 2), L2Loss(), output_types=["prediction"], device=CPU,
                       model_dir=str(tmp_path / "t"))
    assert typed._prediction_outputs == [0] and typed._loss_outputs == [0] and typed._variance_outputs == []


def test_fit_logging_windows_and_callbacks(tmp_path):
    model = _regressor(tmp_path, log_frequency=3)
    seen, means = [], []
    last = model.fit(_data(), nb_epoch=2, checkpoint_interval=0, deterministic=True, all_losses=means,
                     callbacks=[lambda m, step, iteration_loss=None: seen.append((step, float(iteration_loss))),
                                lambda m, step: seen.append(("old-style", step))])
    steps = [s for s in seen if s[0] != "old-style"]
    assert [s for s, _ in steps] == list(range(1, 9))  # 4 batches x 2 epochs
    assert len([s for s in seen if s[0] == "old-style"]) == 8
    losses = [l for _, l in steps]
    # windows close at steps 3 and 6, the rest (7, 8) at the end
    assert np.allclose(means, [np.mean(losses[0:3]), np.mean(losses[3:6]), np.mean(losses[6:8])], rtol=1e-5)
    assert last == pytest.approx(means[-1])
    assert model.get_global_step() == 8


def test_checkpoint_rotation_and_restore(tmp_path):
    model = _regressor(tmp_path)
    ds = _data()
    model.fit(ds, nb_epoch=1, checkpoint_interval=0)
    snapshots = []
    for _ in range(4):
        model.fit(ds, nb_epoch=1, checkpoint_interval=0)
        model.save_checkpoint(max_checkpoints_to_keep=3)
        snapshots.append({k: v.clone() for k, v in model.model.state_dict().items()})
    names = sorted(os.path.basename(p) for p in model.get_checkpoints())
    assert names == ["checkpoint1.pt", "checkpoint2.pt", "checkpoint3.pt"]
    data = torch.load(os.path.join(str(tmp_path), "checkpoint1.pt"))
    assert set(data) == {"model_state_dict", "optimizer_state_dict", "global_step"}
    # checkpoint1 is the newest, checkpoint3 the oldest kept
    for slot, snap in ((1, snapshots[3]), (3, snapshots[1])):
        d = torch.load(os.path.join(str(tmp_path), "checkpoint%d.pt" % slot))
        assert all(torch.equal(d["model_state_dict"][k], snap[k]) for k in snap)
    model.fit(ds, nb_epoch=1, checkpoint_interval=0)
    model.restore()
    assert all(torch.equal(model.model.state_dict()[k], snapshots[3][k]) for k in snapshots[3])
    assert model.get_global_step() == data["global_step"]
    empty = _regressor(tmp_path / "empty")
    with pytest.raises(ValueError, match="No checkpoint found"):
        empty.restore()


def test_fit_on_batch_checkpoints_only_when_asked(tmp_path):
    model = _regressor(tmp_path)
    ds = _data(10)
    model.fit_on_batch(ds.X, ds.y, ds.w, checkpoint=False)
    assert model.get_checkpoints() == []
    model.fit_on_batch(ds.X, ds.y, ds.w, checkpoint=True)
    # the step hits the interval and fit_generator saves once more at its end, as the reference does
    assert len(model.get_checkpoints()) == 2


def test_variable_subsets_get_their_own_persistent_optimizer(tmp_path):
    model = _regressor(tmp_path)
    ds = _data()
    head = list(model.model[2].parameters())
    before = [p.detach().clone() for p in model.model[0].parameters()]
    model.fit(ds, nb_epoch=1, checkpoint_interval=0, variables=head)
    assert all(torch.equal(a, b) for a, b in zip(before, model.model[0].parameters()))
    opt_a = model._optimizer_for_vars[tuple(head)][0]
    model.fit(ds, nb_epoch=1, checkpoint_interval=0, variables=head)
    assert model._optimizer_for_vars[tuple(head)][0] is opt_a


def test_load_from_pretrained_copies_all_but_the_top(tmp_path):
    src = _regressor(tmp_path / "src")
    src.fit(_data(), nb_epoch=1)
    dst = _regressor(tmp_path / "dst")
    with torch.no_grad():
        for p in dst.model.parameters():
            p.add_(1.0)
    top_before = [p.detach().clone() for p in dst.model[2].parameters()]
    dst.load_from_pretrained(src, include_top=False, model_dir=str(tmp_path / "src"))
    assert all(torch.equal(a, b) for a, b in zip(src.model[0].parameters(), dst.model[0].parameters()))
    assert all(torch.equal(a, b) for a, b in zip(top_before, dst.model[2].parameters()))


def test_flat_arena_homes_parameters_and_keeps_their_values():
    """dist.FlatGradArena(home_params=True): parameters become views of one flat buffer without changing value, shape or
    state_dict; gradients are zeroed views of a second one; load_state_dict keeps both (it copies in place)."""
    import torch
    from deepchem_amd.dist import FlatGradArena
    from deepchem_amd import ops
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    arena = FlatGradArena(net, home_params=True)
    assert arena.params_homed() and arena.covers(net)
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    assert all(off % 4 == 0 for off, _ in arena.slices)
    arena.attach()
    assert arena.intact() and all(float(p.grad.abs().sum()) == 0.0 for p in net.parameters())
    net(torch.randn(4, 5)).sum().backward()
    assert arena.intact() and float(arena.flat.abs().sum()) > 0.0
    net.load_state_dict({k: v + 1 for k, v in before.items()})
    assert arena.params_homed() and torch.equal(net[0].weight, before["0.weight"] + 1)
    assert torch.equal(arena.pflat[:35].view(7, 5), net[0].weight)
    # the switch for in-place weight gradients is off outside its context and restores the previous state
    p = net[0].weight
    assert ops.grad_target(p) is None
    with ops.direct_param_grads(True):
        assert ops.grad_target(p) is p.grad
        with ops.direct_param_grads(False):
            assert ops.grad_target(p) is None
        assert ops.grad_target(p) is p.grad
    assert ops.grad_target(p) is None

"""The one-pass backward of a GraphConv / dense block (csrc/bwd_fused.hip, GCMI_OPT_FUSED_BWD) against the separate
BatchNorm-backward, weight-gradient and input-gradient launches it replaces: same split-bf16 arithmetic, so every
gradient agrees to summation order; and against the oracle's autograd on a batch small enough for the CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _step(packed, y, w, tasks, grad_mode, fused, batch_norm=True, state=None, widths=(64, 64), dense=128, tweak=None,
          mode="classification"):
    """forward + loss + backward of the whole-model entry points on one collated batch; returns the loss, the
    gradient arena and the parameter names with their slices."""
    import deepchem_amd as dc
    from deepchem_amd import _lib
    from deepchem_amd.data.collate import collate_to_device
    from deepchem_amd.metrics import to_one_hot
    n = packed.n_mols
    dbatch = collate_to_device(packed, None, DEV)
    if mode == "classification":
        labels = torch.as_tensor(to_one_hot(y.flatten(), 2).reshape(-1, tasks, 2).astype(np.float32), device=DEV)
    else:
        labels = torch.as_tensor(np.asarray(y, np.float32).reshape(-1, tasks), device=DEV)
    weights = torch.as_tensor(w.astype(np.float32), device=DEV)
    torch.manual_seed(11)
    model = dc.models.torch_models.GraphConvModel(tasks, graph_conv_layers=list(widths), dense_layer_size=dense,
                                                  number_input_features=[75] + list(widths[:-1]), batch_size=n,
                                                  mode=mode, grad_mode=grad_mode, batch_normalize=batch_norm,
                                                  device=DEV)
    if state is not None:
        model.model.load_state_dict({k: v.clone() for k, v in state.items()})
    if tweak is not None:
        with torch.no_grad():
            tweak(model.model)
    native = model.model._native_net()
    assert native is not None
    g = dbatch.graph
    g.set_mols(n)
    import ctypes
    def launches():
        v = ctypes.c_int32(0)
        _lib.call("gcmi_get_option", _lib.GCMI_OPT_FUSED_BWD_LAUNCHES, ctypes.byref(v))
        return v.value
    _lib.call("gcmi_set_option", _lib.GCMI_OPT_FUSED_BWD, 1 if fused else 0)
    try:
        model.model.train()
        before = launches()
        logits, _, fp = native.forward(dbatch.atom_features, g, True, want_probs=False)
        loss = native.loss_backward(labels, weights, n)
        torch.cuda.synchronize()
        # the one-pass kernel really ran (dense layer; plus one per GraphConv layer when everything trains) / did not
        n_conv = len(widths) if grad_mode == "full" else 0
        assert launches() - before == ((1 if batch_norm else 0) + n_conv if fused else 0)
    finally:
        _lib.call("gcmi_set_option", _lib.GCMI_OPT_FUSED_BWD, 1)
    names = [k for k, _ in model.model.named_parameters()]
    model.forward_results = (logits.clone(), fp.clone(),
                             [(bn.running_mean.clone(), bn.running_var.clone()) for bn in model.model.batch_norms
                              if hasattr(bn, "running_mean")])
    return float(loss), native.grad_flat.clone(), list(zip(names, native._slices)), native.grad_range, model


def _compare(ga, gb, slices, rng, tol):
    lo, hi = rng
    worst = (0.0, None)
    for name, (off, n) in slices:
        if not (lo <= off and off + n <= hi):
            continue
        a = ga[off:off + n].double().cpu().numpy()
        b = gb[off:off + n].double().cpu().numpy()
        assert np.isfinite(a).all() and np.isfinite(b).all(), name
        scale = max(np.abs(b).max(), 1e-6)
        err = np.abs(a - b).max() / scale
        if err > worst[0]:
            worst = (err, name)
        assert err <= tol, (name, err, np.abs(b).max())
    return worst


@pytest.mark.parametrize("grad_mode", ["full", "reference"])
@pytest.mark.parametrize("n_mols", [37, 1500])
def test_fused_backward_equals_separate_kernels(grad_mode, n_mols):
    """Both paths split every operand into the same three bf16 pieces and accumulate in fp32; they differ in the
    order rows are added up (64-row tiles walked by persistent workgroups vs row slabs), so gradients agree to
    ~1e-5 of each tensor's scale.  37 molecules: every segment is a ragged tile; 1 500: workgroups cross segment
    boundaries, degrees 0 (single atoms) to 6 and 10 are present, and six molecules of 100-132 atoms get windows of
    their own (the two-stage gather hands those to the separate passes)."""
    from deepchem_amd.utils.synthetic import (concat_packed, single_atom_and_edge_cases, synthetic_labels,
                                              synthetic_molecules)
    parts = [synthetic_molecules(n_mols, seed=5, max_atoms=40), single_atom_and_edge_cases(75, seed=2)]
    if n_mols > 100:  # molecules above the window cap of 96 atoms: oversized windows beside the ordinary ones
        parts.append(synthetic_molecules(6, seed=3, mean_atoms=118, max_atoms=132, min_atoms=100))
    packed = concat_packed(parts)
    tasks = 3
    y, w = synthetic_labels(packed.n_mols, tasks, "classification", 5, pos_rate=0.4)
    l1, g1, sl, r1, m1 = _step(packed, y, w, tasks, grad_mode, True)
    l0, g0, _, r0, m0 = _step(packed, y, w, tasks, grad_mode, False)
    assert r0 == r1
    assert abs(l1 - l0) <= 1e-6 * max(abs(l0), 1.0)
    worst = _compare(g1, g0, sl, r1, 2e-5)
    print("worst relative difference", worst)
    # the forward products of the same switch (fwd_fused.hip against seg_gemm4_kernel): outputs and the BatchNorm
    # statistics taken in their epilogues
    (lg1, fp1, bn1), (lg0, fp0, bn0) = m1.forward_results, m0.forward_results
    assert (lg1 - lg0).abs().max() <= 2e-5 * max(float(lg0.abs().max()), 1.0)
    assert (fp1 - fp0).abs().max() <= 2e-5
    for (rm1, rv1), (rm0, rv0) in zip(bn1, bn0):
        assert (rm1 - rm0).abs().max() <= 1e-6 * max(float(rm0.abs().max()), 1e-3)
        assert (rv1 - rv0).abs().max() <= 1e-6 * max(float(rv0.abs().max()), 1e-3)


def test_fused_backward_without_batchnorm():
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    packed = synthetic_molecules(300, seed=9, max_atoms=30)
    y, w = synthetic_labels(300, 2, "classification", 9, pos_rate=0.4)
    l1, g1, sl, r1, _ = _step(packed, y, w, 2, "full", True, batch_norm=False)
    l0, g0, _, r0, _ = _step(packed, y, w, 2, "full", False, batch_norm=False)
    assert r0 == r1 and abs(l1 - l0) <= 1e-6 * max(abs(l0), 1.0)
    _compare(g1, g0, sl, r1, 2e-5)


def test_fused_backward_against_the_oracle():
    """north_star bound on the gradients themselves: 1e-4 of each tensor's scale against the torch-CPU oracle
    (autograd of the reference's layers, `full` mode) from the same state on the same batch."""
    from oracle import graphconv_oracle as O
    from tests.util import oracle_batch, oracle_convmols
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    n, tasks = 200, 4
    packed = synthetic_molecules(n, seed=21, max_atoms=35)
    y, w = synthetic_labels(n, tasks, "classification", 21, pos_rate=0.4)
    cfg = O.ModelConfig(tasks, batch_size=n)
    state = O.init_state(cfg, 21)
    loss, grads, slices, rng, _ = _step(packed, y, w, tasks, "full", True, state=state)
    tr = O.OracleTrainer(cfg, state, grad_mode="full")
    inputs, labels, weights = oracle_batch(cfg, oracle_convmols(packed), y, w, np.arange(n), n, True)
    ref, _ = tr.loss(inputs, labels, weights)
    ref.backward()
    ref_grads = tr.grads()
    assert abs(loss - float(ref)) <= 1e-4 * max(abs(float(ref)), 1.0)
    checked = 0
    for name, (off, cnt) in slices:
        if ref_grads.get(name) is None:
            continue
        a = grads[off:off + cnt].cpu().numpy()
        b = np.asarray(ref_grads[name], np.float32).reshape(-1)
        scale = max(np.abs(b).max(), 1e-6)
        assert np.abs(a - b).max() <= 1e-4 * scale + 1e-7, (name, np.abs(a - b).max(), scale)
        checked += 1
    assert checked > 40


@pytest.mark.parametrize("grad_mode", ["full", "reference"])
def test_pooled_batchnorm_sums_fall_back_when_ill_conditioned(grad_mode):
    """The BatchNorm backward of a GraphConv block takes sum dy and sum dy * xhat from sums over the POOLED rows that
    the block above leaves behind, (sum dP * P - beta sum dP) / gamma.  A column with |beta| > 64 |gamma| makes that
    division ill-conditioned: the direct column sums take over (kernels launched every step that return at once
    otherwise).  Both regimes against the separate kernels: a tiny gamma under a large beta, and a gamma of zero."""
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    packed = synthetic_molecules(400, seed=13, max_atoms=30)
    y, w = synthetic_labels(400, 2, "classification", 13, pos_rate=0.4)

    def tweak(module):
        for i in (0, 1):
            bn = module.batch_norms[i]
            bn.weight[3] = 1e-3
            bn.bias[3] = 0.7
            bn.weight[9] = 0.0
            bn.bias[9] = -0.2
            bn.bias[20] = 30.0  # |beta| = 30 |gamma|: still the pooled sums

    l1, g1, sl, r1, _ = _step(packed, y, w, 2, grad_mode, True, tweak=tweak)
    l0, g0, _, r0, _ = _step(packed, y, w, 2, grad_mode, False, tweak=tweak)
    assert r0 == r1 and abs(l1 - l0) <= 1e-6 * max(abs(l0), 1.0)
    _compare(g1, g0, sl, r1, 2e-5)


def test_fused_backward_regression_head():
    """L2Loss through the per-molecule kernel (head_bwd.hip, kind 1): loss and every gradient against the separate
    launches, with per-molecule weights and a padded tail (n_rows < batch is covered by the fit tests)."""
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    n, tasks = 500, 3
    packed = synthetic_molecules(n, seed=31, max_atoms=30)
    y, w = synthetic_labels(n, tasks, "regression", 31)
    l1, g1, sl, r1, _ = _step(packed, y, w, tasks, "full", True, mode="regression")
    l0, g0, _, r0, _ = _step(packed, y, w, tasks, "full", False, mode="regression")
    assert r0 == r1 and abs(l1 - l0) <= 1e-6 * max(abs(l0), 1.0)
    _compare(g1, g0, sl, r1, 2e-5)

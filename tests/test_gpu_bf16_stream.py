"""bf16 activation storage in the STREAMING (large-batch) kernels: gcmi_model_* with ``storage = 1`` (VERDICT r2 item 1,
BASELINE.json config 2 "bf16/fp32").

Every matrix the step writes and reads back -- a copy of the atom features, the neighbour sums, the GraphConv outputs,
the pooled rows, the dense output -- is kept as bfloat16 (one round-to-nearest-even when it is stored); products and
sums accumulate in fp32, BatchNorm sums in fp64 FROM THE ROUNDED VALUES, parameters, gradients, gradient streams and
Adam state stay fp32.  Judged three ways:

* against the oracle with the same rounding restated at the same places (``O.bf16_storage()``: straight-through
  gradient, i.e. a backward over the stored values) -- tight: this pins WHERE the kernels round;
* against the plain float32 oracle -- within bf16 resolution (a stored matrix carries 8 significant bits, five of them
  lie between the input and the loss);
* on the real Tox21 file with the REFERENCE-trained model: per-task ROC-AUC within the north_star's +-0.002.
"""
import numpy as np
import pytest
import torch

from tests.test_gpu_round2 import _tox21_model, tox21_splits
from tests.util import load_golden

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _native_step(packed, y, w, tasks, grad_mode, state, storage, train=True):
    import deepchem_amd as dc
    from deepchem_amd.data.collate import collate_to_device
    from deepchem_amd.metrics import to_one_hot
    n = packed.n_mols
    dbatch = collate_to_device(packed, None, DEV)
    labels = torch.as_tensor(to_one_hot(y.flatten(), 2).reshape(-1, tasks, 2).astype(np.float32), device=DEV)
    weights = torch.as_tensor(w.astype(np.float32), device=DEV)
    model = dc.models.torch_models.GraphConvModel(tasks, number_input_features=[75, 64], batch_size=n,
                                                  grad_mode=grad_mode, device=DEV, activation_storage=storage)
    model.model.load_state_dict({k: v.clone() for k, v in state.items()})
    native = model.model._native_net()
    assert native is not None and native.desc.storage == {"fp32": 0, "bf16": 1, "bf16+grads": 2}[storage]
    g = dbatch.graph
    g.set_mols(n)
    assert g.c.n_win > 0
    model.model.train(train)
    logits, _, fp = native.forward(dbatch.atom_features, g, train, want_probs=False)
    if not train:
        torch.cuda.synchronize()
        return logits.cpu(), fp.cpu()
    loss = native.loss_backward(labels, weights, n)
    torch.cuda.synchronize()
    names = [k for k, _ in model.model.named_parameters()]
    stats = [(bn.running_mean.clone().cpu(), bn.running_var.clone().cpu()) for bn in model.model.batch_norms]
    return (float(loss), logits.cpu(), fp.cpu(), native.grad_flat.clone().cpu(), list(zip(names, native._slices)),
            native.grad_range, stats)


def _oracle_step(packed, y, w, tasks, grad_mode, state, bf16):
    import contextlib
    from oracle import graphconv_oracle as O
    from tests.util import oracle_batch, oracle_convmols
    n = packed.n_mols
    cfg = O.ModelConfig(tasks, batch_size=n)
    inputs, labels, weights = oracle_batch(cfg, oracle_convmols(packed), y, w, np.arange(n), n, True)
    tr = O.OracleTrainer(cfg, state, grad_mode=grad_mode, faithful=False)
    with (O.bf16_storage() if bf16 else contextlib.nullcontext()):
        ref, outs = tr.loss(inputs, labels, weights)
        ref.backward()
    return float(ref.detach()), [o.detach() for o in outs], tr.grads(), tr


def _deviations(native, oracle):
    """How far two versions of one step are apart.  Outputs: maximum and mean absolute deviation (logits relative to
    their scale).  Gradients: relative L2 distance of the WHOLE trained gradient vector, and per tensor (rms over the
    tensor's scale).  Maxima over single entries say little here: a bf16 element that rounds the other way (because
    two summation orders differ in the last fp32 bit) moves an entry by 2^-8 of its value, BatchNorm columns of small
    variance and a 130-atom molecule's readout sum amplify that, and a per-degree bias gradient that only a handful of
    atoms feed (degree 10) moves by tens of percent when one of its arg-max routes changes."""
    loss, logits, fp, grads, slices, rng, _ = native
    ref_loss, ref_outs, ref_grads, _ = oracle
    out = {"loss": abs(loss - ref_loss) / max(abs(ref_loss), 1e-6)}
    rl = ref_outs[1].reshape(logits.shape)
    scale = float(rl.abs().max().clamp_min(1.0))
    out["logits_max"] = float((logits - rl).abs().max()) / scale
    out["logits_mean"] = float((logits - rl).abs().mean()) / scale
    out["fp_max"] = float((fp - ref_outs[2]).abs().max())
    out["fp_mean"] = float((fp - ref_outs[2]).abs().mean())
    lo, hi = rng
    num = den = 0.0
    per = {}
    for name, (off, cnt) in slices:
        b = ref_grads.get(name)
        if b is None:
            continue
        assert lo <= off and off + cnt <= hi, name
        a = grads[off:off + cnt].double().numpy()
        assert np.isfinite(a).all(), name
        b = np.asarray(b, np.float64).reshape(-1)
        num += float(((a - b) ** 2).sum())
        den += float((b ** 2).sum())
        per[name] = float(np.sqrt(((a - b) ** 2).mean()) / max(np.abs(b).max(), 1e-12))
    out["grad_l2"] = float(np.sqrt(num / max(den, 1e-300)))
    out["grad_worst_tensor"] = max((v, k) for k, v in per.items())
    return out


def _fmt(d):
    return ("loss %.1e | logits max %.1e mean %.1e | fingerprint max %.1e mean %.1e | gradient: whole vector L2 %.1e, "
            "worst tensor rms/scale %.1e (%s)" % (d["loss"], d["logits_max"], d["logits_mean"], d["fp_max"], d["fp_mean"],
                                                   d["grad_l2"], d["grad_worst_tensor"][0], d["grad_worst_tensor"][1]))


@pytest.fixture(scope="module")
def batch_4096():
    from oracle import graphconv_oracle as O
    from deepchem_amd.utils.synthetic import (concat_packed, single_atom_and_edge_cases, synthetic_labels,
                                              synthetic_molecules)
    packed = concat_packed([synthetic_molecules(4096, seed=11), single_atom_and_edge_cases(75, seed=3),
                            synthetic_molecules(3, seed=6, mean_atoms=118, max_atoms=132, min_atoms=100)])
    tasks = 12
    y, w = synthetic_labels(packed.n_mols, tasks, "classification", 11, pos_rate=0.3)
    cfg = O.ModelConfig(tasks, batch_size=packed.n_mols)
    return packed, y, w, tasks, O.init_state(cfg, 17)


@pytest.mark.parametrize("storage", ["bf16", "bf16+grads"])
@pytest.mark.parametrize("grad_mode", ["full", "reference"])
def test_streaming_step_in_bf16_storage_against_the_oracle(batch_4096, grad_mode, storage):
    """``bf16``: the activations; ``bf16+grads``: also the gradient streams between kernels (dpool, dy, dS, dXs), which
    the restated oracle does NOT round -- so for that mode the gradient bounds against it are looser by the rounding of
    three to five gradient matrices (2^-9 each, averaged out in the weight gradients)."""
    packed, y, w, tasks, state = batch_4096
    native = _native_step(packed, y, w, tasks, grad_mode, state, storage)
    same = _deviations(native, _oracle_step(packed, y, w, tasks, grad_mode, state, bf16=True))
    plain = _deviations(native, _oracle_step(packed, y, w, tasks, grad_mode, state, bf16=False))
    fp32 = _deviations(_native_step(packed, y, w, tasks, grad_mode, state, "fp32"),
                       _oracle_step(packed, y, w, tasks, grad_mode, state, bf16=False))
    print(grad_mode, storage, "| vs the oracle with the same rounding:", _fmt(same))
    print(grad_mode, storage, "| vs the float32 oracle:              ", _fmt(plain))
    print(grad_mode, "fp32 storage | vs the float32 oracle:      ", _fmt(fp32))
    assert fp32["loss"] != plain["loss"]  # the mode is really on
    # (1) WHERE the kernels round: against the oracle with the same rounding restated the step is several times closer
    # than against plain float32 -- a kernel that rounded somewhere else, twice, or by truncation would not be
    assert same["fp_mean"] <= 0.35 * plain["fp_mean"] and same["logits_mean"] <= 0.35 * plain["logits_mean"], (same, plain)
    assert same["grad_l2"] <= (0.5 if storage == "bf16" else 0.8) * plain["grad_l2"], (same, plain)
    assert same["loss"] <= 1e-4
    # (2) against float32: bf16 resolution.  A stored matrix carries 2^-9 relative rounding, five of them lie between
    # the input and the loss; the loss and the gradient vector average it out, single fingerprint entries do not
    assert plain["loss"] <= 1e-3, plain
    assert plain["logits_mean"] <= 4e-3 and plain["logits_max"] <= 8e-2, plain
    assert plain["fp_mean"] <= 6e-3, plain
    # (gradients: rounding makes candidates of GraphPool / GraphGather tie or swap, and every changed arg-max route moves
    # a gradient by one atom's contribution -- the whole vector stays within a few percent, a bias only a handful of
    # degree-10 atoms feed within tens of percent)
    assert plain["grad_l2"] <= 7e-2 and plain["grad_worst_tensor"][0] <= 0.35, plain
    # (3) and the fp32 storage of the same build, same batch, for scale: float32 accuracy
    assert fp32["logits_max"] <= 1e-4 and fp32["fp_mean"] <= 1e-5 and fp32["grad_l2"] <= 2e-3, fp32


def test_streaming_prediction_in_bf16_storage(batch_4096):
    """Eval mode (BatchNorm from the running statistics, no statistics taken): bf16 storage against fp32 storage."""
    packed, y, w, tasks, state = batch_4096
    lg16, fp16 = _native_step(packed, y, w, tasks, "reference", state, "bf16", train=False)
    lg32, fp32 = _native_step(packed, y, w, tasks, "reference", state, "fp32", train=False)
    assert not torch.equal(lg16, lg32)
    d = float((lg16 - lg32).abs().max() / lg32.abs().max())
    dm = float((lg16 - lg32).abs().mean() / lg32.abs().max())
    print("eval logits: deviation max %.2e mean %.2e of scale, fingerprint max %.2e mean %.2e" %
          (d, dm, float((fp16 - fp32).abs().max()), float((fp16 - fp32).abs().mean())))
    # (single fingerprint entries: a 2^-8 step of one stored element through a BatchNorm column of small variance)
    assert d <= 3e-2 and dm <= 2e-3 and float((fp16 - fp32).abs().max()) <= 0.3 and float((fp16 - fp32).abs().mean()) <= 2e-3


def test_streaming_fit_in_bf16_storage_follows_fp32():
    """Two epochs of fit() at a batch size that takes the streaming kernels (2 048 molecules x 18 atoms > the
    small-batch engine's limit), bf16 storage against fp32 storage from the same state on the same batches: per-step
    losses within bf16 resolution of each other, far outside what a wrong stride or a missing conversion would give."""
    import deepchem_amd as dc
    from oracle import graphconv_oracle as O
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    n, B, tasks = 8192, 2048, 12
    packed = synthetic_molecules(n, seed=23)
    y, w = synthetic_labels(n, tasks, "classification", 23, pos_rate=0.3)
    state = O.init_state(O.ModelConfig(tasks, batch_size=B), 5)
    runs = {}
    for storage in ("fp32", "bf16", "bf16+grads"):
        model = dc.models.torch_models.GraphConvModel(tasks, number_input_features=[75, 64], batch_size=B, device=DEV,
                                                      grad_mode="full", learning_rate=1e-3, activation_storage=storage,
                                                      log_frequency=1)
        model.model.load_state_dict({k: v.clone() for k, v in state.items()})
        model.small_batch_engine = False  # (2 048 x 18 atoms is above its limit anyway)
        ds = dc.data.PackedDataset(packed, y, w)
        losses = []
        model.fit(ds, nb_epoch=2, deterministic=True, checkpoint_interval=0, all_losses=losses)
        probs = model.predict(ds)
        runs[storage] = (np.array(losses), probs)
    l32, p32 = runs["fp32"]
    for storage in ("bf16", "bf16+grads"):
        l16, p16 = runs[storage]
        print("fit losses fp32", l32, storage, l16, "max |dprob| %.3f mean %.4f" % (np.abs(p16 - p32).max(), np.abs(p16 - p32).mean()))
        assert len(l32) == len(l16) > 0 and not np.array_equal(l32, l16)
        assert np.allclose(l16, l32, rtol=2e-2), (storage, l16, l32)
        assert np.abs(p16 - p32).mean() < 2e-2
    assert not np.array_equal(runs["bf16"][0], runs["bf16+grads"][0])  # the second mode is really another one


@pytest.mark.parametrize("run", ["b64", "b100"])
def test_real_tox21_bf16_storage_auc_through_the_streaming_kernels(run):
    """north_star on the real file, through the STREAMING path: the REFERENCE's trained parameters and BatchNorm
    statistics (tests/golden/tox21_ref.npz) in the drop-in model with bf16 storage, the whole valid split as one
    1 024-row batch (gcmi_model_forward, storage = 1): per-task ROC-AUC within +-0.002 of the reference's, mean within
    0.001."""
    from deepchem_amd.metrics import roc_auc_per_task
    g = load_golden("tox21_ref.npz")
    prefix = run + "_trained__"
    trained = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    model, _, _, _ = _tox21_model(g, run, trained, activation_storage="bf16")
    model.batch_size = 1024
    model.model.graph_gather.batch_size = 1024
    model.small_batch_engine = False
    _, valid = tox21_splits()
    probs = model.predict(valid)
    ref = g[run + "_valid_probs"]
    assert probs.shape == ref.shape
    auc = roc_auc_per_task(valid.y, probs, valid.w)
    d_auc = np.abs(auc - g[run + "_valid_auc"])
    print(run, "bf16 storage, streaming kernels: max |dprob| %.4f mean |dprob| %.5f max |dAUC| %.4f |d mean AUC| %.5f" %
          (np.abs(probs - ref).max(), np.abs(probs - ref).mean(), np.nanmax(d_auc),
           abs(np.nanmean(auc) - np.nanmean(g[run + "_valid_auc"]))))
    assert np.abs(probs - ref).max() < 0.1 and np.abs(probs - ref).mean() < 5e-3
    assert np.nanmax(d_auc) <= 0.002 and abs(np.nanmean(auc) - np.nanmean(g[run + "_valid_auc"])) <= 0.001

"""ORACLE (test infrastructure, not product code): GraphConvModel on torch-CPU.

A CPU restatement of the reference's graph-convolution path, op for op, on
torch CPU tensors (torch's own kernels -- index, sum, matmul, max, scatter_add,
batch_norm, Adam -- are the reference's third-party tensor runtime and are used
as such).  Every function cites the reference lines it follows
(paths relative to /root/reference/deepchem).

Who may use it: tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg, as the checker / the timed CPU arm only.  The product
(``deepchem_amd``) never imports it; the product raises when ``libgcmi.so``
is missing.

How it is pinned (tests/test_oracle_golden.py):
  * the reference's own golden assets for ``['CCC','C']``
    (models/tests/assets/graphconvlayer_*.npy, graphpoollayer_result.npy,
    graphgatherlayer_result.npy, graphconvmodel_*_classification.npy,
    utils/test/assets/result_segment_{sum,max}.npy), copied as data into
    tests/golden/ref_assets.npz;
  * outputs, loss, gradients, post-Adam parameters and BatchNorm running
    statistics of the reference itself, run in the build container by
    oracle/gen_golden.py (reference imported from /root/reference with an
    rdkit import stub) and committed as tests/golden/model_*.npz.

Two gradient modes:
  * ``"reference"`` -- the torch reference cuts autograd at every GraphConv
    (models/torch_models/layers.py:6204, :6216, :6226, :6244 round-trip through
    NumPy), so GraphConv weights and batch_norms.0 never receive a gradient.
  * ``"full"`` -- the same forward without the cut: the mathematically
    complete backward (what the Keras twin, models/layers.py:151-213, trains
    with).  Its forward is pinned as above; its backward is pinned by fixtures
    taken from the reference forward with the NumPy hops patched out in memory
    (gen_golden.py) and by float64 finite differences (tests).
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

MAX_DEG = 10

# The reference casts to float32 in GraphConv (layers.py:6210-6214 `.type(torch.float32)`) and sums segments in float
# (`data.float()`, utils/pytorch_utils.py:71): WORK_DTYPE is that float32.  ``precision(torch.float64)`` re-runs the
# SAME op sequence in double precision -- not the reference's arithmetic, but the yardstick that tells how far the
# reference's own float32 accumulation (BatchNorm statistics over 10^5..10^6 rows) is from exact arithmetic, which
# bounds how closely ANY other implementation can be asked to match it at large batch sizes.
WORK_DTYPE = torch.float32


def _identity(t):
    return t


# What the reference does NOT have: bf16 storage of the activations (deepchem_amd's opt-in activation_storage="bf16",
# BASELINE.json config 2).  STORE is applied wherever that mode keeps a matrix in HBM (atom features, neighbour sums,
# GraphConv outputs, pooled rows, the dense output); by default it is the identity and this file is the reference's
# arithmetic.  ``bf16_storage()`` makes it a round-to-nearest-even to bfloat16 with a straight-through gradient, which
# is what a backward pass over the STORED values computes.  Used by tests of that mode only; it is not pinned by the
# reference (which has no such mode) -- the float32 oracle stays the anchor those tests also compare with.
STORE = _identity


def _round_bf16_ste(t):
    if not t.is_floating_point():
        return t
    r = t.detach().to(torch.float32).to(torch.bfloat16).to(t.dtype)
    return t + (r - t.detach())


class bf16_storage:
    def __enter__(self):
        global STORE
        self.prev = STORE
        STORE = _round_bf16_ste

    def __exit__(self, *exc):
        global STORE
        STORE = self.prev


class precision:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global WORK_DTYPE
        self.prev = WORK_DTYPE
        WORK_DTYPE = self.dtype

    def __exit__(self, *exc):
        global WORK_DTYPE
        WORK_DTYPE = self.prev


# --------------------------------------------------------------------------- layers
def sum_neigh(atoms: torch.Tensor, deg_adj_lists: Sequence[torch.Tensor],
              max_degree: int = MAX_DEG) -> List[torch.Tensor]:
    """Per degree d=1..max: sum over the d neighbour rows of every degree-d atom.
    models/torch_models/layers.py:6236-6246."""
    out = []
    for deg in range(1, max_degree + 1):
        gathered = atoms[deg_adj_lists[deg - 1]]  # (n_d, d, F)
        out.append(torch.sum(gathered, 1))
    return out


def graph_conv(inputs: Sequence[torch.Tensor], W_list: Sequence[torch.Tensor],
               b_list: Sequence[torch.Tensor], min_degree: int = 0, max_degree: int = MAX_DEG,
               activation=None, grad_mode: str = "reference") -> torch.Tensor:
    """GraphConv.forward, models/torch_models/layers.py:6167-6234.

    inputs = [atom_features, deg_slice, membership, deg_adj_1 .. deg_adj_max].
    Parameter order in W_list / b_list: rel_1, self_1, rel_2, self_2, ...,
    rel_max, self_max, self_0 (:6189-6224).  Both biases are added (:6210-6214).
    """
    atom_features = inputs[0]
    deg_slice = inputs[1]
    deg_adj_lists = inputs[3:]
    cut = grad_mode == "reference"
    summed = [STORE(t) for t in sum_neigh(atom_features, deg_adj_lists, max_degree)]
    if cut:
        summed = [s.detach() for s in summed]  # :6244 / :6204
    split = torch.split(atom_features, deg_slice[:, 1].tolist())  # :6199-6201
    w = iter(W_list)
    b = iter(b_list)
    blocks = []
    for deg in range(1, max_degree + 1):
        rel = summed[deg - 1]
        self_atoms = split[deg - min_degree]
        rel_out = torch.matmul(rel.type(WORK_DTYPE), next(w)) + next(b)
        self_out = torch.matmul(self_atoms.type(WORK_DTYPE), next(w)) + next(b)
        o = rel_out + self_out
        blocks.append(o.detach() if cut else o)  # :6216
    if min_degree == 0:
        o = torch.matmul(split[0].type(WORK_DTYPE), next(w)) + next(b)
        blocks.insert(0, o.detach() if cut else o)  # :6226
    out = torch.concat(blocks, 0)
    if activation is not None:
        out = activation(out)
    return out


def graph_pool(inputs: Sequence[torch.Tensor], min_degree: int = 0,
               max_degree: int = MAX_DEG) -> torch.Tensor:
    """GraphPool.forward, models/torch_models/layers.py:6319-6367: per atom the
    element-wise max over {self} U neighbours; candidate order is self first,
    then neighbours in table order (:6358-6360) -- torch.max(dim) sends the
    gradient to the FIRST maximal candidate."""
    atom_features = inputs[0]
    deg_slice = inputs[1]
    deg_adj_lists = inputs[3:]
    split = torch.split(atom_features, deg_slice[:, 1].tolist())
    blocks = []
    for deg in range(1, max_degree + 1):
        self_atoms = split[deg - min_degree]
        if deg_adj_lists[deg - 1].shape[0] == 0:
            blocks.append(torch.zeros((0, self_atoms.shape[-1]), dtype=WORK_DTYPE))  # :6346-6350
        else:
            cand = torch.concat(
                [torch.unsqueeze(self_atoms, 1), atom_features[deg_adj_lists[deg - 1]]], 1)
            blocks.append(torch.max(cand, 1)[0])
    if min_degree == 0:
        blocks.insert(0, split[0])
    return torch.concat(blocks, 0)


def unsorted_segment_sum(data: torch.Tensor, segment_ids: torch.Tensor,
                         num_segments: int) -> torch.Tensor:
    """utils/pytorch_utils.py:20-74: zeros(num_segments, ...).scatter_add(0, ids, data)."""
    if len(segment_ids.shape) != 1:
        raise AssertionError("segment_ids have be a 1-D tensor")
    if data.shape[0] != segment_ids.shape[0]:
        raise AssertionError("segment_ids should be the same size as dimension 0 of input.")
    idx = segment_ids.view(-1, *([1] * (data.dim() - 1))).expand_as(data)
    out = torch.zeros(num_segments, *data.shape[1:], dtype=WORK_DTYPE).scatter_add(0, idx, data.type(WORK_DTYPE))
    return out.type(data.dtype)


def unsorted_segment_max(data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int,
                         faithful: bool = True) -> torch.Tensor:
    """utils/pytorch_utils.py:473-528: -inf initialised; for every segment a
    masked max over ALL rows (O(num_segments * N * F); :524-526).  Gradient
    goes to the first (lowest row index) maximal row of the segment.

    ``faithful=False`` is a same-result O(N*F) form (sort rows by segment,
    reduce each run) used only where the quadratic loop cannot finish; the two
    are compared in tests, values AND gradients."""
    if len(segment_ids.shape) != 1:
        raise AssertionError("segment_ids have to be a 1-D tensor")
    if data.shape[0] != segment_ids.shape[0]:
        raise AssertionError("segment_ids should be the same size as dimension 0 of input.")
    shape = [num_segments] + list(data.shape[1:])
    if faithful:
        out = torch.full(shape, float("-inf"), dtype=data.dtype)
        expanded = segment_ids.unsqueeze(-1).expand(-1, *data.shape[1:])
        for i in range(num_segments):
            mask = expanded == i
            out[i] = torch.max(data.masked_fill(~mask, float("-inf")), dim=0)[0]
        return out
    # O(N F) with the same values and the same gradient routing: rows grouped by segment with a STABLE sort (row order
    # inside a segment kept), the segment maxima by one scatter-reduce, then -- per (segment, column) -- the first
    # sorted position that attains the maximum, and ONE differentiable gather of those rows: the gradient reaches
    # exactly the first maximal row, as torch.max(dim=0) over the masked rows does.  (A per-segment slice + max is
    # O(N F) forward but its backward builds an N x F zero gradient per segment: 27 s per step at 4 096 molecules.)
    n_rows = data.shape[0]
    if n_rows == 0:
        return torch.full(shape, float("-inf"), dtype=data.dtype)
    flat = data.reshape(n_rows, -1)
    order = torch.argsort(segment_ids, stable=True)
    sdata = flat[order]
    sid = segment_ids[order].unsqueeze(-1).expand_as(sdata)
    with torch.no_grad():
        seg_max = torch.full((num_segments, flat.shape[1]), float("-inf"), dtype=data.dtype)
        seg_max = seg_max.scatter_reduce(0, sid, sdata, "amax", include_self=True)
        pos = torch.arange(n_rows).unsqueeze(-1).expand_as(sdata)
        hit = torch.where(sdata == seg_max.gather(0, sid), pos, torch.full_like(pos, n_rows))
        first = torch.full((num_segments, flat.shape[1]), n_rows, dtype=torch.int64)
        first = first.scatter_reduce(0, sid, hit, "amin", include_self=True)
        empty = first == n_rows
    picked = sdata.gather(0, first.clamp(max=n_rows - 1))
    out = torch.where(empty, torch.full_like(picked, float("-inf")), picked)
    return out.reshape(shape)


def graph_gather(inputs: Sequence[torch.Tensor], batch_size: int, activation=None,
                 faithful: bool = True) -> torch.Tensor:
    """GraphGather.forward, models/torch_models/layers.py:6450-6479:
    concat([segment_sum, segment_max], 1) over molecules, always batch_size rows."""
    atom_features = inputs[0]
    membership = inputs[2].to(torch.int64)
    assert batch_size > 1, "graph_gather requires batches larger than 1"
    s = unsorted_segment_sum(atom_features, membership, batch_size)
    m = unsorted_segment_max(atom_features, membership, batch_size, faithful=faithful)
    out = torch.concat([s, m], 1)
    if activation is not None:
        out = activation(out)
    return out


# --------------------------------------------------------------------------- model
class ModelConfig:
    """Constructor arguments of _GraphConvTorchModel (graphconvmodel.py:77-88)."""

    def __init__(self, n_tasks, number_input_features=(75, 64), graph_conv_layers=(64, 64),
                 dense_layer_size=128, mode="classification", n_classes=2,
                 batch_normalize=True, uncertainty=False, batch_size=100):
        if mode not in ("classification", "regression"):
            raise ValueError("mode must be either 'classification' or 'regression'")
        if uncertainty and mode != "regression":
            raise ValueError("Uncertainty is only supported in regression mode")
        self.n_tasks = n_tasks
        self.number_input_features = list(number_input_features)
        self.graph_conv_layers = list(graph_conv_layers)
        self.dense_layer_size = dense_layer_size
        self.mode = mode
        self.n_classes = n_classes
        self.batch_normalize = batch_normalize
        self.uncertainty = uncertainty
        self.batch_size = batch_size


def init_state(cfg: ModelConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """A full state_dict with the reference's key names, drawn
    from a NumPy RandomState so the same weights can be rebuilt anywhere.
    Xavier-uniform / zero biases like layers.py:6139-6151; nn.Linear weights
    are (out, in)."""
    rng = np.random.RandomState(seed)
    st: Dict[str, torch.Tensor] = {}

    def xavier(fan_in, fan_out, shape):
        a = math.sqrt(6.0 / (fan_in + fan_out))
        return torch.from_numpy(rng.uniform(-a, a, size=shape).astype(np.float32))

    for li, (width, fin) in enumerate(zip(cfg.graph_conv_layers, cfg.number_input_features)):
        for k in range(2 * MAX_DEG + 1):
            st["graph_convs.%d.W_list.%d" % (li, k)] = xavier(fin, width, (fin, width))
        for k in range(2 * MAX_DEG + 1):
            st["graph_convs.%d.b_list.%d" % (li, k)] = torch.from_numpy(
                rng.uniform(-0.1, 0.1, size=(width,)).astype(np.float32))
    # Widths.  The reference's torch model hard-codes 64 in two places -- BatchNorm1d(num_features=64)
    # (graphconvmodel.py:151) and nn.Linear(64, dense_layer_size) (:172) -- so it cannot run graph_conv_layers other than
    # 64 wide at all (MolNet's regression preset [128, 128] / 256, molnet/preset_hyper_parameters.py:128-135, is a
    # preset of the Keras model, whose BatchNormalization and Dense take their width from the layer before them).  For
    # 64-wide layers this is the reference's construction, number for number; for other widths it is the Keras model's
    # rule, and such runs are parity against the restated ALGORITHM only (tests say so).
    last = cfg.graph_conv_layers[-1]
    if cfg.batch_normalize:
        widths = list(cfg.graph_conv_layers) + [cfg.dense_layer_size]  # 64-wide layers: graphconvmodel.py:151
        for i, wdt in enumerate(widths):
            st["batch_norms.%d.weight" % i] = torch.from_numpy(
                rng.uniform(0.5, 1.5, size=(wdt,)).astype(np.float32))
            st["batch_norms.%d.bias" % i] = torch.from_numpy(
                rng.uniform(-0.2, 0.2, size=(wdt,)).astype(np.float32))
            st["batch_norms.%d.running_mean" % i] = torch.zeros(wdt)
            st["batch_norms.%d.running_var" % i] = torch.ones(wdt)
            st["batch_norms.%d.num_batches_tracked" % i] = torch.tensor(0, dtype=torch.int64)
    d = cfg.dense_layer_size
    st["dense.weight"] = xavier(last, d, (d, last))  # nn.Linear(64, dense) graphconvmodel.py:172
    st["dense.bias"] = torch.from_numpy(rng.uniform(-0.1, 0.1, size=(d,)).astype(np.float32))
    if cfg.mode == "classification":
        o = cfg.n_tasks * cfg.n_classes
        st["reshape_dense.weight"] = xavier(2 * d, o, (o, 2 * d))
        st["reshape_dense.bias"] = torch.from_numpy(
            rng.uniform(-0.1, 0.1, size=(o,)).astype(np.float32))
    else:
        st["regression_dense.weight"] = xavier(2 * d, cfg.n_tasks, (cfg.n_tasks, 2 * d))
        st["regression_dense.bias"] = torch.from_numpy(
            rng.uniform(-0.1, 0.1, size=(cfg.n_tasks,)).astype(np.float32))
        if cfg.uncertainty:
            st["uncertainty_dense.weight"] = xavier(2 * d, cfg.n_tasks, (cfg.n_tasks, 2 * d))
            st["uncertainty_dense.bias"] = torch.from_numpy(
                rng.uniform(-0.1, 0.1, size=(cfg.n_tasks,)).astype(np.float32))
    return st


def is_parameter(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var") or
                key.endswith("num_batches_tracked"))


def _batch_norm(x, st, i, training):
    """nn.BatchNorm1d(eps=1e-3, momentum=0.99), graphconvmodel.py:150-165: torch
    momentum semantics, running = (1-m)*running + m*batch (biased variance to
    normalise, unbiased into running_var); updates ``st`` in place."""
    pre = "batch_norms.%d." % i
    if training:
        st[pre + "num_batches_tracked"] += 1
    return F.batch_norm(x, st[pre + "running_mean"], st[pre + "running_var"], st[pre + "weight"],
                        st[pre + "bias"], training, 0.99, 1e-3)


def model_forward(cfg: ModelConfig, st: Dict[str, torch.Tensor], inputs: Sequence[torch.Tensor],
                  bn_training: bool = False, grad_mode: str = "reference",
                  faithful: bool = True) -> List[torch.Tensor]:
    """_GraphConvTorchModel.forward, graphconvmodel.py:188-249.

    inputs = [atom_features, degree_slice, membership, n_samples, deg_adj_1..10].
    Dropout is never applied (training= is never passed by TorchModel,
    torch_model.py:436/:603), so it is absent here.
    Returns the same output list: classification [probs, logits, fingerprint];
    regression [pred, fingerprint]; uncertainty [pred, var, pred, log_var, fp].
    """
    atom_features = inputs[0]
    degree_slice = inputs[1]
    membership = inputs[2].to(torch.int64)
    n_samples = int(inputs[3])
    deg_adjs = [a.to(torch.int64) for a in inputs[4:]]
    x = STORE(atom_features)
    for i in range(len(cfg.graph_conv_layers)):
        W = [st["graph_convs.%d.W_list.%d" % (i, k)] for k in range(2 * MAX_DEG + 1)]
        b = [st["graph_convs.%d.b_list.%d" % (i, k)] for k in range(2 * MAX_DEG + 1)]
        gc = STORE(graph_conv([x, degree_slice, membership] + deg_adjs, W, b, activation=F.relu,
                              grad_mode=grad_mode))
        if cfg.batch_normalize:
            gc = _batch_norm(gc, st, i, bn_training)
        x = STORE(graph_pool([gc, degree_slice, membership] + deg_adjs))
    dense = STORE(F.relu(F.linear(x, st["dense.weight"], st["dense.bias"])))
    if cfg.batch_normalize:
        dense = _batch_norm(dense, st, len(cfg.graph_conv_layers), bn_training)
    fp = graph_gather([dense, degree_slice, membership] + deg_adjs, cfg.batch_size,
                      activation=torch.tanh, faithful=faithful)
    if cfg.mode == "classification":
        logits = torch.reshape(F.linear(fp, st["reshape_dense.weight"], st["reshape_dense.bias"]),
                               (-1, cfg.n_tasks, cfg.n_classes))
        logits = logits[0:n_samples]  # TrimGraphOutput, graphconvmodel.py:31-33
        return [F.softmax(logits, dim=2), logits, fp]
    out = F.linear(fp, st["regression_dense.weight"], st["regression_dense.bias"])[0:n_samples]
    if cfg.uncertainty:
        log_var = F.linear(fp, st["uncertainty_dense.weight"],
                           st["uncertainty_dense.bias"])[0:n_samples]
        return [out, torch.exp(log_var), out, log_var, fp]
    return [out, fp]


def loss_outputs(cfg: ModelConfig, outputs: List[torch.Tensor]) -> List[torch.Tensor]:
    """Which outputs feed the loss: output_types, graphconvmodel.py:351-375 and
    torch_model.py:260-270/:439-440."""
    if cfg.mode == "classification":
        return [outputs[1]]
    if cfg.uncertainty:
        return [outputs[2], outputs[3]]
    return [outputs[0]]


def _consistent(output, labels):
    """models/losses.py:1522-1543 (_make_pytorch_shapes_consistent): pad the
    shorter shape with trailing 1s when the longer one's extra dims are 1."""
    l1, l2 = output.dim(), labels.dim()
    if l1 == l2:
        return output, labels
    if l1 > l2 and all(i == 1 for i in output.shape[l2:]):
        for _ in range(l1 - l2):
            labels = labels.unsqueeze(-1)
        return output, labels
    if l2 > l1 and all(i == 1 for i in labels.shape[l1:]):
        for _ in range(l2 - l1):
            output = output.unsqueeze(-1)
        return output, labels
    raise ValueError("Incompatible shapes for outputs and labels: %s versus %s" %
                     (tuple(output.shape), tuple(labels.shape)))


def batch_loss(cfg: ModelConfig, outs: List[torch.Tensor], labels: torch.Tensor,
               weights: torch.Tensor) -> torch.Tensor:
    """SoftmaxCrossEntropy (models/losses.py:251-259) or L2Loss (:85-94) through
    _StandardLoss (torch_model.py:1275-1294): mean over (B,T) of loss*w; or the
    uncertainty closure (graphconvmodel.py:360-372)."""
    if cfg.mode == "classification":
        losses = -torch.sum(labels * F.log_softmax(outs[0], dim=-1), dim=-1)
    elif cfg.uncertainty:
        o, l = _consistent(outs[0], labels)
        losses = torch.square(o - l) / torch.exp(outs[1]) + outs[1]
    else:
        o, l = _consistent(outs[0], labels)
        losses = F.mse_loss(o, l, reduction="none")
    w = weights
    if w.dim() < losses.dim():
        w = w.reshape(tuple(w.shape) + (1,) * (losses.dim() - w.dim()))
    return (losses * w).mean()


# --------------------------------------------------------------------------- batches
def to_one_hot(y: np.ndarray, n_classes: int = 2) -> np.ndarray:
    """metrics/metric.py:371-400."""
    n = y.shape[0]
    out = np.zeros((n, n_classes))
    out[np.arange(n), y.astype(np.int64)] = 1
    return out


def pad_batch(batch_size, X_b, y_b, w_b, ids_b):
    """data/datasets.py:142-218: tile X, y, ids up to batch_size; weights of the
    padding rows are zero."""
    n = len(X_b)
    if n == batch_size:
        return X_b, y_b, w_b, ids_b
    reps = np.arange(batch_size) % n
    X_out = X_b[reps]
    y_out = None if y_b is None else y_b[reps]
    ids_out = ids_b[reps]
    if w_b is None:
        w_out = None
    else:
        w_out = np.zeros((batch_size,) + w_b.shape[1:], dtype=w_b.dtype)
        w_out[:n] = w_b
    return X_out, y_out, w_out, ids_out


def batch_tensors(multi: dict, n_samples: int, y_b=None, w_b=None, cfg: ModelConfig = None,
                  predict: bool = False):
    """default_generator + _prepare_batch, graphconvmodel.py:405-422 and
    torch_model.py:923-952: float64 -> float32, torch.as_tensor.
    ``multi`` is a dict from oracle.mol_graphs_oracle.agglomerate (or any object
    exposing the same fields converted to a dict)."""
    inputs = [
        torch.as_tensor(np.asarray(multi["atom_features"]).astype(np.float32)),
        torch.as_tensor(np.asarray(multi["deg_slice"])),
        torch.as_tensor(np.asarray(multi["membership"])),
        torch.as_tensor(np.array(n_samples)),
    ] + [torch.as_tensor(np.asarray(a)) for a in multi["deg_adj_lists"][1:]]
    labels = weights = None
    if y_b is not None:
        if cfg is not None and cfg.mode == "classification" and not predict:
            y_b = to_one_hot(y_b.flatten(), cfg.n_classes).reshape(-1, cfg.n_tasks, cfg.n_classes)
        labels = torch.as_tensor(np.asarray(y_b).astype(np.float32))
    if w_b is not None:
        weights = torch.as_tensor(np.asarray(w_b).astype(np.float32))
    return inputs, labels, weights


class OracleTrainer:
    """fit_generator's inner loop (torch_model.py:423-446) on the oracle model:
    zero_grad, forward in train mode, loss, backward, Adam(lr=1e-3,
    betas=(0.9,0.999), eps=1e-8) over ALL parameters (torch_model.py:281-282,
    optimizers.py:231-241; torch skips parameters whose grad is None)."""

    def __init__(self, cfg: ModelConfig, state: Dict[str, torch.Tensor], grad_mode="reference",
                 learning_rate=1e-3, faithful=True):
        self.cfg = cfg
        self.grad_mode = grad_mode
        self.faithful = faithful
        self.state = {k: v.clone() for k, v in state.items()}
        for k, v in self.state.items():
            if is_parameter(k):
                v.requires_grad_(True)
        self.params = [v for k, v in self.state.items() if is_parameter(k)]
        self.opt = torch.optim.Adam(self.params, lr=learning_rate, betas=(0.9, 0.999), eps=1e-8)

    def forward(self, inputs, training):
        return model_forward(self.cfg, self.state, inputs, bn_training=training,
                             grad_mode=self.grad_mode, faithful=self.faithful)

    def loss(self, inputs, labels, weights):
        outs = self.forward(inputs, training=True)
        return batch_loss(self.cfg, loss_outputs(self.cfg, outs), labels, weights), outs

    def train_step(self, inputs, labels, weights) -> float:
        self.opt.zero_grad()
        loss, _ = self.loss(inputs, labels, weights)
        loss.backward()
        self.opt.step()
        return float(loss)

    def predict(self, inputs):
        with torch.no_grad():
            return self.forward(inputs, training=False)

    def grads(self) -> Dict[str, Optional[np.ndarray]]:
        return {
            k: (None if v.grad is None else v.grad.detach().numpy().copy())
            for k, v in self.state.items() if is_parameter(k)
        }

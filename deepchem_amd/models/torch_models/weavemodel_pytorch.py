"""``Weave`` / ``WeaveModel`` with the reference's contract
(deepchem/models/torch_models/weavemodel_pytorch.py: ``Weave`` :19-329, ``WeaveModel`` :331-618),
computed by libgcmi.so.

Reference behaviour kept on purpose (it decides what trains):

* the weave layers, ``dense1`` and the gather re-wrap their inputs (layers.py:4350, :4581), every
  BatchNorm and Dropout runs in eval mode (weavemodel_pytorch.py:291-312): only the fully connected
  stack ``layers2`` (+ their ``layer_bn`` affine parameters) and ``layer_2`` receive gradients;
* ``torch.manual_seed(22)`` before the layers are created (:183), truncated-normal initialisation;
* two fully connected layers are always used (``zip([0, 1], ...)``, :247).

``WeaveMol`` (feat/mol_graphs.py:378-410) is provided here because the featurizer that produces it
needs rdkit, which this package does not ship.
"""
from collections.abc import Sequence as SequenceCollection
from typing import Iterable, List, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from deepchem_amd import ops
from deepchem_amd._lib import GcmiError
from deepchem_amd.metrics import to_one_hot
from deepchem_amd.models.losses import L2Loss, SoftmaxCrossEntropy
from deepchem_amd.models.torch_models import weave_layers as torch_layers
from deepchem_amd.models.torch_models.torch_model import TorchModel


class WeaveMol(object):
    """What the weave featurizer hands to the model for one molecule: atom features ``nodes`` (n, Fa), pair
    features ``pairs`` (n_pairs, Fp) and the ordered atom pairs ``pair_edges`` (2, n_pairs) the pair rows belong
    to, grouped by source atom.  Accessor names are the reference's (feat/mol_graphs.py:378-410)."""

    def __init__(self, nodes, pairs, pair_edges):
        self.nodes, self.pairs, self.pair_edges = nodes, pairs, pair_edges
        self.num_atoms, self.n_features = nodes.shape[:2]

    get_atom_features = lambda self: self.nodes  # noqa: E731
    get_pair_features = lambda self: self.pairs  # noqa: E731
    get_pair_edges = lambda self: self.pair_edges  # noqa: E731
    get_num_atoms = lambda self: self.num_atoms  # noqa: E731
    get_num_features = lambda self: self.n_features  # noqa: E731


class EvalNormActFn(torch.autograd.Function):
    """act(BatchNorm1d in eval mode (x)) with gradients for x, gamma and beta: the reference trains
    the affine parameters of ``layer_bn`` even though the layer normalises with its running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps: float, relu: bool):
        x = ops.rowmajor(x)
        n = gamma.numel()
        one, zero = torch.ones(n, device=x.device), torch.zeros(n, device=x.device)
        invstd, _ = ops.bn_fold_eval(one, zero, running_mean, running_var, eps)
        scale, shift = ops.bn_fold_eval(gamma.detach().contiguous(), beta.detach().contiguous(), running_mean,
                                        running_var, eps)
        y = ops.bn_apply(x, scale, shift)
        if relu:
            ops.relu_bwd_(y, y)  # y *= (y > 0)
        ctx.relu = relu
        ctx.save_for_backward(x, gamma.detach().contiguous(), running_mean.clone(), invstd, scale, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd, scale, y = ctx.saved_tensors
        dy = ops.rowmajor(dy).clone()
        if ctx.relu:
            ops.relu_bwd_(dy, y)
        dgamma, dbeta, _ = ops.bn_bwd(dy, x, gamma, mean, invstd, False)
        dx = ops.bn_apply(dy, scale, torch.zeros_like(scale)) if ctx.needs_input_grad[0] else None
        return dx, dgamma, dbeta, None, None, None, None


def _per_layer(value, count: int) -> list:
    """A per-layer setting: a sequence is taken as given, anything else (a string included) is used for all."""
    if isinstance(value, SequenceCollection) and not isinstance(value, str):
        return value
    return [value] * count


def _act_name(fn) -> str:
    if isinstance(fn, str):
        return fn.lower()
    if fn in (F.relu, torch.relu) or isinstance(fn, nn.ReLU):
        return "relu"
    if fn in (F.tanh, torch.tanh) or isinstance(fn, nn.Tanh):
        return "tanh"
    if fn is None:
        return "linear"
    raise GcmiError("activation %r has no kernel (relu, tanh or None)" % (fn,))


class Weave(nn.Module):
    """Weave modules -> final convolution (dense1) -> WeaveGather -> fully connected stack -> head."""

    def __init__(self, n_tasks: int, n_atom_feat=75, n_pair_feat=14, n_hidden: int = 50, n_graph_feat: int = 128,
                 n_weave: int = 2, fully_connected_layer_sizes: List[int] = [2000, 100],
                 conv_weight_init_stddevs=0.03, weight_init_stddevs=0.01, bias_init_consts=0.0, dropouts=0.25,
                 final_conv_activation_fn=F.tanh, activation_fns='relu', batch_normalize: bool = True,
                 gaussian_expand: bool = True, compress_post_gaussian_expansion: bool = False,
                 mode: str = "classification", n_classes: int = 2, batch_size: int = 100, device=None):
        super(Weave, self).__init__()
        if mode not in ['classification', 'regression']:
            raise ValueError("mode must be either 'classification' or 'regression'")
        n_layers = len(fully_connected_layer_sizes)
        n_atom_feat, n_pair_feat, conv_weight_init_stddevs = (
            _per_layer(v, n_weave) for v in (n_atom_feat, n_pair_feat, conv_weight_init_stddevs))
        weight_init_stddevs, bias_init_consts, dropouts, activation_fns = (
            _per_layer(v, n_layers) for v in (weight_init_stddevs, bias_init_consts, dropouts, activation_fns))
        self.n_tasks, self.n_atom_feat, self.n_pair_feat = n_tasks, n_atom_feat, n_pair_feat
        self.n_hidden, self.n_graph_feat, self.mode, self.n_classes = n_hidden, n_graph_feat, mode, n_classes
        self.n_layers = n_layers
        self.fully_connected_layer_sizes = fully_connected_layer_sizes
        self.weight_init_stddevs, self.bias_init_consts, self.dropouts = weight_init_stddevs, bias_init_consts, dropouts
        self.activation_names = [_act_name(a) for a in activation_fns]
        self.batch_normalize = batch_normalize
        self.n_weave = n_weave
        self.device = torch.device("cuda:0") if device is None else torch.device(device)

        torch.manual_seed(22)
        self.layers = nn.ModuleList()
        for ind in range(n_weave):
            last = ind == n_weave - 1
            layer = torch_layers.WeaveLayer(n_atom_input_feat=n_atom_feat[ind], n_pair_input_feat=n_pair_feat[ind],
                                            n_atom_output_feat=n_hidden if last else n_atom_feat[ind + 1],
                                            n_pair_output_feat=n_hidden if last else n_pair_feat[ind + 1],
                                            batch_normalize=batch_normalize, device=self.device)
            names = ["W_AA", "W_PA", "W_A"] + (["W_AP", "W_PP", "W_P"] if layer.update_pair else [])
            for name in names:  # initialised on the host in the reference's order, then moved
                w = torch.empty(tuple(getattr(layer, name).shape))
                nn.init.trunc_normal_(w, 0, std=conv_weight_init_stddevs[ind])
                setattr(layer, name, w.to(self.device))
            self.layers.append(layer)
        self.dense1 = nn.Linear(n_hidden, self.n_graph_feat)
        self.dense1_act = _act_name(final_conv_activation_fn)
        self.dense1_bn = nn.BatchNorm1d(num_features=self.n_graph_feat, eps=1e-3, momentum=0.99, affine=True,
                                        track_running_stats=True)
        self.weave_gather = torch_layers.WeaveGather(batch_size, n_input=self.n_graph_feat,
                                                     gaussian_expand=gaussian_expand,
                                                     compress_post_gaussian_expansion=compress_post_gaussian_expansion,
                                                     device=self.device)
        if n_layers > 0:
            self.layers2 = nn.ModuleList()
            in_size = self.n_graph_feat * 11
            for ind, layer_size, weight_stddev, bias_const, dropout in zip(
                    [0, 1], fully_connected_layer_sizes, weight_init_stddevs, bias_init_consts, dropouts):
                layer = nn.Linear(in_size, layer_size)
                nn.init.trunc_normal_(layer.weight, 0, std=weight_stddev)
                if layer.bias is not None:
                    layer.bias = nn.Parameter(torch.full(layer.bias.shape, bias_const))
                layer.layer_bn = nn.BatchNorm1d(num_features=layer_size, eps=1e-3, momentum=0.99, affine=True,
                                                track_running_stats=True)
                layer.weight_stddev = weight_stddev
                layer.bias_const = bias_const
                layer.dropout = nn.Dropout(dropout)
                self.layers2.append(layer)
                in_size = layer_size
        head_columns = n_tasks * n_classes if mode == 'classification' else n_tasks
        self.layer_2 = nn.Linear(fully_connected_layer_sizes[1], head_columns)

    def forward(self, inputs) -> List[torch.Tensor]:
        """inputs = [atom_features, pair_features, pair_split, atom_split, atom_to_pair]."""
        dev = self.dense1.weight.device
        if dev.type != 'cuda':
            raise GcmiError("Weave: the model must live on the GPU; deepchem_amd has no CPU implementation")
        layer_in = [inputs[0], inputs[1], inputs[2], inputs[4]]
        for ind in range(self.n_weave):
            A, P = self.layers[ind](layer_in)
            layer_in = [A, P, inputs[2], inputs[4]]
        # final convolution: the gather re-wraps its input in the reference, so nothing in front of
        # it trains -- plain kernels, no autograd bookkeeping
        with torch.no_grad():
            dense1 = ops.LinearFn.apply(A, self.dense1.weight, self.dense1.bias, False)
            if self.dense1_act == "tanh":
                ops.tanh_(dense1)
            elif self.dense1_act == "relu":
                ops.relu_bwd_(dense1, dense1)
            if self.batch_normalize:
                bn = self.dense1_bn
                scale, shift = ops.bn_fold_eval(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
                dense1 = ops.bn_apply(dense1, scale, shift)
            gathered = self.weave_gather([dense1, inputs[3]])
        out = gathered
        if self.n_layers > 0:
            for ind, act in zip([0, 1], self.activation_names):
                layer = self.layers2[ind]
                relu = act == "relu"
                if act not in ("relu", "linear"):
                    raise GcmiError("fully connected activation %r has no kernel" % act)
                if self.batch_normalize:
                    out = ops.LinearFn.apply(out, layer.weight, layer.bias, False)
                    bn = layer.layer_bn
                    out = EvalNormActFn.apply(out, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)
                else:
                    out = ops.LinearFn.apply(out, layer.weight, layer.bias, relu)
        if self.mode == 'classification':
            logits = ops.LinearFn.apply(out, self.layer_2.weight, self.layer_2.bias, False)
            logits = torch.reshape(logits, (-1, self.n_tasks, self.n_classes))
            return [ops.SoftmaxFn.apply(logits), logits]
        return [ops.LinearFn.apply(out, self.layer_2.weight, self.layer_2.bias, False)]


class WeaveModel(TorchModel):
    """Google-style weave graph convolutions (Kearnes et al. 2016) under the TorchModel loop."""

    def __init__(self, n_tasks: int, n_atom_feat=75, n_pair_feat=14, n_hidden: int = 50, n_graph_feat: int = 128,
                 n_weave: int = 2, fully_connected_layer_sizes: List[int] = [2000, 100],
                 conv_weight_init_stddevs=0.03, weight_init_stddevs=0.01, bias_init_consts=0.0,
                 weight_decay_penalty: float = 0.0, weight_decay_penalty_type: str = "l2", dropouts=0.25,
                 final_conv_activation_fn=F.tanh, activation_fns='relu', batch_normalize: bool = True,
                 gaussian_expand: bool = True, compress_post_gaussian_expansion: bool = False,
                 mode: str = "classification", n_classes: int = 2, batch_size: int = 100, **kwargs):
        self.mode = mode
        if mode not in ['classification', 'regression']:
            raise ValueError("mode must be either 'classification' or 'regression'")
        device = kwargs.get("device")
        self.model = Weave(n_tasks=n_tasks, n_atom_feat=n_atom_feat, n_pair_feat=n_pair_feat, n_hidden=n_hidden,
                           n_graph_feat=n_graph_feat, n_weave=n_weave,
                           fully_connected_layer_sizes=fully_connected_layer_sizes,
                           conv_weight_init_stddevs=conv_weight_init_stddevs, weight_init_stddevs=weight_init_stddevs,
                           bias_init_consts=bias_init_consts, dropouts=dropouts,
                           final_conv_activation_fn=final_conv_activation_fn, activation_fns=activation_fns,
                           batch_normalize=batch_normalize, gaussian_expand=gaussian_expand,
                           compress_post_gaussian_expansion=compress_post_gaussian_expansion, mode=mode,
                           n_classes=n_classes, batch_size=batch_size, device=device)
        penalised = [layer.weight for layer in getattr(self.model, 'layers2', [])]
        magnitude = torch.abs if weight_decay_penalty_type == 'l1' else torch.square

        def weight_penalty():
            return weight_decay_penalty * torch.stack([magnitude(w).sum() for w in penalised]).sum()

        classify = mode == 'classification'
        super(WeaveModel, self).__init__(self.model, loss=SoftmaxCrossEntropy() if classify else L2Loss(),
                                         output_types=['prediction', 'loss'] if classify else ['prediction'],
                                         batch_size=batch_size,
                                         regularization_loss=weight_penalty if weight_decay_penalty != 0.0 else None,
                                         **kwargs)
        self._flat_step = True  # parameters, gradients and Adam moments in flat buffers (TorchModel._ensure_built)

    def compute_features_on_batch(self, X_b):
        """WeaveMol objects -> ``(atom_feat, pair_feat, pair_split, atom_split, atom_to_pair)``
        (weavemodel_pytorch.py:516-578): atom and pair rows concatenated molecule by molecule; ``atom_to_pair``
        (n_pairs, 2) holds the pairs in batch atom numbering, ``pair_split`` its first column (the atom a pair row
        is summed into), ``atom_split`` the molecule of each atom."""
        sizes = [mol.get_num_atoms() for mol in X_b]
        first_atom = np.concatenate(([0], np.cumsum(sizes)[:-1])).tolist()
        atom_to_pair = np.concatenate([mol.get_pair_edges().T + shift for mol, shift in zip(X_b, first_atom)], axis=0)
        atom_split = np.repeat(np.arange(len(sizes)), sizes)
        atom_feat = np.concatenate([mol.get_atom_features() for mol in X_b], axis=0)
        pair_feat = np.concatenate([mol.get_pair_features() for mol in X_b], axis=0)
        return atom_feat, pair_feat, np.ascontiguousarray(atom_to_pair[:, 0]), atom_split, atom_to_pair

    def _prepare_batch(self, batch):
        # the index arrays stay on the host: the layers build their CSR plans from them
        inputs, labels, weights = batch
        atom_feat, pair_feat, pair_split, atom_split, atom_to_pair = inputs
        input_tensors = [self._to_device(atom_feat), self._to_device(pair_feat), np.asarray(pair_split),
                         np.asarray(atom_split), np.asarray(atom_to_pair)]
        label_tensors = [self._to_device(x) for x in labels] if labels is not None else []
        weight_tensors = [self._to_device(x) for x in weights] if weights is not None else []
        return (input_tensors, label_tensors, weight_tensors)

    def default_generator(self, dataset, epochs: int = 1, mode: str = 'fit', deterministic: bool = True,
                          pad_batches: bool = True) -> Iterable[Tuple[List, List, List]]:
        net = self.model
        for _ in range(epochs):
            for X_b, y_b, w_b, _ids in dataset.iterbatches(batch_size=self.batch_size, deterministic=deterministic,
                                                           pad_batches=pad_batches):
                if net.mode == 'classification' and y_b is not None:
                    y_b = to_one_hot(y_b.flatten(), net.n_classes).reshape(-1, net.n_tasks, net.n_classes)
                yield (self.compute_features_on_batch(X_b), [y_b], [w_b])

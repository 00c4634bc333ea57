"""``_GraphConvTorchModel`` / ``GraphConvModel`` with the reference's contract
(deepchem/models/torch_models/graphconvmodel.py), running on libgcmi.so.

Same constructor arguments, sub-module names (= checkpoint keys:
``graph_convs.{i}.W_list.{k}``, ``batch_norms.{i}.*``, ``dense.*``,
``reshape_dense.*`` | ``regression_dense.*`` | ``uncertainty_dense.*``), output
lists, ``output_types``, losses, ``default_generator`` and errors.  A
checkpoint written by either implementation loads into the other.

What runs where: every tensor op of the forward and backward pass is a HIP
kernel behind the C ABI (ops.py); BatchNorm is folded into its consumer
(GraphPool / GraphGather read ``x*scale+shift`` on the fly), so the normalised
activations are never written to HBM.  The ``nn.BatchNorm1d`` / ``nn.Linear``
sub-modules only hold parameters and buffers under the reference's names.

Reference behaviours kept on purpose: dropout is a no-op
unless ``training=True`` is passed to the module directly
(graphconvmodel.py:217/:226 vs torch_model.py:436); BatchNorm momentum 0.99 in
torch semantics; hard-coded width 64 of BatchNorm / dense input (:151, :172);
GraphGather emits ``batch_size`` rows and only predictions are trimmed to
``n_samples``.  ``grad_mode="reference"`` (default) also keeps the autograd cut
at GraphConv; ``grad_mode="full"`` trains every parameter.
"""
from collections.abc import Sequence as SequenceCollection
from typing import Any, Callable, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from deepchem_amd import ops
from deepchem_amd.feat.mol_graphs import ConvMol
from deepchem_amd.graph import BatchGraph, graph_for_layer_inputs
from deepchem_amd.metrics import to_one_hot
from deepchem_amd.models.losses import L2Loss, SoftmaxCrossEntropy, _make_pytorch_shapes_consistent
from deepchem_amd.models.torch_models import layers as torch_layers
from deepchem_amd.models.torch_models.torch_model import TorchModel
from deepchem_amd.utils.pytorch_utils import get_activation


_BF16_NATIVE_ONLY = ("activation_storage='bf16' runs on the library's own step only (gcmi_small_* for small batches, "
                     "gcmi_model_* with storage = 1 for large ones): the model's own loss and optimizer, no dropout in "
                     "training mode, no uncertainty head; the layer-by-layer autograd path has no bf16 form")


class TrimGraphOutput(nn.Module):
    """Trim the fixed-size GraphGather batch to the real sample count
    (graphconvmodel.py:21-33)."""

    def forward(self, inputs):
        n_samples = int(inputs[1])
        return inputs[0][0:n_samples]


class _BNApplyFn(torch.autograd.Function):
    """Stand-alone BatchNorm1d (only used when dropout sits between it and its consumer)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, training, eps, momentum):
        x = ops.rowmajor(x)
        if training:
            mean, invstd, scale, shift = ops.bn_stats(x, gamma, beta, rm, rv, eps, momentum)
        else:
            mean = invstd = None
            scale, shift = ops.bn_fold_eval(gamma, beta, rm, rv, eps)
        ctx.training = training
        ctx.save_for_backward(x, gamma, mean, invstd)
        return ops.bn_apply(x, scale, shift)

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise NotImplementedError("gradients through an eval-mode BatchNorm are not implemented")
        x, gamma, mean, invstd = ctx.saved_tensors
        dgamma, dbeta, dx = ops.bn_bwd(ops.rowmajor(dy), x, gamma, mean, invstd, True)
        return dx, dgamma, dbeta, None, None, None, None, None


class _GraphConvTorchModel(nn.Module):
    """Graph convolution network of Duvenaud et al. (graphconvmodel.py:36-249)."""

    def __init__(self, n_tasks: int, number_input_features: List[int],
                 graph_conv_layers: List[int] = [64, 64], dense_layer_size: int = 128, dropout=0.0,
                 mode: str = "classification", number_atom_features: int = 75, n_classes: int = 2,
                 batch_normalize: bool = True, uncertainty: bool = False, batch_size: int = 100,
                 grad_mode: str = "reference", activation_storage: str = "fp32"):
        super(_GraphConvTorchModel, self).__init__()
        if mode not in ['classification', 'regression']:
            raise ValueError("mode must be either 'classification' or 'regression'")
        # "bf16": the activations a step writes and reads back; "bf16+grads": also the gradient streams between kernels
        if activation_storage not in ("fp32", "bf16", "bf16+grads"):
            raise ValueError("activation_storage must be 'fp32', 'bf16' or 'bf16+grads'")
        # "bf16": every matrix a step writes and reads back (GraphConv outputs, pooled rows, dense output) is kept
        # as bfloat16, arithmetic and accumulation stay fp32 (the reference has no such mode).  Opt-in.
        self.activation_storage = activation_storage
        self.n_tasks: int = n_tasks
        self.n_classes: int = n_classes
        self.mode: str = mode
        self.uncertainty: bool = uncertainty
        self.grad_mode = grad_mode
        if not isinstance(dropout, SequenceCollection):
            dropout = [dropout] * (len(graph_conv_layers) + 1)
        if len(dropout) != len(graph_conv_layers) + 1:
            raise ValueError('Wrong number of dropout probabilities provided')
        if uncertainty:
            if mode != "regression":
                raise ValueError("Uncertainty is only supported in regression mode")
            if any(d == 0.0 for d in dropout):
                raise ValueError('Dropout must be included in every layer to predict uncertainty')
        self.graph_convs = nn.ModuleList([
            torch_layers.GraphConv(layer_size, input_size, activation_fn=get_activation('relu'),
                                   grad_mode=grad_mode)
            for layer_size, input_size in zip(graph_conv_layers, number_input_features)
        ])
        # the reference hard-codes width 64 here (:151, :172); same for the default
        # [64, 64] layers, generalised to the actual layer widths otherwise
        bn_widths = list(graph_conv_layers)
        self.batch_norms = nn.ModuleList([
            nn.BatchNorm1d(num_features=w, eps=1e-3, momentum=0.99, affine=True,
                           track_running_stats=True) if batch_normalize else nn.Identity()
            for w in bn_widths
        ])
        self.batch_norms.append(
            nn.BatchNorm1d(num_features=dense_layer_size, eps=1e-3, momentum=0.99, affine=True,
                           track_running_stats=True) if batch_normalize else nn.Identity())
        self.dropouts = nn.ModuleList(
            [nn.Dropout(rate) if rate > 0.0 else nn.Identity() for rate in dropout])
        self.graph_pools = nn.ModuleList([torch_layers.GraphPool() for _ in graph_conv_layers])
        self.dense = nn.Linear(graph_conv_layers[-1] if graph_conv_layers else 64, dense_layer_size)
        self.dense_act = F.relu
        self.graph_gather = torch_layers.GraphGather(batch_size=batch_size,
                                                     activation_fn=get_activation('tanh'))
        self.trim = TrimGraphOutput()
        if self.mode == 'classification':
            self.reshape_dense = nn.Linear(dense_layer_size * 2, n_tasks * n_classes)
        else:
            self.regression_dense = nn.Linear(dense_layer_size * 2, n_tasks)
            if self.uncertainty:
                self.uncertainty_dense = nn.Linear(dense_layer_size * 2, n_tasks)
                self.uncertainty_trim = TrimGraphOutput()

    def _has_dropout(self) -> bool:
        return any(not isinstance(d, nn.Identity) for d in self.dropouts)

    def _native_net(self, quick: bool = False):
        """The fused whole-model driver (deepchem_amd/native.py), or None when this
        configuration is outside what it covers (uncertainty head, exotic widths)."""
        from deepchem_amd.native import NativeNet, NativeUnsupported
        nat = self.__dict__.get("_native")
        if nat is None and not self.__dict__.get("_native_failed", False):
            try:
                nat = NativeNet(self)
                self.__dict__["_native"] = nat
            except NativeUnsupported:
                self.__dict__["_native_failed"] = True
                return None
        if nat is not None:
            try:
                nat.refresh(quick)
            except NativeUnsupported:
                self.__dict__["_native"] = None
                self.__dict__["_native_failed"] = True
                return None
        return nat

    def _native_outputs(self, native, x, graph, n_samples: int, bn_training: bool):
        graph.set_mols(self.graph_gather.batch_size)
        logits, probs, fp = native.forward(x, graph, bn_training)
        if self.mode == 'classification':
            shape = (-1, self.n_tasks, self.n_classes)
            return [probs.view(shape)[0:n_samples], logits.view(shape)[0:n_samples], fp]
        return [logits[0:n_samples], fp]

    def _bn_args(self, i: int):
        bn = self.batch_norms[i]
        if isinstance(bn, nn.BatchNorm1d):
            training = self.training and bn.training
            if training:
                bn.num_batches_tracked += 1
            return (bn.weight, bn.bias, bn.running_mean, bn.running_var), True, training, bn.eps, bn.momentum
        return (None, None, None, None), False, False, 0.0, 0.0

    def forward(self, inputs, training=False) -> List[torch.Tensor]:
        """inputs = [atom_features, degree_slice, membership, n_samples, deg_adj_1..10]
        (graphconvmodel.py:202-208), or a ``deepchem_amd.data.collate.DeviceBatch``."""
        graph = getattr(inputs, "graph", None)
        if graph is not None:  # pre-collated batch already resident on the GPU
            atom_features, n_samples = inputs.atom_features, inputs.n_samples
        else:
            atom_features = inputs[0]
            n_samples = inputs[3]
            if not atom_features.is_cuda:
                raise ops._lib.GcmiError(
                    "GraphConvModel: inputs must be CUDA tensors (no CPU path in deepchem_amd)")
            graph = graph_for_layer_inputs([inputs[0], inputs[1], inputs[2]] + list(inputs[4:]),
                                           atom_features.device)
        n_samples = int(n_samples)
        x = atom_features.to(torch.float32)
        if not torch.is_grad_enabled() and not (training and self._has_dropout()):
            native = self._native_net()
            if native is not None:  # prediction: the whole forward is one C call
                return self._native_outputs(native, x, graph, n_samples, self.training)
        if self.activation_storage != "fp32":
            raise NotImplementedError(_BF16_NATIVE_ONLY)
        for i in range(len(self.graph_convs)):
            bn_t, has_bn, bn_train, eps, mom = self._bn_args(i)
            use_dropout = training and not isinstance(self.dropouts[i], nn.Identity)
            # a training-mode BatchNorm folded into the pool also folds the ReLU derivative
            # of the GraphConv in front of it into its backward (one pass less over N x 64)
            mask_in_bn = has_bn and bn_train and not use_dropout and self.grad_mode == "full"
            gc = self.graph_convs[i]([x, None, None], graph=graph, grad_masked=mask_in_bn)
            if use_dropout:
                if has_bn:
                    gc = _BNApplyFn.apply(gc, *bn_t, bn_train, eps, mom)
                gc = self.dropouts[i](gc)
                x = ops.PoolFn.apply(gc, None, None, None, None, graph, False, False, 0.0, 0.0, False)
            else:
                x = ops.PoolFn.apply(gc, *bn_t, graph, has_bn, bn_train, eps, mom, mask_in_bn)
        bn_t, has_bn, bn_train, eps, mom = self._bn_args(len(self.graph_convs))
        use_dropout = training and not isinstance(self.dropouts[-1], nn.Identity)
        mask_in_bn = has_bn and bn_train and not use_dropout
        dense = ops.LinearFn.apply(x, self.dense.weight, self.dense.bias, True, mask_in_bn)
        batch_size = self.graph_gather.batch_size
        assert batch_size > 1, "graph_gather requires batches larger than 1"
        if use_dropout:
            if has_bn:
                dense = _BNApplyFn.apply(dense, *bn_t, bn_train, eps, mom)
            dense = self.dropouts[-1](dense)
            neural_fingerprint = ops.ReadoutFn.apply(dense, None, None, None, None, graph, batch_size,
                                                     False, False, 0.0, 0.0, True, False)
        else:
            neural_fingerprint = ops.ReadoutFn.apply(dense, *bn_t, graph, batch_size, has_bn, bn_train,
                                                     eps, mom, True, mask_in_bn)
        if self.mode == 'classification':
            logits = ops.LinearFn.apply(neural_fingerprint, self.reshape_dense.weight,
                                        self.reshape_dense.bias, False, False)
            logits = torch.reshape(logits, (-1, self.n_tasks, self.n_classes))
            logits = self.trim([logits, n_samples])
            output = ops.SoftmaxFn.apply(logits)
            outputs = [output, logits, neural_fingerprint]
        else:
            output = ops.LinearFn.apply(neural_fingerprint, self.regression_dense.weight,
                                        self.regression_dense.bias, False, False)
            output = self.trim([output, n_samples])
            if self.uncertainty:
                log_var = ops.LinearFn.apply(neural_fingerprint, self.uncertainty_dense.weight,
                                             self.uncertainty_dense.bias, False, False)
                log_var = self.uncertainty_trim([log_var, n_samples])
                var = torch.exp(log_var)
                outputs = [output, var, output, log_var, neural_fingerprint]
            else:
                outputs = [output, neural_fingerprint]
        return outputs


class _SmallPredictPlan:
    """A prediction pass over small batches: ``run()`` yields the model's output list once per chunk of
    batches (one C call each, deepchem_amd/small.py), predictions trimmed to the real molecules of every
    batch, the embedding left at ``batch_size`` rows per batch as the reference leaves it
    (graphconvmodel.py:234-236)."""

    def __init__(self, owner, engine, packed, index_batches):
        self.owner, self.engine, self.packed, self.index_batches = owner, engine, packed, index_batches

    def __iter__(self):  # only _predict consumes this object
        raise TypeError("a small-batch prediction plan is consumed by GraphConvModel._predict")

    def run(self):
        from deepchem_amd.small import ChunkCollator, HeldChunks
        o = self.owner
        B, T = o.batch_size, o.n_tasks
        C = o.n_classes if o.mode == 'classification' else 1
        TC, F2 = T * C, 2 * o.model.dense.out_features
        dev = o.device
        collator = o.__dict__.get("_small_collator")
        if collator is None or collator.packed is not self.packed or collator.mols_out != B:
            collator = ChunkCollator(self.packed, dev, B)
            o.__dict__["_small_collator"] = collator
        self.engine.native.refresh()
        held = HeldChunks(dev)
        buf_idx, buf_real = [], []

        def flush():
            ch = collator.collate(buf_idx, buf_real)
            nb = ch.n_batches
            logits = torch.empty((nb * B, TC), dtype=torch.float32, device=dev)
            probs = torch.empty((nb * B, TC), dtype=torch.float32, device=dev) if o.mode == 'classification' else None
            fp = torch.empty((nb * B, F2), dtype=torch.float32, device=dev)
            collator.bind(ch, buf_real, logits=logits, probs=probs, logit_stride=TC, fp=fp, fp_stride=F2)
            self.engine.predict(ch.descs, ch.max_atoms, B)
            held.hold(ch)
            if all(r == B for r in buf_real[:-1]):
                rows = slice(0, (nb - 1) * B + buf_real[-1])
                take = lambda t: t[rows]
            else:
                keep = torch.as_tensor(np.concatenate([np.arange(b * B, b * B + r) for b, r in enumerate(buf_real)]),
                                       device=dev)
                take = lambda t: t.index_select(0, keep)
            del buf_idx[:], buf_real[:]
            if o.mode == 'classification':
                return [take(probs).view(-1, T, C), take(logits).view(-1, T, C), fp]
            return [take(logits), fp]

        for idx, n_real in self.index_batches:
            buf_idx.append(idx)
            buf_real.append(int(n_real))
            if len(buf_idx) >= o.small_chunk_batches:
                yield flush()
        if buf_idx:
            yield flush()
        held.drain()


class GraphConvModel(TorchModel):
    """Graph convolutional model with the ``dc.models.torch_models.GraphConvModel``
    interface (graphconvmodel.py:252-422).

    >>> model = GraphConvModel(12, number_input_features=[75, 64], batch_size=100,
    ...                        mode='classification')                      # doctest: +SKIP
    >>> loss = model.fit(dataset, nb_epoch=10)                             # doctest: +SKIP
    """

    def __init__(self, n_tasks: int, number_input_features: List[int],
                 graph_conv_layers: List[int] = [64, 64], dense_layer_size: int = 128,
                 dropout: float = 0.0, mode: str = "classification", number_atom_features: int = 75,
                 n_classes: int = 2, batch_size: int = 100, batch_normalize: bool = True,
                 uncertainty: bool = False, grad_mode: str = "reference", activation_storage: str = "fp32", **kwargs):
        self.mode: str = mode
        self.n_tasks: int = n_tasks
        self.n_classes: int = n_classes
        self.batch_size: int = batch_size
        self.uncertainty: bool = uncertainty
        self._native_checked = False
        model = _GraphConvTorchModel(n_tasks, graph_conv_layers=graph_conv_layers,
                                     number_input_features=number_input_features,
                                     dense_layer_size=dense_layer_size, dropout=dropout, mode=mode,
                                     number_atom_features=number_atom_features, n_classes=n_classes,
                                     batch_normalize=batch_normalize, uncertainty=uncertainty,
                                     batch_size=batch_size, grad_mode=grad_mode, activation_storage=activation_storage)
        loss: Union[SoftmaxCrossEntropy, L2Loss, Callable[[Any, Any, Any], Any]]
        if mode == "classification":
            output_types = ['prediction', 'loss', 'embedding']
            loss = SoftmaxCrossEntropy()
        else:
            if self.uncertainty:
                output_types = ['prediction', 'variance', 'loss', 'loss', 'embedding']

                def loss(outputs, labels, weights):
                    output, labels = _make_pytorch_shapes_consistent(outputs[0], labels[0])
                    losses = torch.square(output - labels) / torch.exp(outputs[1]) + outputs[1]
                    w = weights[0]
                    if len(w.shape) < len(losses.shape):
                        shape = tuple(w.shape)
                        w = torch.reshape(w, shape + (1,) * (len(losses.shape) - len(w.shape)))
                    return torch.mean(losses * w)
            else:
                output_types = ['prediction', 'embedding']
                loss = L2Loss()
        super(GraphConvModel, self).__init__(model, loss, output_types=output_types,
                                             batch_size=batch_size, **kwargs)

    def _train_step(self, inputs, labels, weights, loss, optimizer):
        """One optimizer step.  With the model's own loss and optimizer this is three C calls
        (forward, loss + backward, Adam over the flat gradient range); anything custom goes through
        the autograd path of TorchModel._train_step."""
        from deepchem_amd.models.optimizers import GcmiAdam
        from deepchem_amd.models.torch_models.torch_model import _StandardLoss
        native = None
        if (loss is self._loss_fn and isinstance(loss, _StandardLoss) and self.regularization_loss is None
                and isinstance(optimizer, GcmiAdam) and optimizer is self._pytorch_optimizer
                and len(labels) == 1 and len(weights) == 1 and not self.uncertainty):
            native = self.model._native_net(quick=self._native_checked)
            self._native_checked = native is not None
        if native is None:
            return super(GraphConvModel, self)._train_step(inputs, labels, weights, loss, optimizer)
        graph = getattr(inputs, "graph", None)
        if graph is not None:
            x, n_samples = inputs.atom_features, inputs.n_samples
        else:
            x, n_samples = inputs[0], int(inputs[3])
            graph = graph_for_layer_inputs([inputs[0], inputs[1], inputs[2]] + list(inputs[4:]), x.device)
        graph.set_mols(self.model.graph_gather.batch_size)
        if optimizer._flat is None or optimizer._flat["p"].data_ptr() != native.flat.data_ptr():
            optimizer.attach_flat(native.flat, native.grad_flat, native._slices)
        native.forward(x.to(torch.float32), graph, True, want_probs=False)
        batch_loss = native.loss_backward(labels[0], weights[0], int(n_samples))
        lo, hi = native.grad_range
        if self._grad_sync is not None:
            if hasattr(self._grad_sync, "reduce_flat"):
                self._grad_sync.reduce_flat(native.grad_flat[lo:hi])  # zero-copy bucket
            else:
                self._grad_sync(self.model)
        optimizer.step_flat(lo, hi)
        return batch_loss

    def fit_generator(self, *args, **kwargs):
        self._native_checked = False  # full parameter-view check on the first step of every fit
        return super(GraphConvModel, self).fit_generator(*args, **kwargs)

    # ------------------------------------------------------------------ the small-batch engine
    # set to False to keep fit()/predict() on the per-batch path whatever the batch size
    small_batch_engine: bool = True
    # batches of a chunk (one C call); chunks end early at checkpoint steps
    small_chunk_batches: int = 48

    def _packed_view(self, dataset, epochs, deterministic, pad_batches):
        """(packed molecule set, y, w, iterator of (molecule indices, real molecules) per batch) for the dataset
        kinds the native pipelines understand, else None.  The index batches reproduce ``iterbatches`` (order,
        per-epoch shuffles drawn from np.random exactly as the reference draws them, shard walk, padding)."""
        from deepchem_amd.data.datasets import DiskDataset, NumpyDataset
        from deepchem_amd.data.packed_dataset import (PackedDataset, disk_index_batches, packed_from_convmols,
                                                      packed_from_disk)
        if type(dataset) is DiskDataset and len(dataset) > 0:
            conv = packed_from_disk(dataset)
            if conv is None:
                return None
            packed, y, w, shard_offsets = conv
            return packed, y, w, disk_index_batches(dataset, shard_offsets, self.batch_size, epochs, deterministic,
                                                    pad_batches)
        if type(dataset) is PackedDataset:
            packed, y, w = dataset.packed, dataset.y, dataset.w
        elif type(dataset) is NumpyDataset and getattr(dataset.X, "dtype", None) == object and len(dataset) > 0 \
                and hasattr(dataset.X[0], "get_atom_features"):
            packed = dataset.__dict__.get("_gcmi_packed")
            if packed is None:  # one conversion per dataset object
                packed = packed_from_convmols(dataset.X)
                dataset.__dict__["_gcmi_packed"] = packed
            y, w = dataset.y, dataset.w
        else:
            return None
        helper = PackedDataset(packed, y, w)
        return packed, y, w, helper.iter_index_batches(self.batch_size, epochs, deterministic, pad_batches)

    def _small_engine(self):
        """The engine for this model and these batch sizes, or None (per-batch path)."""
        from deepchem_amd.small import SMALL_MAX_ATOMS, SmallBatchEngine, SmallUnsupported
        # (dropout layers do not matter: fit() and predict() never pass training=True, graphconvmodel.py:217/:226)
        if not self.small_batch_engine or self.device.type != 'cuda' or self.uncertainty or self.batch_size < 2:
            return None
        native = self.model._native_net()
        if native is None:
            return None
        eng = self.__dict__.get("_small")
        if eng is None or eng.native is not native:
            try:
                eng = SmallBatchEngine(native)
            except SmallUnsupported:
                return None
            self.__dict__["_small"] = eng
        return eng

    def _labels_for_small(self, packed, y, w, fit: bool):
        """Whole-set labels (after the one-hot transform) and weights (one per task) in HBM, float32."""
        from deepchem_amd.data.packed_dataset import resident_labels
        n = packed.n_mols
        T = self.n_tasks
        y = np.asarray(y)
        # a y of another size goes to the per-batch path, which raises the reference's own shape error for it (the
        # one-hot reshape below would raise an opaque one -- or silently misalign labels when the sizes divide)
        if y.size != n * T:
            return None
        if self.mode == 'classification':
            def one_hot(a):
                return to_one_hot(a.flatten(), self.n_classes).reshape(-1, T, self.n_classes)
            y_dev = resident_labels(packed, self.device, y, one_hot, ("y", ("one_hot", T, self.n_classes)))
            label_stride = T * self.n_classes
        else:
            y_dev = resident_labels(packed, self.device, y.reshape(n, T), None, ("y", None))
            label_stride = T
        w = np.asarray(w)
        if w.size == n and T > 1:  # one weight per molecule: broadcast over the tasks (_StandardLoss)
            w = np.repeat(w.reshape(n, 1), T, axis=1)
        if w.size != n * T:
            return None
        w_dev = resident_labels(packed, self.device, w.reshape(n, T), None, ("w", None))
        return y_dev.reshape(n, label_stride), label_stride, w_dev

    def fit(self, dataset, nb_epoch: int = 10, max_checkpoints_to_keep: int = 5, checkpoint_interval: int = 1000,
            deterministic: bool = False, restore: bool = False, variables=None, loss=None, callbacks=[],
            all_losses=None) -> float:
        """``TorchModel.fit`` (torch_model.py:289-343).  Batches small enough to live in L2 take the small-batch
        engine: the whole loop of fit_generator runs inside libgcmi.so, ``small_chunk_batches`` optimizer steps
        per call (same batches, same order, same losses, logging windows and checkpoint steps)."""
        done = None
        # (data parallel: the engine takes a gradient exchange that works on the flat arena, deepchem_amd.dist)
        plain = (variables is None and loss is None and not callbacks and self.regularization_loss is None
                 and (self._grad_sync is None or hasattr(self._grad_sync, "reduce_flat")))
        if plain:
            done = self._fit_small(dataset, nb_epoch, max_checkpoints_to_keep, checkpoint_interval, deterministic,
                                   restore, all_losses)
        if done is not None:
            return done
        return super(GraphConvModel, self).fit(dataset, nb_epoch, max_checkpoints_to_keep, checkpoint_interval,
                                               deterministic, restore, variables, loss, callbacks, all_losses)

    def _fit_small(self, dataset, nb_epoch, max_keep, interval, deterministic, restore, all_losses):
        from deepchem_amd.models.optimizers import GcmiAdam
        from deepchem_amd.models.torch_models.torch_model import _LossWindow, _StandardLoss, logger
        from deepchem_amd.small import SMALL_MAX_ATOMS, ChunkCollator, HeldChunks
        if not isinstance(self._loss_fn, _StandardLoss) or self.device.type != 'cuda':
            return None
        self._ensure_built()
        if self._lr_schedule is not None or not isinstance(self._pytorch_optimizer, GcmiAdam):
            return None
        engine = self._small_engine()
        if engine is None:
            return None
        view = self._packed_view(dataset, nb_epoch, deterministic, True)
        if view is None:
            return None
        packed, y, w, index_batches = view
        if packed.n_mols == 0 or y is None or w is None:
            return None
        # the engine is for batches whose activations stay in L2: judge by the set's mean molecule size
        if self.batch_size * (packed.n_atoms / max(packed.n_mols, 1)) > SMALL_MAX_ATOMS:
            return None
        from deepchem_amd.data.collate import _is_symmetric
        if not _is_symmetric(packed):
            return None
        labels = self._labels_for_small(packed, y, w, True)
        if labels is None:
            return None
        y_dev, label_stride, w_dev = labels
        if restore:
            self.restore()
        self.model.train()
        engine.native.refresh()
        B = self.batch_size
        T = self.n_tasks
        collator = self.__dict__.get("_small_collator")
        if collator is None or collator.packed is not packed or collator.mols_out != B:
            collator = ChunkCollator(packed, self.device, B)
            self.__dict__["_small_collator"] = collator
        held = HeldChunks(self.device)
        window = _LossWindow(all_losses)
        pending = []  # (first step, device losses) not yet folded into the logging windows
        import time as _time
        started = _time.time()

        def fold(final: bool):
            # logging windows close at multiples of log_frequency; the host reads the losses chunk by chunk, late
            while pending and (final or len(pending) > 3):
                first, dev_losses = pending.pop(0)
                for k, v in enumerate(dev_losses.cpu().tolist()):
                    window.total += v
                    window.count += 1
                    if (first + k) % self.log_frequency == 0:
                        window.close(first + k)

        from deepchem_amd.small import chunks_ahead
        step0 = self._global_step

        def cut_after(k):  # the optimizer step that a checkpoint follows ends its chunk
            return interval > 0 and (step0 + k) % interval == interval - 1

        def checked(batches):
            for idx, n_real in batches:
                if idx.shape[0] != B:
                    raise ValueError("the small-batch engine needs padded batches")
                yield idx, n_real

        for host_chunk, at_checkpoint in chunks_ahead(collator, checked(index_batches), self.small_chunk_batches,
                                                      cut_after):
            ch = collator.to_device(host_chunk)
            y_t = y_dev.index_select(0, ch.sel_dev)
            w_t = w_dev.index_select(0, ch.sel_dev)
            for b, r in enumerate(ch.n_real):
                if r < B:
                    w_t[b * B + r:(b + 1) * B] = 0  # padding rows of a ragged last batch carry no weight
            collator.bind(ch, [B] * ch.n_batches, labels=y_t, label_stride=label_stride, weights=w_t, weight_stride=T)
            losses = engine.fit(ch.descs, self._pytorch_optimizer, ch.max_atoms, B, grad_sync=self._grad_sync)
            held.hold(ch)
            pending.append((self._global_step + 1, losses))
            self._global_step += ch.n_batches
            fold(False)
            if at_checkpoint:
                self.save_checkpoint(max_keep)
        fold(True)
        window.close(self._global_step)
        held.drain()
        if interval > 0:
            self.save_checkpoint(max_keep)
        logger.info("TIMING: model fitting took %0.3f s" % (_time.time() - started))
        return window.last_mean

    # set to False to force the reference's per-batch Python collation
    native_batches: bool = True

    def _fast_generator(self, dataset, epochs, mode, deterministic, pad_batches):
        """Same batches as ``default_generator`` (order, shuffling, padding, one-hot labels), but
        collated natively into a pinned arena and copied to the GPU by a prefetching worker; the
        model receives a ``DeviceBatch`` instead of the 14 host arrays.  Returns None when the
        dataset is not one this path understands."""
        from deepchem_amd.data.datasets import DiskDataset, NumpyDataset
        from deepchem_amd.data.packed_dataset import (DeviceBatchPipeline, PackedDataset, disk_index_batches,
                                                      packed_from_convmols, packed_from_disk)
        if not self.native_batches or self.device.type != 'cuda':
            return None
        index_batches = None
        if type(dataset) is DiskDataset and len(dataset) > 0:
            # every shard is converted to flat arrays once; the reference's shard walk (shard
            # order, per-shard shuffles, carry-over, padding) then only moves molecule indices
            conv = packed_from_disk(dataset)
            if conv is None:
                return None
            packed, y, w, shard_offsets = conv
            index_batches = disk_index_batches(dataset, shard_offsets, self.batch_size, epochs, deterministic,
                                               pad_batches)
        elif type(dataset) is PackedDataset:
            packed, y, w = dataset.packed, dataset.y, dataset.w
        elif type(dataset) is NumpyDataset and getattr(dataset.X, "dtype", None) == object and len(dataset) > 0 \
                and hasattr(dataset.X[0], "get_atom_features"):
            packed = dataset.__dict__.get("_gcmi_packed")
            if packed is None:  # one conversion per dataset object
                packed = packed_from_convmols(dataset.X)
                dataset.__dict__["_gcmi_packed"] = packed
            y, w = dataset.y, dataset.w
        else:
            return None
        if index_batches is None:
            helper = PackedDataset(packed, y, w)
            index_batches = helper.iter_index_batches(self.batch_size, epochs, deterministic, pad_batches)
        label_fn = None
        if self.mode == 'classification' and mode != 'predict':
            label_fn = lambda y_b: to_one_hot(y_b.flatten(), self.n_classes).reshape(
                -1, self.n_tasks, self.n_classes)
        # label_key names the transform, so that the converted labels uploaded by one fit() serve the next
        pipe = DeviceBatchPipeline(packed, y, w, index_batches, self.device, label_fn,
                                   label_key=None if label_fn is None else ("one_hot", self.n_tasks, self.n_classes))

        def gen():
            for batch, y_t, w_t in pipe:
                yield (batch, [y_t], [w_t])

        return gen()

    def _small_predict_plan(self, dataset, deterministic):
        from deepchem_amd.small import SMALL_MAX_ATOMS
        engine = self._small_engine()
        if engine is None:
            return None
        view = self._packed_view(dataset, 1, deterministic, False)
        if view is None:
            return None
        packed, _, _, index_batches = view
        if packed.n_mols == 0 or self.batch_size * (packed.n_atoms / packed.n_mols) > SMALL_MAX_ATOMS:
            return None
        return _SmallPredictPlan(self, engine, packed, index_batches)

    def _predict(self, generator, transformers, uncertainty, other_output_types):
        if not isinstance(generator, _SmallPredictPlan):
            return super(GraphConvModel, self)._predict(generator, transformers, uncertainty, other_output_types)
        from deepchem_amd.models.torch_models.torch_model import _OutputSink
        _OutputSink.check(self._roles, uncertainty, other_output_types)
        sink = _OutputSink(self._roles, transformers, uncertainty, other_output_types)
        self._ensure_built()
        self.model.eval()
        for outputs in generator.run():
            sink.push(outputs)
        return sink.result()

    def _batch_generator(self, dataset, epochs: int = 1, mode: str = 'fit',
                         deterministic: bool = True, pad_batches: bool = True):
        """What fit()/predict*() iterate: the native pipeline when the dataset allows it, else
        ``default_generator`` (which keeps the reference's exact contract for direct callers)."""
        if mode == 'predict' and epochs == 1 and not pad_batches:
            plan = self._small_predict_plan(dataset, deterministic)
            if plan is not None:
                return plan
        fast = self._fast_generator(dataset, epochs, mode, deterministic, pad_batches)
        if fast is not None:
            return fast
        return self.default_generator(dataset, epochs=epochs, mode=mode, deterministic=deterministic,
                                      pad_batches=pad_batches)

    def default_generator(self, dataset, epochs: int = 1, mode: str = 'fit',
                          deterministic: bool = True, pad_batches: bool = True):
        """Batches as ``([atom_features, deg_slice, membership, n_samples, adj_1..adj_10],
        [y], [w])`` (graphconvmodel.py:382-422)."""
        for epoch in range(epochs):
            for (X_b, y_b, w_b, ids_b) in dataset.iterbatches(batch_size=self.batch_size,
                                                              deterministic=deterministic,
                                                              pad_batches=pad_batches):
                if y_b is not None and self.mode == 'classification' and not (mode == 'predict'):
                    y_b = to_one_hot(y_b.flatten(), self.n_classes).reshape(
                        -1, self.n_tasks, self.n_classes)
                multiConvMol = ConvMol.agglomerate_mols(X_b)
                n_samples = np.array(X_b.shape[0])
                inputs = [
                    multiConvMol.get_atom_features(), multiConvMol.deg_slice,
                    np.array(multiConvMol.membership), n_samples
                ]
                for i in range(1, len(multiConvMol.get_deg_adjacency_lists())):
                    inputs.append(multiConvMol.get_deg_adjacency_lists()[i])
                yield (inputs, [y_b], [w_b])

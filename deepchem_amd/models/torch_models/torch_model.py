"""``TorchModel``: the fit / predict / checkpoint driver the GraphConv models run under.

The public surface is the reference's (deepchem/models/torch_models/torch_model.py: ``fit`` :289,
``fit_generator`` :345, ``fit_on_batch`` :498, ``predict*`` :654-839, ``save_checkpoint`` :996,
``restore`` :1061, ``load_from_pretrained`` :1196, ``_StandardLoss`` :1267): method names, argument
meaning, return values, checkpoint dictionary keys, ``checkpoint<N>.pt`` rotation and error messages
are contract.  How the work is organised underneath is this repository's own:

* a fit is a ``_FitRun``: it resolves optimizer + schedule once, owns a ``_LossWindow`` that sums the
  batch losses ON THE DEVICE (the host reads one number per logging window, never one per step) and a
  checkpoint cadence; ``TorchModel._train_step`` is the single overridable unit of work (the
  GraphConv model swaps in its fused native step there);
* a prediction pass is an ``_OutputSink``: it picks the requested output roles, leaves the batch
  outputs in HBM (no device->host copy, hence no stream drain, per batch), joins them on the device,
  copies once and undoes the y-transformers on the joined array (they are row-wise maps);
* ``_OutputRoles`` indexes the model's output list by role once, at construction.

TensorBoard / W&B hooks and ``torch.compile`` are not part of this path and are absent.
"""
import logging
import os
import time
from collections.abc import Sequence as SequenceCollection
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from deepchem_amd import ops
from deepchem_amd.data.datasets import NumpyDataset
from deepchem_amd.models.losses import Loss
from deepchem_amd.models.models import Model
from deepchem_amd.models.optimizers import Adam, LearningRateSchedule, Optimizer
from deepchem_amd.trans.transformers import undo_transforms

logger = logging.getLogger(__name__)


class _OutputRoles:
    """Positions of the model outputs by role (``output_types``; torch_model.py:229-254)."""

    ROLES = ("prediction", "loss", "variance")

    def __init__(self, output_types: Optional[List[str]]):
        self.declared = output_types is not None
        by_role: Dict[str, List[int]] = {r: [] for r in self.ROLES}
        other: List[int] = []
        for pos, kind in enumerate(output_types or ()):
            by_role[kind].append(pos) if kind in by_role else other.append(pos)
        self.prediction, self.variance, self.other = by_role["prediction"], by_role["variance"], other
        # a model without dedicated loss outputs is scored on its predictions
        self.loss = by_role["loss"] or self.prediction

    def attr(self, positions: List[int]) -> Optional[List[int]]:
        """The reference exposes ``None`` instead of lists when no output_types were given."""
        return positions if self.declared else None


class _LossWindow:
    """Running mean of the batch losses between two log lines; the sum lives on the device."""

    def __init__(self, collect: Optional[List[float]]):
        self.total: Any = 0.0
        self.count = 0
        self.last_mean = 0.0
        self.collect = collect

    def add(self, batch_loss: torch.Tensor) -> None:
        self.total = self.total + batch_loss.detach()
        self.count += 1

    def close(self, step: int) -> None:
        if self.count == 0:
            return
        self.last_mean = float(self.total) / self.count  # the only device->host read of the window
        logger.info('Ending global_step %d: Average loss %g' % (step, self.last_mean))
        if self.collect is not None:
            self.collect.append(self.last_mean)
        self.total, self.count = 0.0, 0


class _FitRun:
    """Everything one ``fit_generator`` call needs besides the batches."""

    def __init__(self, owner: "TorchModel", variables, loss, callbacks, max_keep: int, interval: int,
                 all_losses: Optional[List[float]]):
        self.owner = owner
        self.loss = owner._loss_fn if loss is None else loss
        self.optimizer, self.schedule = owner._optimizer_and_schedule(variables)
        self.callbacks = list(callbacks) if isinstance(callbacks, SequenceCollection) else [callbacks]
        self.max_keep, self.interval = max_keep, interval
        self.window = _LossWindow(all_losses)

    def after_step(self, batch_loss: torch.Tensor) -> None:
        owner = self.owner
        if self.schedule is not None:
            self.schedule.step()
        owner._global_step += 1
        step = owner._global_step
        self.window.add(batch_loss)
        if step % owner.log_frequency == 0:
            self.window.close(step)
        if self.interval > 0 and step % self.interval == self.interval - 1:
            owner.save_checkpoint(self.max_keep)
        for cb in self.callbacks:
            try:
                cb(owner, step, iteration_loss=batch_loss)
            except TypeError:  # callbacks written against the two-argument form
                cb(owner, step)

    def finish(self) -> float:
        self.window.close(self.owner._global_step)
        if self.interval > 0:
            self.owner.save_checkpoint(self.max_keep)
        return self.window.last_mean


class _OutputSink:
    """Collects the requested outputs of a prediction pass batch by batch."""

    def __init__(self, roles: _OutputRoles, transformers: List, uncertainty: bool, other_output_types):
        self.roles = roles
        self.transformers = list(transformers)
        self.uncertainty = uncertainty
        if other_output_types:
            self.take = list(roles.other)
        elif roles.declared:
            self.take = list(roles.prediction)
        else:
            self.take = []
        self.columns: Optional[List["_Column"]] = None
        self.var_columns: Optional[List["_Column"]] = None
        self._pending = 0

    @staticmethod
    def check(roles: _OutputRoles, uncertainty: bool, other_output_types) -> None:
        if uncertainty and (other_output_types is not None):
            raise ValueError(
                'This model cannot compute uncertainties and other output types simultaneously. Please invoke one at a time.'
            )
        if uncertainty:
            if not roles.declared or len(roles.variance) == 0:
                raise ValueError('This model cannot compute uncertainties')
            if len(roles.variance) != len(roles.prediction):
                raise ValueError('The number of variances must exactly match the number of outputs')
        if other_output_types:
            if not roles.declared or len(roles.other) == 0:
                raise ValueError(
                    'This model cannot compute other outputs since no other output_types were specified.'
                )

    FLUSH_BYTES = 1 << 30  # outputs wait in HBM until this much has piled up, then move to the host in one go

    def push(self, outputs: List[torch.Tensor]) -> None:
        """Queue one batch.  Nothing is copied to the host here: a per-batch ``.cpu()`` would drain the
        stream after every batch, which is what bounds small-batch prediction."""
        outputs = [t.detach() for t in outputs]
        if self.uncertainty:
            var = [outputs[i] for i in self.roles.variance]
            if self.var_columns is None:
                self.var_columns = [_Column() for _ in var]
            for col, v in zip(self.var_columns, var):
                col.add(v)
        picked = [outputs[i] for i in self.take] if self.take else outputs
        if self.transformers and len(picked) > 1:
            raise ValueError("predict() does not support Transformers for models with multiple outputs.")
        if self.columns is None:
            self.columns = [_Column() for _ in picked]
        for col, p in zip(self.columns, picked):
            col.add(p)
        self._pending += sum(p.numel() * p.element_size() for p in picked)
        if self._pending > self.FLUSH_BYTES:
            for col in (self.columns or []) + (self.var_columns or []):
                col.to_host()
            self._pending = 0

    def result(self):
        joined = [c.joined() for c in (self.columns or [])]
        if self.transformers:  # row-wise maps: undoing them on the joined array equals batch by batch
            joined = [undo_transforms(a, self.transformers) for a in joined]
        if self.uncertainty and self.var_columns is not None:
            return zip(joined, [c.joined() for c in self.var_columns])
        return joined[0] if len(joined) == 1 else joined


class _Column:
    """One output across the batches of a pass: device tensors until moved, then numpy pieces."""

    def __init__(self):
        self.device_parts: List[torch.Tensor] = []
        self.host_parts: List[np.ndarray] = []

    def add(self, t: torch.Tensor) -> None:
        self.device_parts.append(t)

    def to_host(self) -> None:
        if self.device_parts:
            whole = self.device_parts[0] if len(self.device_parts) == 1 else torch.cat(self.device_parts, dim=0)
            self.host_parts.append(whole.cpu().numpy())
            self.device_parts = []

    def joined(self) -> np.ndarray:
        self.to_host()
        return self.host_parts[0] if len(self.host_parts) == 1 else np.concatenate(self.host_parts, axis=0)


class TorchModel(Model):

    def __init__(self, model: torch.nn.Module, loss, output_types: Optional[List[str]] = None,
                 batch_size: int = 100, model_dir: Optional[str] = None,
                 learning_rate: Union[float, LearningRateSchedule] = 0.001,
                 optimizer: Optional[Optimizer] = None, log_frequency: int = 100,
                 device: Optional[torch.device] = None,
                 regularization_loss: Optional[Callable] = None, **kwargs) -> None:
        super(TorchModel, self).__init__(model=model, model_dir=model_dir, **kwargs)
        self.loss = loss
        self._loss_fn = _StandardLoss(self, loss) if isinstance(loss, Loss) else loss
        self.learning_rate = learning_rate
        self.optimizer = optimizer if optimizer is not None else Adam(learning_rate=learning_rate)
        self.regularization_loss = regularization_loss
        self.batch_size = batch_size
        self.log_frequency = log_frequency
        self.device = self._pick_device(device)
        self.model = model.to(self.device)
        self.output_types = output_types
        self._roles = _OutputRoles(output_types)
        # the reference's attribute names (read by subclasses and user code)
        self._prediction_outputs = self._roles.attr(self._roles.prediction)
        self._loss_outputs = self._roles.attr(self._roles.loss)
        self._variance_outputs = self._roles.attr(self._roles.variance)
        self._other_outputs = self._roles.attr(self._roles.other)
        self._built = False
        self._optimizer_for_vars: Dict[Any, Any] = {}
        # set by deepchem_amd.dist.shard_model(); None = single process
        self._grad_sync: Optional[Callable] = None

    @staticmethod
    def _pick_device(device: Optional[torch.device]) -> torch.device:
        if device is None:
            device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        return device

    def _new_schedule(self, torch_optimizer):
        lr = self.optimizer.learning_rate
        return lr._create_pytorch_schedule(torch_optimizer) if isinstance(lr, LearningRateSchedule) else None

    def _ensure_built(self) -> None:
        if self._built:
            return
        self._built = True
        self._global_step = 0
        self._pytorch_optimizer = self.optimizer._create_pytorch_optimizer(self.model.parameters())
        self._lr_schedule = self._new_schedule(self._pytorch_optimizer)
        if getattr(self, "_flat_step", False) and hasattr(self._pytorch_optimizer, "attach_flat"):
            # models whose backward is autograd over libgcmi.so kernels (MPNNModel): parameters, gradients and Adam
            # moments in flat buffers -- one memset, gradients written in place, one Adam launch per step
            from deepchem_amd.dist import FlatGradArena
            try:
                arena = FlatGradArena(self.model, home_params=True)
                self._pytorch_optimizer.attach_flat(arena.pflat, arena.flat, arena.slices)
                self._grad_arena = arena
            except ValueError:
                self._grad_arena = None

    def _optimizer_and_schedule(self, variables):
        """The model-wide optimizer, or one per distinct ``variables`` subset (kept across calls so
        that its moments survive; torch_model.py:402-421)."""
        if variables is None:
            return self._pytorch_optimizer, self._lr_schedule
        key = tuple(variables)
        if key not in self._optimizer_for_vars:
            opt = self.optimizer._create_pytorch_optimizer(key)
            self._optimizer_for_vars[key] = (opt, self._new_schedule(opt))
        return self._optimizer_for_vars[key]

    # ------------------------------------------------------------------ fitting
    def fit(self, dataset, nb_epoch: int = 10, max_checkpoints_to_keep: int = 5,
            checkpoint_interval: int = 1000, deterministic: bool = False, restore: bool = False,
            variables=None, loss=None, callbacks: Union[Callable, List[Callable]] = [],
            all_losses: Optional[List[float]] = None) -> float:
        batches = self._batch_generator(dataset, epochs=nb_epoch, deterministic=deterministic)
        return self.fit_generator(batches, max_checkpoints_to_keep, checkpoint_interval, restore, variables,
                                  loss, callbacks, all_losses)

    def fit_generator(self, generator: Iterable[Tuple[Any, Any, Any]],
                      max_checkpoints_to_keep: int = 5, checkpoint_interval: int = 1000,
                      restore: bool = False, variables=None, loss=None,
                      callbacks: Union[Callable, List[Callable]] = [],
                      all_losses: Optional[List[float]] = None) -> float:
        self._ensure_built()
        self.model.train()
        run = _FitRun(self, variables, loss, callbacks, max_checkpoints_to_keep, checkpoint_interval, all_losses)
        started = time.time()
        pending_restore = restore
        for batch in generator:
            if pending_restore:  # after the first batch exists, as the reference does (lazily built models)
                self.restore()
                pending_restore = False
            inputs, labels, weights = self._prepare_batch(batch)
            run.after_step(self._train_step(self._unwrap_single(inputs), labels, weights, run.loss, run.optimizer))
        mean_loss = run.finish()
        logger.info("TIMING: model fitting took %0.3f s" % (time.time() - started))
        return mean_loss

    @staticmethod
    def _unwrap_single(inputs):
        return inputs[0] if isinstance(inputs, list) and len(inputs) == 1 else inputs

    def _forward_lists(self, inputs) -> List[torch.Tensor]:
        out = self.model(inputs)
        return [out] if isinstance(out, torch.Tensor) else list(out)

    def _train_step(self, inputs, labels, weights, loss, optimizer):
        """One optimizer step on one prepared batch (torch_model.py:435-443): zero_grad, forward,
        loss over the loss outputs, backward, gradient all-reduce on data-parallel ranks, step.
        Subclasses replace this by a fused native step."""
        arena = getattr(self, "_grad_arena", None)
        if arena is not None and self._grad_sync is None and not getattr(self, "_flat_step", False):
            arena = None
        if arena is not None and not arena.covers(self.model):
            arena = None  # somebody replaced parameters since the arena was built: the per-tensor paths handle any set
        if arena is not None:
            arena.attach()  # zeroed views of one flat buffer behind every p.grad (deepchem_amd.dist.FlatGradArena)
        else:
            optimizer.zero_grad()
        outputs = self._forward_lists(inputs)
        if self._roles.declared:
            outputs = [outputs[i] for i in self._roles.loss]
        batch_loss = loss(outputs, labels, weights)
        from deepchem_amd.ops import direct_param_grads
        with direct_param_grads(arena is not None):  # weight gradients land in the arena as the kernels produce them
            batch_loss.backward()
        intact = arena is not None and arena.intact()
        if self._grad_sync is not None:
            if intact and hasattr(self._grad_sync, "reduce_flat"):
                self._grad_sync.reduce_flat(arena.flat)  # ONE zero-copy collective
            else:
                self._grad_sync(self.model)
        flat = getattr(optimizer, "_flat", None)
        if intact and flat is not None and flat.get("g") is arena.flat and arena.params_homed():
            optimizer.step_flat(0, arena.flat.numel())  # ONE launch over every parameter
        else:
            optimizer.step()
        return batch_loss

    def fit_on_batch(self, X: Sequence, y: Sequence, w: Sequence, variables=None, loss=None,
                     callbacks: Union[Callable, List[Callable]] = [], checkpoint: bool = True,
                     max_checkpoints_to_keep: int = 5) -> float:
        self._ensure_built()
        # an interval that the single step of this call hits exactly when a checkpoint is wanted
        interval = self._global_step + 2 if checkpoint else 0
        return self.fit(NumpyDataset(X, y, w), nb_epoch=1, max_checkpoints_to_keep=max_checkpoints_to_keep,
                        checkpoint_interval=interval, variables=variables, loss=loss, callbacks=callbacks)

    # ------------------------------------------------------------------ prediction
    def _predict(self, generator: Iterable[Tuple[Any, Any, Any]], transformers: List,
                 uncertainty: bool, other_output_types):
        _OutputSink.check(self._roles, uncertainty, other_output_types)
        sink = _OutputSink(self._roles, transformers, uncertainty, other_output_types)
        self._ensure_built()
        self.model.eval()
        with torch.no_grad():
            for inputs, _labels, _weights in generator:
                prepared, _, _ = self._prepare_batch((inputs, None, None))
                sink.push(self._forward_lists(self._unwrap_single(prepared)))
        return sink.result()

    def predict_on_generator(self, generator, transformers: List = [], output_types=None):
        return self._predict(generator, transformers, False, output_types)

    def predict_on_batch(self, X, transformers: List = []):
        return self.predict(NumpyDataset(X=X, y=None), transformers)

    def predict_uncertainty_on_batch(self, X: Sequence, masks: int = 50):
        return self.predict_uncertainty(NumpyDataset(X=X, y=None), masks)

    def predict(self, dataset, transformers: List = [], output_types: Optional[List[str]] = None):
        batches = self._batch_generator(dataset, mode='predict', pad_batches=False)
        return self.predict_on_generator(batches, transformers=transformers, output_types=output_types)

    def predict_embedding(self, dataset):
        batches = self._batch_generator(dataset, mode='predict', pad_batches=False)
        return self._predict(batches, [], False, ['embedding'])

    def predict_uncertainty(self, dataset, masks: int = 50):
        """Mean prediction and total standard deviation over ``masks`` stochastic passes: aleatoric
        (mean predicted variance) + epistemic (spread of the predictions) (torch_model.py:784-839)."""
        moments = None  # per output: [sum p, sum p^2, sum var]
        for _ in range(masks):
            batches = self._batch_generator(dataset, mode='uncertainty', pad_batches=False)
            passes = list(self._predict(batches, [], True, None))
            if moments is None:
                moments = [[np.zeros_like(p), np.zeros_like(p), np.zeros_like(v)] for p, v in passes]
            for acc, (p, v) in zip(moments, passes):
                acc[0] += p
                acc[1] += p * p
                acc[2] += v
        pairs = []
        for s1, s2, sv in moments or []:
            mean = s1 / masks
            pairs.append((mean, np.sqrt(s2 / masks - mean * mean + sv / masks)))
        return pairs[0] if len(pairs) == 1 else pairs

    # ------------------------------------------------------------------ batches
    def _to_device(self, x):
        if torch.is_tensor(x):
            return x if (x.dim() == 0 and not x.is_floating_point()) else x.to(self.device)
        x = np.asarray(x)
        if x.dtype == np.float64:
            x = x.astype(np.float32)
        if x.ndim == 0 and x.dtype.kind in "iu":
            return torch.as_tensor(x)  # sizes stay on the host: trimming never synchronises the stream
        return torch.as_tensor(x, device=self.device)

    def _prepare_batch(self, batch: Tuple[Any, Any, Any]):
        """float64 -> float32, everything onto the device (torch_model.py:923-952); a ``DeviceBatch``
        (already collated in HBM) passes through."""
        inputs, labels, weights = batch
        moved = inputs if hasattr(inputs, "graph") else [self._to_device(x) for x in inputs]
        return (moved, [self._to_device(x) for x in labels or ()], [self._to_device(x) for x in weights or ()])

    def _batch_generator(self, dataset, epochs: int = 1, mode: str = 'fit', deterministic: bool = True,
                         pad_batches: bool = True):
        """What fit()/predict*() iterate; subclasses may substitute a faster equivalent of
        ``default_generator``."""
        return self.default_generator(dataset, epochs=epochs, mode=mode, deterministic=deterministic,
                                      pad_batches=pad_batches)

    def default_generator(self, dataset, epochs: int = 1, mode: str = 'fit',
                          deterministic: bool = True, pad_batches: bool = True):
        for _ in range(epochs):
            for X_b, y_b, w_b, _ids in dataset.iterbatches(batch_size=self.batch_size, deterministic=deterministic,
                                                           pad_batches=pad_batches):
                yield ([X_b], [y_b], [w_b])

    # ------------------------------------------------------------------ checkpoints
    @staticmethod
    def _slot(model_dir: str, n: int) -> str:
        return os.path.join(model_dir, 'checkpoint%d.pt' % n)

    def save_checkpoint(self, max_checkpoints_to_keep: int = 5, model_dir: Optional[str] = None) -> None:
        """checkpoint1.pt is always the newest; older ones shift up by one and the oldest beyond
        ``max_checkpoints_to_keep`` is dropped (torch_model.py:996-1042)."""
        if max_checkpoints_to_keep == 0:
            return
        self._ensure_built()
        model_dir = model_dir or self.model_dir
        os.makedirs(model_dir, exist_ok=True)
        staged = os.path.join(model_dir, 'temp_checkpoint.pt')
        torch.save({'model_state_dict': self.model.state_dict(),
                    'optimizer_state_dict': self._pytorch_optimizer.state_dict(),
                    'global_step': self._global_step}, staged)
        oldest = self._slot(model_dir, max_checkpoints_to_keep)
        if os.path.exists(oldest):
            os.remove(oldest)
        for n in range(max_checkpoints_to_keep - 1, 0, -1):
            if os.path.exists(self._slot(model_dir, n)):
                os.rename(self._slot(model_dir, n), self._slot(model_dir, n + 1))
        os.rename(staged, self._slot(model_dir, 1))

    def get_checkpoints(self, model_dir: Optional[str] = None):
        model_dir = model_dir or self.model_dir
        return [os.path.join(model_dir, f) for f in sorted(os.listdir(model_dir))
                if f.startswith('checkpoint') and f.endswith('.pt')]

    def restore(self, checkpoint: Optional[str] = None, model_dir: Optional[str] = None,
                strict: Optional[bool] = True) -> None:
        logger.info('Restoring model')
        self._ensure_built()
        if checkpoint is None:
            found = sorted(self.get_checkpoints(model_dir))
            if not found:
                raise ValueError('No checkpoint found')
            checkpoint = found[0]  # checkpoint1.pt: the newest
        data = torch.load(checkpoint, map_location=self.device)
        self.model.load_state_dict(data['model_state_dict'], strict=strict)
        self._pytorch_optimizer.load_state_dict(data['optimizer_state_dict'])
        self._global_step = data['global_step']

    def get_global_step(self) -> int:
        return self._global_step

    # ------------------------------------------------------------------ transfer
    def _create_assignment_map(self, source_model: "TorchModel", include_top: bool = True, **kwargs):
        """source parameter -> own parameter, pairwise in ``parameters()`` order; without the top
        the last weight + bias pair is left out (torch_model.py:1134-1170)."""
        pairs = list(zip(source_model.model.parameters(), self.model.parameters()))
        return dict(pairs if include_top else pairs[:-2])

    def _create_value_map(self, source_model: "TorchModel", **kwargs):
        return {p: p.detach().cpu().numpy() for p in source_model.model.parameters()}

    def load_from_pretrained(self, source_model: "TorchModel", assignment_map=None, value_map=None,
                             checkpoint: Optional[str] = None, model_dir: Optional[str] = None,
                             include_top: bool = True, inputs=None, **kwargs) -> None:
        if inputs is not None:  # lazily built modules get their shapes from one forward each
            source_model.model(inputs)
            self.model(inputs)
        self._ensure_built()
        if value_map is None:
            source_model.restore(model_dir=model_dir, checkpoint=checkpoint)
            value_map = self._create_value_map(source_model=source_model)
        if assignment_map is None:
            assignment_map = self._create_assignment_map(source_model=source_model, include_top=include_top)
        for src, dst in assignment_map.items():
            assert src.shape == dst.shape
            dst.data = torch.as_tensor(value_map[src], device=self.device)


class _StandardLoss(object):
    """``mean(w * criterion(outputs, labels))`` [+ regularization] (torch_model.py:1267-1294).
    L2Loss / SoftmaxCrossEntropy on GPU tensors run criterion, weighting and mean as ONE HIP kernel."""

    def __init__(self, model: TorchModel, loss: Loss) -> None:
        self.model = model
        self.loss = loss
        self.criterion = loss._create_pytorch_loss()

    def __call__(self, outputs: List, labels: List, weights: List):
        if len(outputs) != 1 or len(labels) != 1 or len(weights) != 1:
            raise ValueError("Loss functions expects exactly one each of outputs, labels, and weights")
        out, lab, w = outputs[0], labels[0], weights[0]
        kind = getattr(self.loss, "_gcmi_kind", None)
        if kind is not None and out.is_cuda and self._fusable(kind, out, lab, w):
            value = ops.StandardLossFn.apply(out, lab, w, kind)
        else:
            per_element = self.criterion(out, lab)
            missing = per_element.dim() - w.dim()
            if missing > 0:  # weights broadcast over trailing axes
                w = w.reshape(tuple(w.shape) + (1,) * missing)
            value = (per_element * w).mean()
        reg = self.model.regularization_loss
        return value if reg is None else value + reg()

    @staticmethod
    def _fusable(kind, out, lab, w) -> bool:
        if out.dtype != torch.float32 or not torch.is_tensor(w) or not w.is_cuda:
            return False
        if kind == 0:  # (rows, tasks, classes) logits with one weight per (row, task)
            return out.dim() == 3 and tuple(lab.shape) == tuple(out.shape) and \
                w.numel() == out.shape[0] * out.shape[1]
        return out.dim() == 2 and lab.numel() == out.numel() and w.numel() == out.numel()

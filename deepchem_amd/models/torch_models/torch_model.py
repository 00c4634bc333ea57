"""``TorchModel``: the training / prediction / checkpoint loop the GraphConv
model runs under.  A restatement of the parts of
deepchem/models/torch_models/torch_model.py the path uses -- ``fit`` :289,
``fit_generator`` :345-496, ``_predict`` :547-652, ``predict`` :731,
``predict_embedding`` :763, ``predict_uncertainty`` :784, ``_prepare_batch``
:923-952, ``save_checkpoint`` :996-1042, ``restore`` :1061-1090,
``load_from_pretrained`` :1196-1264, ``_StandardLoss`` :1267-1294 -- with the
same argument meaning, return values and errors.  TensorBoard / W&B hooks and
``compile`` are not part of the path and are absent.

MI355X-side changes (none alters results):
* ``_StandardLoss`` hands criterion + weighting + mean to one fused HIP kernel
  when the loss is L2Loss / SoftmaxCrossEntropy and the tensors are on the GPU;
* the default optimizer steps on the HIP Adam kernel;
* 0-dim integer inputs (``n_samples``) stay on the host, so trimming the
  output never synchronises the stream.
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

logger = logging.getLogger(__name__)


class TorchModel(Model):

    def __init__(self, model: torch.nn.Module, loss, output_types: Optional[List[str]] = None,
                 batch_size: int = 100, model_dir: Optional[str] = None,
                 learning_rate: Union[float, LearningRateSchedule] = 0.001,
                 optimizer: Optional[Optimizer] = None, log_frequency: int = 100,
                 device: Optional[torch.device] = None,
                 regularization_loss: Optional[Callable] = None, **kwargs) -> None:
        super(TorchModel, self).__init__(model=model, model_dir=model_dir, **kwargs)
        self.loss = loss
        self.learning_rate = learning_rate
        self.output_types = output_types
        if isinstance(loss, Loss):
            self._loss_fn = _StandardLoss(self, loss)
        else:
            self._loss_fn = loss
        self.batch_size = batch_size
        self.optimizer = Adam(learning_rate=learning_rate) if optimizer is None else optimizer
        self.regularization_loss = regularization_loss
        if device is None:
            device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = device
        self.model = model.to(device)
        self.log_frequency = log_frequency
        if output_types is None:
            self._prediction_outputs = None
            self._loss_outputs = None
            self._variance_outputs = None
            self._other_outputs = None
        else:
            self._prediction_outputs = []
            self._loss_outputs = []
            self._variance_outputs = []
            self._other_outputs = []
            for i, type_ in enumerate(output_types):
                if type_ == 'prediction':
                    self._prediction_outputs.append(i)
                elif type_ == 'loss':
                    self._loss_outputs.append(i)
                elif type_ == 'variance':
                    self._variance_outputs.append(i)
                else:
                    self._other_outputs.append(i)
            if len(self._loss_outputs) == 0:
                self._loss_outputs = self._prediction_outputs
        self._built = False
        self._optimizer_for_vars: Dict[Any, Any] = {}
        # set by deepchem_amd.dist.shard_model(); None = single process
        self._grad_sync: Optional[Callable] = None

    def _ensure_built(self) -> None:
        if self._built:
            return
        self._built = True
        self._global_step = 0
        self._pytorch_optimizer = self.optimizer._create_pytorch_optimizer(self.model.parameters())
        if isinstance(self.optimizer.learning_rate, LearningRateSchedule):
            self._lr_schedule = self.optimizer.learning_rate._create_pytorch_schedule(
                self._pytorch_optimizer)
        else:
            self._lr_schedule = None

    # ------------------------------------------------------------------ fitting
    def fit(self, dataset, nb_epoch: int = 10, max_checkpoints_to_keep: int = 5,
            checkpoint_interval: int = 1000, deterministic: bool = False, restore: bool = False,
            variables=None, loss=None, callbacks: Union[Callable, List[Callable]] = [],
            all_losses: Optional[List[float]] = None) -> float:
        return self.fit_generator(
            self._batch_generator(dataset, epochs=nb_epoch, deterministic=deterministic),
            max_checkpoints_to_keep, checkpoint_interval, restore, variables, loss, callbacks,
            all_losses)

    def fit_generator(self, generator: Iterable[Tuple[Any, Any, Any]],
                      max_checkpoints_to_keep: int = 5, checkpoint_interval: int = 1000,
                      restore: bool = False, variables=None, loss=None,
                      callbacks: Union[Callable, List[Callable]] = [],
                      all_losses: Optional[List[float]] = None) -> float:
        if not isinstance(callbacks, SequenceCollection):
            callbacks = [callbacks]
        self._ensure_built()
        self.model.train()
        avg_loss = 0.0
        last_avg_loss = 0.0
        averaged_batches = 0
        if loss is None:
            loss = self._loss_fn
        if variables is None:
            optimizer = self._pytorch_optimizer
            lr_schedule = self._lr_schedule
        else:
            variables_tuple = tuple(variables)
            if variables_tuple in self._optimizer_for_vars:
                optimizer, lr_schedule = self._optimizer_for_vars[variables_tuple]
            else:
                optimizer = self.optimizer._create_pytorch_optimizer(variables_tuple)
                if isinstance(self.optimizer.learning_rate, LearningRateSchedule):
                    lr_schedule = self.optimizer.learning_rate._create_pytorch_schedule(optimizer)
                else:
                    lr_schedule = None
                self._optimizer_for_vars[variables_tuple] = (optimizer, lr_schedule)
        time1 = time.time()
        current_step = self._global_step

        for batch in generator:
            if restore:
                self.restore()
                restore = False
            inputs, labels, weights = self._prepare_batch(batch)
            if isinstance(inputs, list) and len(inputs) == 1:
                inputs = inputs[0]
            batch_loss = self._train_step(inputs, labels, weights, loss, optimizer)
            if lr_schedule is not None:
                lr_schedule.step()
            self._global_step += 1
            current_step = self._global_step

            avg_loss = avg_loss + batch_loss.detach()  # stays on the device: no sync per step
            averaged_batches += 1
            should_log = (current_step % self.log_frequency == 0)
            if should_log:
                avg_loss = float(avg_loss) / averaged_batches
                logger.info('Ending global_step %d: Average loss %g' % (current_step, avg_loss))
                if all_losses is not None:
                    all_losses.append(avg_loss)
                last_avg_loss = avg_loss
                avg_loss = 0.0
                averaged_batches = 0
            if checkpoint_interval > 0 and current_step % checkpoint_interval == checkpoint_interval - 1:
                self.save_checkpoint(max_checkpoints_to_keep)
            for c in callbacks:
                try:
                    c(self, current_step, iteration_loss=batch_loss)
                except TypeError:
                    c(self, current_step)

        if averaged_batches > 0:
            avg_loss = float(avg_loss) / averaged_batches
            logger.info('Ending global_step %d: Average loss %g' % (current_step, avg_loss))
            if all_losses is not None:
                all_losses.append(avg_loss)
            last_avg_loss = avg_loss
        if checkpoint_interval > 0:
            self.save_checkpoint(max_checkpoints_to_keep)
        time2 = time.time()
        logger.info("TIMING: model fitting took %0.3f s" % (time2 - time1))
        return last_avg_loss

    def _train_step(self, inputs, labels, weights, loss, optimizer):
        """zero_grad, forward, loss, backward, (gradient all-reduce), optimizer step
        (torch_model.py:435-443).  Subclasses may replace it by a fused native step."""
        optimizer.zero_grad()
        outputs = self.model(inputs)
        if isinstance(outputs, torch.Tensor):
            outputs = [outputs]
        if self._loss_outputs is not None:
            outputs = [outputs[i] for i in self._loss_outputs]
        batch_loss = loss(outputs, labels, weights)
        batch_loss.backward()
        if self._grad_sync is not None:
            self._grad_sync(self.model)  # data-parallel ranks: one flat all-reduce per step
        optimizer.step()
        return batch_loss

    def fit_on_batch(self, X: Sequence, y: Sequence, w: Sequence, variables=None, loss=None,
                     callbacks: Union[Callable, List[Callable]] = [], checkpoint: bool = True,
                     max_checkpoints_to_keep: int = 5) -> float:
        self._ensure_built()
        dataset = NumpyDataset(X, y, w)
        return self.fit(dataset, nb_epoch=1, max_checkpoints_to_keep=max_checkpoints_to_keep,
                        checkpoint_interval=self._global_step + 2 if checkpoint else 0,
                        variables=variables, loss=loss, callbacks=callbacks)

    # ------------------------------------------------------------------ prediction
    def _predict(self, generator: Iterable[Tuple[Any, Any, Any]], transformers: List,
                 uncertainty: bool, other_output_types):
        results: Optional[List[List[np.ndarray]]] = None
        variances: Optional[List[List[np.ndarray]]] = None
        if uncertainty and (other_output_types is not None):
            raise ValueError(
                'This model cannot compute uncertainties and other output types simultaneously. Please invoke one at a time.'
            )
        if uncertainty:
            if self._variance_outputs is None or len(self._variance_outputs) == 0:
                raise ValueError('This model cannot compute uncertainties')
            if len(self._variance_outputs) != len(self._prediction_outputs):
                raise ValueError('The number of variances must exactly match the number of outputs')
        if other_output_types:
            if self._other_outputs is None or len(self._other_outputs) == 0:
                raise ValueError(
                    'This model cannot compute other outputs since no other output_types were specified.'
                )
        if len(transformers) > 0:
            raise NotImplementedError("undo_transforms is outside the GraphConv hot path")
        self._ensure_built()
        self.model.eval()
        for batch in generator:
            inputs, labels, weights = batch
            inputs, _, _ = self._prepare_batch((inputs, None, None))
            if isinstance(inputs, list) and len(inputs) == 1:
                inputs = inputs[0]
            with torch.no_grad():
                output_values = self.model(inputs)
            if isinstance(output_values, torch.Tensor):
                output_values = [output_values]
            output_values = [t.detach().cpu().numpy() for t in output_values]
            if uncertainty:
                var = [output_values[i] for i in self._variance_outputs]
                if variances is None:
                    variances = [var]
                else:
                    for i, t in enumerate(var):
                        variances[i].append(t)
            access_values = []
            if other_output_types:
                access_values += self._other_outputs
            elif self._prediction_outputs is not None:
                access_values += self._prediction_outputs
            if len(access_values) > 0:
                output_values = [output_values[i] for i in access_values]
            if results is None:
                results = [[] for i in range(len(output_values))]
            for i, t in enumerate(output_values):
                results[i].append(t)

        final_results = []
        final_variances = []
        if results is not None:
            for r in results:
                final_results.append(np.concatenate(r, axis=0))
        if uncertainty and variances is not None:
            for v in variances:
                final_variances.append(np.concatenate(v, axis=0))
            return zip(final_results, final_variances)
        if len(final_results) == 1:
            return final_results[0]
        return final_results

    def predict_on_generator(self, generator, transformers: List = [], output_types=None):
        return self._predict(generator, transformers, False, output_types)

    def predict_on_batch(self, X, transformers: List = []):
        dataset = NumpyDataset(X=X, y=None)
        return self.predict(dataset, transformers)

    def predict_uncertainty_on_batch(self, X: Sequence, masks: int = 50):
        dataset = NumpyDataset(X=X, y=None)
        return self.predict_uncertainty(dataset, masks)

    def predict(self, dataset, transformers: List = [], output_types: Optional[List[str]] = None):
        generator = self._batch_generator(dataset, mode='predict', pad_batches=False)
        return self.predict_on_generator(generator, transformers=transformers,
                                         output_types=output_types)

    def predict_embedding(self, dataset):
        generator = self._batch_generator(dataset, mode='predict', pad_batches=False)
        return self._predict(generator, [], False, ['embedding'])

    def predict_uncertainty(self, dataset, masks: int = 50):
        sum_pred: List[np.ndarray] = []
        sum_sq_pred: List[np.ndarray] = []
        sum_var: List[np.ndarray] = []
        for i in range(masks):
            generator = self._batch_generator(dataset, mode='uncertainty', pad_batches=False)
            results = self._predict(generator, [], True, None)
            if len(sum_pred) == 0:
                for p, v in results:
                    sum_pred.append(p)
                    sum_sq_pred.append(p * p)
                    sum_var.append(v)
            else:
                for j, (p, v) in enumerate(results):
                    sum_pred[j] += p
                    sum_sq_pred[j] += p * p
                    sum_var[j] += v
        output = []
        std = []
        for i in range(len(sum_pred)):
            p = sum_pred[i] / masks
            output.append(p)
            std.append(np.sqrt(sum_sq_pred[i] / masks - p * p + sum_var[i] / masks))
        if len(output) == 1:
            return (output[0], std[0])
        return list(zip(output, std))

    # ------------------------------------------------------------------ batches
    def _to_device(self, x):
        if torch.is_tensor(x):
            return x if (x.dim() == 0 and not x.is_floating_point()) else x.to(self.device)
        x = np.asarray(x)
        if x.dtype == np.float64:
            x = x.astype(np.float32)
        if x.ndim == 0 and x.dtype.kind in "iu":
            return torch.as_tensor(x)  # sizes stay on the host
        return torch.as_tensor(x, device=self.device)

    def _prepare_batch(self, batch: Tuple[Any, Any, Any]):
        inputs, labels, weights = batch
        if hasattr(inputs, "graph"):  # a DeviceBatch: already collated and resident in HBM
            input_tensors = inputs
        else:
            input_tensors = [self._to_device(x) for x in inputs]
        label_tensors = [self._to_device(x) for x in labels] if labels is not None else []
        weight_tensors = [self._to_device(x) for x in weights] if weights is not None else []
        return (input_tensors, label_tensors, weight_tensors)

    def _batch_generator(self, dataset, epochs: int = 1, mode: str = 'fit', deterministic: bool = True,
                         pad_batches: bool = True):
        """The generator fit()/predict*() use; subclasses may substitute a faster equivalent of
        ``default_generator``."""
        return self.default_generator(dataset, epochs=epochs, mode=mode, deterministic=deterministic,
                                      pad_batches=pad_batches)

    def default_generator(self, dataset, epochs: int = 1, mode: str = 'fit',
                          deterministic: bool = True, pad_batches: bool = True):
        for epoch in range(epochs):
            for (X_b, y_b, w_b, ids_b) in dataset.iterbatches(batch_size=self.batch_size,
                                                              deterministic=deterministic,
                                                              pad_batches=pad_batches):
                yield ([X_b], [y_b], [w_b])

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, max_checkpoints_to_keep: int = 5, model_dir: Optional[str] = None) -> None:
        if max_checkpoints_to_keep == 0:
            return
        self._ensure_built()
        if model_dir is None:
            model_dir = self.model_dir
        if not os.path.exists(model_dir):
            os.makedirs(model_dir)
        data = {
            'model_state_dict': self.model.state_dict(),
            'optimizer_state_dict': self._pytorch_optimizer.state_dict(),
            'global_step': self._global_step
        }
        temp_file = os.path.join(model_dir, 'temp_checkpoint.pt')
        torch.save(data, temp_file)
        paths = [os.path.join(model_dir, 'checkpoint%d.pt' % (i + 1)) for i in range(max_checkpoints_to_keep)]
        if os.path.exists(paths[-1]):
            os.remove(paths[-1])
        for i in reversed(range(max_checkpoints_to_keep - 1)):
            if os.path.exists(paths[i]):
                os.rename(paths[i], paths[i + 1])
        os.rename(temp_file, paths[0])

    def get_checkpoints(self, model_dir: Optional[str] = None):
        if model_dir is None:
            model_dir = self.model_dir
        files = sorted(os.listdir(model_dir))
        files = [f for f in files if f.startswith('checkpoint') and f.endswith('.pt')]
        return [os.path.join(model_dir, f) for f in files]

    def restore(self, checkpoint: Optional[str] = None, model_dir: Optional[str] = None,
                strict: Optional[bool] = True) -> None:
        logger.info('Restoring model')
        self._ensure_built()
        if checkpoint is None:
            checkpoints = sorted(self.get_checkpoints(model_dir))
            if len(checkpoints) == 0:
                raise ValueError('No checkpoint found')
            checkpoint = checkpoints[0]
        data = torch.load(checkpoint, map_location=self.device)
        self.model.load_state_dict(data['model_state_dict'], strict=strict)
        self._pytorch_optimizer.load_state_dict(data['optimizer_state_dict'])
        self._global_step = data['global_step']

    def get_global_step(self) -> int:
        return self._global_step

    # ------------------------------------------------------------------ transfer
    def _create_assignment_map(self, source_model: "TorchModel", include_top: bool = True, **kwargs):
        assignment_map: Dict[Any, Any] = {}
        source_vars = list(source_model.model.parameters())
        dest_vars = list(self.model.parameters())
        if not include_top:
            source_vars = source_vars[:-2]
            dest_vars = dest_vars[:-2]
        for source_var, dest_var in zip(source_vars, dest_vars):
            assignment_map[source_var] = dest_var
        return assignment_map

    def _create_value_map(self, source_model: "TorchModel", **kwargs):
        return {v: v.detach().cpu().numpy() for v in source_model.model.parameters()}

    def load_from_pretrained(self, source_model: "TorchModel", assignment_map=None, value_map=None,
                             checkpoint: Optional[str] = None, model_dir: Optional[str] = None,
                             include_top: bool = True, inputs=None, **kwargs) -> None:
        if inputs is not None:
            source_model.model(inputs)
            self.model(inputs)
        self._ensure_built()
        if value_map is None:
            source_model.restore(model_dir=model_dir, checkpoint=checkpoint)
            value_map = self._create_value_map(source_model=source_model)
        if assignment_map is None:
            assignment_map = self._create_assignment_map(source_model=source_model,
                                                         include_top=include_top)
        for source_var, dest_var in assignment_map.items():
            assert source_var.shape == dest_var.shape
            dest_var.data = torch.as_tensor(value_map[source_var], device=self.device)


class _StandardLoss(object):
    """mean(w * criterion(outputs, labels)) [+ regularization]
    (torch_model.py:1267-1294)."""

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
            loss = ops.StandardLossFn.apply(out, lab, w, kind)
        else:
            losses = self.criterion(out, lab)
            if len(w.shape) < len(losses.shape):
                shape = tuple(w.shape)
                w = w.reshape(shape + (1,) * (len(losses.shape) - len(w.shape)))
            loss = (losses * w).mean()
        if self.model.regularization_loss is not None:
            loss = loss + self.model.regularization_loss()
        return loss

    @staticmethod
    def _fusable(kind, out, lab, w) -> bool:
        if out.dtype != torch.float32 or not torch.is_tensor(w) or not w.is_cuda:
            return False
        if kind == 0:  # (rows, tasks, classes) logits with one weight per (row, task)
            return out.dim() == 3 and tuple(lab.shape) == tuple(out.shape) and \
                w.numel() == out.shape[0] * out.shape[1]
        return out.dim() == 2 and lab.numel() == out.numel() and w.numel() == out.numel()

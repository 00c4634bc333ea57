"""Optimizer front-ends with the reference's interface
(deepchem/models/optimizers.py: ``Optimizer`` :13, ``Adam`` :190-241).

``Adam._create_pytorch_optimizer`` returns a ``torch.optim.Optimizer`` whose
``step()`` runs the HIP Adam kernel (gcmi_adam_step) on every CUDA parameter that
has a gradient.  The
state layout (``step``, ``exp_avg``, ``exp_avg_sq``) and ``state_dict()`` are
those of ``torch.optim.Adam``, so checkpoints are interchangeable.
"""
from typing import Dict, Union

import torch

from deepchem_amd import ops


class LearningRateSchedule(object):
    """Marker base class (the GraphConv path uses constant rates)."""

    def _create_pytorch_schedule(self, optimizer):
        raise NotImplementedError("Subclasses must implement this")


class Optimizer(object):

    def __init__(self, learning_rate: Union[float, LearningRateSchedule]):
        self.learning_rate = learning_rate

    def _create_pytorch_optimizer(self, params):
        raise NotImplementedError("Subclasses must implement this")


class GcmiAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad, no weight decay) on the HIP kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise ValueError("weight_decay is not supported by the HIP Adam kernel")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False,
                        fused=None)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue  # torch skips parameters without a gradient
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                if not p.is_cuda:
                    raise RuntimeError("GcmiAdam: parameters must be CUDA tensors")
                ops.adam_step_(p.data, p.grad.data if p.grad.is_contiguous() else p.grad.contiguous(),
                               state["exp_avg"], state["exp_avg_sq"], group["lr"], beta1, beta2,
                               group["eps"], int(state["step"].item()))
        return loss


class Adam(Optimizer):
    """Adam (deepchem/models/optimizers.py:190-241)."""

    def __init__(self, learning_rate: Union[float, LearningRateSchedule] = 0.001,
                 beta1: float = 0.9, beta2: float = 0.999, epsilon: float = 1e-08,
                 weight_decay: float = 0):
        super(Adam, self).__init__(learning_rate)
        self.beta1 = beta1
        self.beta2 = beta2
        self.epsilon = epsilon
        self.weight_decay = weight_decay

    def _create_pytorch_optimizer(self, params):
        if isinstance(self.learning_rate, LearningRateSchedule):
            lr = self.learning_rate.initial_rate
        else:
            lr = self.learning_rate
        params = list(params)
        if params and all(p.is_cuda for p in params) and self.weight_decay == 0:
            return GcmiAdam(params, lr=lr, betas=(self.beta1, self.beta2), eps=self.epsilon)
        # host-side / exotic configurations: torch's own optimizer
        return torch.optim.Adam(params, lr=lr, betas=(self.beta1, self.beta2), eps=self.epsilon,
                                weight_decay=self.weight_decay)

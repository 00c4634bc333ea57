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
        self._flat = None

    def attach_flat(self, param_flat: torch.Tensor, grad_flat: torch.Tensor, slices):
        """All parameters (in order) are views of ``param_flat`` and their gradients views of
        ``grad_flat`` (``slices``: (offset, numel) per parameter): ``step_flat`` then updates any
        contiguous range with ONE launch.  exp_avg / exp_avg_sq become views of two flat buffers,
        so ``state_dict()`` keeps torch.optim.Adam's per-parameter layout."""
        params = [p for g in self.param_groups for p in g["params"]]
        if len(params) != len(slices):
            raise ValueError("attach_flat: %d parameters, %d slices" % (len(params), len(slices)))
        m = torch.zeros_like(param_flat)
        v = torch.zeros_like(param_flat)
        # carry over any state that already exists (restore() before the first native step)
        for p, (off, n) in zip(params, slices):
            st = self.state.get(p)
            if st:
                m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        self._flat = dict(p=param_flat, g=grad_flat, m=m, v=v, slices=list(slices), params=params)

    def _setup_flat_range(self, lo: int, hi: int):
        """State entries (views of the flat moments) for every parameter inside [lo, hi); all of
        them share ONE step tensor, so a step costs one increment instead of a loop."""
        f = self._flat
        step_t = None
        inside = []
        for p, (off, n) in zip(f["params"], f["slices"]):
            if lo <= off and off + n <= hi:
                st = self.state[p]
                if "step" in st:
                    s = float(st["step"])
                    if step_t is None:
                        step_t = torch.tensor(s, dtype=torch.float32)
                    elif float(step_t) != s:
                        raise RuntimeError("parameters of one flat range have different Adam step counts")
                inside.append((p, off, n))
        if step_t is None:
            step_t = torch.tensor(0.0, dtype=torch.float32)
        for p, off, n in inside:
            st = self.state[p]
            st["step"] = step_t
            st["exp_avg"] = f["m"][off:off + n].view(p.shape)
            st["exp_avg_sq"] = f["v"][off:off + n].view(p.shape)
        f["range"] = (lo, hi)
        f["step_t"] = step_t
        f["n_inside"] = len(inside)

    @torch.no_grad()
    def step_flat(self, lo: int, hi: int):
        """Adam on the flat range [lo, hi) (floats): one kernel launch."""
        f = self._flat
        if f.get("range") != (lo, hi):
            self._setup_flat_range(lo, hi)
        if f["n_inside"] == 0:
            return
        group = self.param_groups[0]
        beta1, beta2 = group["betas"]
        f["step_t"] += 1
        ops.adam_step_(f["p"][lo:hi], f["g"][lo:hi], f["m"][lo:hi], f["v"][lo:hi], group["lr"], beta1,
                       beta2, group["eps"], int(f["step_t"].item()))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._flat is not None:  # re-home the loaded moments into the flat buffers
            f = self._flat
            for p, (off, n) in zip(f["params"], f["slices"]):
                st = self.state.get(p)
                if st and "exp_avg" in st:
                    f["m"][off:off + n].copy_(st["exp_avg"].reshape(-1))
                    f["v"][off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    st["exp_avg"] = f["m"][off:off + n].view(p.shape)
                    st["exp_avg_sq"] = f["v"][off:off + n].view(p.shape)
                    st["step"] = torch.as_tensor(st["step"], dtype=torch.float32).cpu()
            f.pop("range", None)  # re-derive the shared step tensor on the next flat step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # after a flat (native) step the parameters of the trained range share ONE step tensor: it advances once
        # per call here, not once per parameter that points at it
        bumped = set()
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
                if id(state["step"]) not in bumped:
                    bumped.add(id(state["step"]))
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

"""Data-parallel sharding over the GPUs of one node.

Molecules are independent graphs, so a global batch shards by molecule with no
data-path exchange at all: every rank collates and runs its own molecules.  The
one collective per step is a sum all-reduce (RCCL over xGMI; ``gloo`` on CPU in
the tests) of ONE flat fp32 bucket holding every gradient that exists --
head + dense + BatchNorm[1..2] (15 k floats for Tox21) in ``grad_mode="reference"``,
all 204 k floats in ``"full"``.  At <= 1 MB the ring is latency-bound on the
7 x 153 GB/s links, so a single fused call beats per-tensor calls; the result
is scaled by 1/world_size, which equals the single-process mean over the global
batch when every rank holds the same number of molecules
(loss = mean over (B, T), torch_model.py:1290-1291).  BatchNorm statistics stay
per rank, as torch DDP would leave them.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one (the same
    partition deepchem/data/pytorch_datasets.py:104-113 applies to disk shards)."""
    lo = n_items * rank // world_size
    hi = n_items * (rank + 1) // world_size
    return lo, hi


def shard_indices(indices: Sequence[int], rank: Optional[int] = None,
                  world_size: Optional[int] = None) -> np.ndarray:
    """The molecules of one global batch that this rank processes."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    idx = np.asarray(indices)
    lo, hi = shard_range(len(idx), rank, world_size)
    return idx[lo:hi]


class FlatGradAllReduce:
    """One flat-bucket all-reduce of all existing gradients per step."""

    def __init__(self, world_size: Optional[int] = None, group=None):
        self.group = group
        self.world_size = world_size if world_size is not None else dist.get_world_size(group)
        self._bucket: Optional[torch.Tensor] = None

    def reduce_flat(self, bucket: torch.Tensor) -> None:
        """Gradients that already live in one flat arena: all-reduce the trained range in place."""
        if self.world_size == 1 or bucket.numel() == 0:
            return
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
        bucket.mul_(1.0 / self.world_size)

    def __call__(self, module: torch.nn.Module) -> None:
        grads: List[torch.Tensor] = [p.grad for p in module.parameters() if p.grad is not None]
        if not grads or self.world_size == 1:
            return
        n = sum(g.numel() for g in grads)
        if self._bucket is None or self._bucket.numel() != n or self._bucket.device != grads[0].device:
            self._bucket = torch.empty(n, dtype=torch.float32, device=grads[0].device)
        off = 0
        views = []
        for g in grads:
            v = self._bucket[off:off + g.numel()].view_as(g)
            v.copy_(g)
            views.append(v)
            off += g.numel()
        dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM, group=self.group)
        self._bucket.mul_(1.0 / self.world_size)
        for g, v in zip(grads, views):
            g.copy_(v)


class FlatGradArena:
    """One flat fp32 buffer behind ``p.grad`` of every trainable parameter of a module whose backward is driven by
    autograd (``MPNNModel``, ``WeaveModel``): autograd accumulates into an existing ``.grad`` in place, so the
    gradients of a step land side by side and the data-parallel exchange is ``FlatGradAllReduce.reduce_flat`` on the
    arena -- one collective, no copy in, no copy out.  (``GraphConvModel`` has its own arena in ``native.NativeNet``.)"""

    def __init__(self, module: torch.nn.Module, home_params: bool = False):
        ps = [p for p in module.parameters() if p.requires_grad]
        if not ps:
            raise ValueError("the module has no trainable parameters")
        dev, dt = ps[0].device, ps[0].dtype
        if any(p.device != dev or p.dtype != dt for p in ps):
            raise ValueError("parameters on several devices / of several dtypes")
        sizes = [((p.numel() + 3) // 4) * 4 for p in ps]  # every block 16-byte aligned
        self.flat = torch.zeros(sum(sizes), dtype=dt, device=dev)
        self.views = []
        self.slices = []
        off = 0
        for p, n in zip(ps, sizes):
            self.views.append((p, self.flat[off:off + p.numel()].view_as(p)))
            self.slices.append((off, p.numel()))
            off += n
        # home_params: the parameters themselves move into one flat buffer of the same layout, so that the optimizer
        # step is ONE launch over [params | grads | moments] (GcmiAdam.attach_flat / step_flat) instead of one per tensor
        self.pflat = None
        if home_params:
            self.pflat = torch.zeros_like(self.flat)
            with torch.no_grad():
                for p, (o, n) in zip(ps, self.slices):
                    v = self.pflat[o:o + n].view_as(p)
                    v.copy_(p.data)
                    p.data = v

    def params_homed(self) -> bool:
        """Are the parameters still the views of ``pflat`` they were made (``load_state_dict`` copies in place and keeps
        them; ``module.to(...)`` or an assignment to ``p.data`` does not)?"""
        return self.pflat is not None and all(
            p.data_ptr() == self.pflat.data_ptr() + 4 * o for (p, _), (o, _n) in zip(self.views, self.slices))

    def attach(self) -> None:
        """Instead of ``optimizer.zero_grad()``: every gradient a zeroed view of the arena."""
        self.flat.zero_()
        for p, v in self.views:
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def covers(self, module: torch.nn.Module) -> bool:
        """Are the module's trainable parameters still exactly the ones this arena was built for?"""
        ps = [p for p in module.parameters() if p.requires_grad]
        return len(ps) == len(self.views) and all(a is b for a, (b, _) in zip(ps, self.views))

    def intact(self) -> bool:
        """Did the backward leave every gradient in the arena (it replaces ``.grad`` only in exotic cases:
        sparse gradients, ``create_graph``)?"""
        return all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in self.views)


def shard_model(model, group=None) -> None:
    """Make ``model.fit*`` data-parallel across the initialised process group:
    broadcast rank 0's parameters and buffers, then all-reduce gradients each step."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    with torch.no_grad():
        for t in list(model.model.parameters()) + list(model.model.buffers()):
            dist.broadcast(t, src=0, group=group)
    model._grad_sync = FlatGradAllReduce(group=group)
    # models whose backward runs through autograd get one flat gradient arena, so that their exchange is the same single
    # zero-copy all-reduce (GraphConvModel's native step brings its own)
    if getattr(model, "_grad_arena", None) is not None and model._grad_arena.covers(model.model):
        return  # (MPNNModel builds its own, with the parameters homed as well)
    model._grad_arena = None
    if not hasattr(model.model, "_native_net"):
        try:
            model._grad_arena = FlatGradArena(model.model)
        except ValueError:
            model._grad_arena = None

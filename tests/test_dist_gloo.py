"""Data-parallel path on CPU: world_size 2, gloo.  The sharding rule and the flat-bucket
gradient all-reduce are device-independent host logic; the GPU run uses the same code
with backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepchem_amd.dist import FlatGradAllReduce, shard_indices, shard_model, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Tiny(torch.nn.Module):

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 5)
        self.frozen = torch.nn.Linear(5, 5)  # never used in the loss: grad stays None (reference mode)
        self.b = torch.nn.Linear(5, 3)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


class _Holder:
    """The two attributes shard_model touches on a TorchModel."""

    def __init__(self, module):
        self.model = module
        self._grad_sync = None


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # different initial weights per rank: broadcast must fix that
        holder = _Holder(_Tiny())
        shard_model(holder)
        gen = torch.Generator().manual_seed(0)
        X = torch.randn((8, 6), generator=gen)
        Y = torch.randn((8, 3), generator=gen)
        idx = shard_indices(np.arange(8))
        assert len(idx) == 4 and idx[0] == 4 * rank
        out = holder.model(X[idx])
        loss = ((out - Y[idx])**2).mean()
        loss.backward()
        holder._grad_sync(holder.model)
        assert holder.model.frozen.weight.grad is None
        grads = {k: p.grad.clone() for k, p in holder.model.named_parameters() if p.grad is not None}
        state = {k: v.clone() for k, v in holder.model.state_dict().items()}
        torch.save({"grads": grads, "state": state}, os.path.join(tmp, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 100, 65536):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_flat_allreduce_world1_is_identity():
    m = _Tiny()
    m(torch.ones(2, 6)).sum().backward()
    before = [p.grad.clone() for p in m.parameters() if p.grad is not None]
    FlatGradAllReduce(world_size=1)(m)
    after = [p.grad for p in m.parameters() if p.grad is not None]
    assert all(torch.equal(a, b) for a, b in zip(before, after))


@pytest.mark.timeout(120)
def test_two_ranks_reproduce_the_global_batch_gradient(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(str(tmp_path / "rank0.pt"))
    r1 = torch.load(str(tmp_path / "rank1.pt"))
    # same parameters on both ranks (rank 0's, broadcast) and same averaged gradients
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k
    for k in r0["grads"]:
        assert torch.allclose(r0["grads"][k], r1["grads"][k], atol=1e-7), k
    # ... equal to the single-process gradient of the mean loss over the global batch
    torch.manual_seed(100)
    ref = _Tiny()
    ref.load_state_dict(r0["state"])
    gen = torch.Generator().manual_seed(0)
    X = torch.randn((8, 6), generator=gen)
    Y = torch.randn((8, 3), generator=gen)
    ((ref(X) - Y)**2).mean().backward()
    for k, p in ref.named_parameters():
        if p.grad is None:
            assert k not in r0["grads"]
            continue
        assert torch.allclose(p.grad, r0["grads"][k], atol=1e-6), k


def _arena_worker(rank, world, port, tmp):
    """The generic (autograd-driven) data-parallel step of TorchModel._train_step with a flat gradient arena:
    attach -> backward -> ONE reduce_flat on the arena."""
    from deepchem_amd.dist import FlatGradArena
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(200 + rank)
        holder = _Holder(_Tiny())
        shard_model(holder)
        arena = holder._grad_arena
        assert isinstance(arena, FlatGradArena) and arena.covers(holder.model)
        gen = torch.Generator().manual_seed(1)
        X = torch.randn((8, 6), generator=gen)
        Y = torch.randn((8, 3), generator=gen)
        idx = shard_indices(np.arange(8))
        for _ in range(2):  # the second round starts from re-zeroed views of the same buffer
            arena.attach()
            ((holder.model(X[idx]) - Y[idx])**2).mean().backward()
            assert arena.intact()
            holder._grad_sync.reduce_flat(arena.flat)
        # every trainable parameter's .grad is a view of the one buffer; a parameter the loss never reaches keeps zeros
        base = arena.flat.data_ptr()
        for p in holder.model.parameters():
            assert base <= p.grad.data_ptr() < base + 4 * arena.flat.numel()
        assert float(holder.model.frozen.weight.grad.abs().max()) == 0.0
        grads = {k: p.grad.clone() for k, p in holder.model.named_parameters()}
        torch.save({"grads": grads, "state": {k: v.clone() for k, v in holder.model.state_dict().items()}},
                   os.path.join(tmp, "arena%d.pt" % rank))
        # a replaced parameter is noticed (TorchModel._train_step then takes the per-tensor exchange)
        holder.model.b.weight = torch.nn.Parameter(holder.model.b.weight.detach().clone())
        assert not arena.covers(holder.model)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_gradient_arena_on_two_ranks(tmp_path):
    world = 2
    mp.spawn(_arena_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(str(tmp_path / "arena0.pt"))
    r1 = torch.load(str(tmp_path / "arena1.pt"))
    for k in r0["grads"]:
        assert torch.allclose(r0["grads"][k], r1["grads"][k], atol=1e-7), k
    ref = _Tiny()
    ref.load_state_dict(r0["state"])
    gen = torch.Generator().manual_seed(1)
    X = torch.randn((8, 6), generator=gen)
    Y = torch.randn((8, 3), generator=gen)
    ((ref(X) - Y)**2).mean().backward()
    for k, p in ref.named_parameters():
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(want, r0["grads"][k], atol=1e-6), k

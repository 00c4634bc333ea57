"""Mirror of the two segment reductions of deepchem/utils/pytorch_utils.py that
GraphGather is made of, plus ``get_activation``.

On the hot path GraphGather calls the fused readout kernel directly; these
free functions exist for callers of the reference's utility API and run the
same kernel (rows are first brought into segment order if they are not).
"""
from typing import Callable, Union

import torch

from deepchem_amd import ops
from deepchem_amd._lib import GcmiError
from deepchem_amd.graph import BatchGraph


def get_activation(fn: Union[Callable, str]):
    """deepchem/utils/pytorch_utils.py:8-17."""
    if isinstance(fn, str):
        return getattr(torch.nn.functional, fn)
    return fn


def _segment_reduce(data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int):
    if len(segment_ids.shape) != 1:
        raise AssertionError("segment_ids have be a 1-D tensor")
    if data.shape[0] != segment_ids.shape[0]:
        raise AssertionError("segment_ids should be the same size as dimension 0 of input.")
    if not data.is_cuda:
        raise GcmiError("unsorted_segment_*: CUDA tensors only (no CPU path)")
    shape = data.shape
    x = data.reshape(shape[0], -1).to(torch.float32)
    ids = segment_ids.to(device=data.device, dtype=torch.int32)
    n = x.shape[0]
    if n > 1 and bool((ids[1:] < ids[:-1]).any()):
        order = torch.argsort(ids, stable=True)  # keeps row order inside a segment: first-max rule
        x = x.index_select(0, order)
        ids = ids.index_select(0, order)
    g = BatchGraph([n] + [0] * 10, torch.empty(0, dtype=torch.int32, device=data.device),
                   ids.contiguous())
    out, _ = ops.readout(g, ops.rowmajor(x), num_segments)
    f = x.shape[1]
    return out[:, :f].reshape((num_segments,) + tuple(shape[1:])), \
        out[:, f:].reshape((num_segments,) + tuple(shape[1:]))


def unsorted_segment_sum(data: torch.Tensor, segment_ids: torch.Tensor,
                         num_segments: int) -> torch.Tensor:
    """deepchem/utils/pytorch_utils.py:20-74."""
    return _segment_reduce(data, segment_ids, num_segments)[0].type(data.dtype)


def unsorted_segment_max(data: torch.Tensor, segment_ids: torch.Tensor,
                         num_segments: int) -> torch.Tensor:
    """deepchem/utils/pytorch_utils.py:473-528 (empty segments: -inf)."""
    return _segment_reduce(data, segment_ids, num_segments)[1].type(data.dtype)

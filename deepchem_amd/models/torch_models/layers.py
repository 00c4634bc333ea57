"""GraphConv / GraphPool / GraphGather with the reference's layer contract
(deepchem/models/torch_models/layers.py:6061-6479), computed by libgcmi.so.

Same constructors, same ``forward(inputs: List[Tensor])`` with
``inputs = [atom_features, deg_slice, membership, deg_adj_1 .. deg_adj_10]``,
same parameter containers (``W_list`` / ``b_list`` are plain ``nn.ParameterList``
attributes that callers may replace wholesale, test_layers.py:1480-1485), same
errors.  Differences, all opt-in:

* ``grad_mode`` -- ``"reference"`` (default) reproduces the reference's autograd
  cut at GraphConv (layers.py:6204/:6216/:6226/:6244: the output carries no
  gradient); ``"full"`` back-propagates through the gather and the per-degree
  affine maps with hand-written backward kernels.
* inputs must live on the GPU (the reference's layers only run on the CPU
  because of their NumPy hops); CPU tensors raise -- there is no CPU path.
"""
from typing import Callable, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init as initializers

from deepchem_amd import ops
from deepchem_amd._lib import GcmiError
from deepchem_amd.graph import BatchGraph, graph_for_layer_inputs

_RELU_FNS = (F.relu, torch.relu)
_TANH_FNS = (torch.tanh, F.tanh)


def _is_relu(fn) -> bool:
    return fn in _RELU_FNS or isinstance(fn, nn.ReLU)


def _is_tanh(fn) -> bool:
    return fn in _TANH_FNS or isinstance(fn, nn.Tanh)


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise GcmiError("%s: atom_features must be a CUDA tensor; deepchem_amd has no CPU "
                        "implementation of the graph-convolution path" % what)


def _graph(inputs, graph: Optional[BatchGraph]) -> BatchGraph:
    if graph is not None:
        return graph
    return graph_for_layer_inputs(inputs, inputs[0].device)


class GraphConv(nn.Module):
    """Graph convolution of Duvenaud et al. (reference layers.py:6061-6246):
    per degree d, ``act(sum_neighbours . W_rel[d] + b_rel[d] + self . W_self[d] + b_self[d])``.
    """

    def __init__(self, out_channel: int, number_input_features: int, min_deg: int = 0,
                 max_deg: int = 10, activation_fn: Optional[Callable] = None,
                 grad_mode: str = "reference", **kwargs):
        super(GraphConv, self).__init__(**kwargs)
        if min_deg != 0:
            raise ValueError("only min_deg=0 is supported (the collated layout starts at degree 0)")
        if grad_mode not in ("reference", "full"):
            raise ValueError("grad_mode must be 'reference' or 'full'")
        self.out_channel: int = out_channel
        self.min_degree: int = min_deg
        self.max_degree: int = max_deg
        self.number_input_features: int = number_input_features
        self.activation_fn: Optional[Callable] = activation_fn
        self.grad_mode = grad_mode
        num_deg = 2 * self.max_degree + (1 - self.min_degree)
        # order: rel_1, self_1, ..., rel_max, self_max, self_0  (reference :6189-6224)
        self.W_list = nn.ParameterList([
            nn.Parameter(initializers.xavier_uniform_(torch.empty(number_input_features, out_channel)))
            for _ in range(num_deg)
        ])
        self.b_list = nn.ParameterList(
            [nn.Parameter(initializers.zeros_(torch.empty(out_channel,))) for _ in range(num_deg)])
        self.built = True

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}(out_channel:{self.out_channel},min_deg:{self.min_degree},'
                f'max_deg:{self.max_degree},activation_fn:{self.activation_fn})')

    def _packed(self):
        w = torch.stack(list(self.W_list))  # (2*max+1, K, out)
        b = torch.stack(list(self.b_list))  # (2*max+1, out)
        m = self.max_degree
        bsum = torch.cat([b[2 * m:2 * m + 1], b[0:2 * m:2] + b[1:2 * m:2]], 0)  # (max+1, out): both biases
        return w, bsum

    def forward(self, inputs: List[torch.Tensor], graph: Optional[BatchGraph] = None,
                grad_masked: bool = False) -> torch.Tensor:
        """``graph`` / ``grad_masked`` are used by ``_GraphConvTorchModel`` only: a prebuilt
        BatchGraph, and the promise that the consumer's backward applies the ReLU mask."""
        atom_features = inputs[0]
        _require_cuda(atom_features, "GraphConv")
        g = _graph(inputs, graph)
        fused_relu = _is_relu(self.activation_fn)
        x = atom_features.to(torch.float32)
        if self.grad_mode == "reference":
            with torch.no_grad():
                w, bsum = self._packed()
                out = ops.GraphConvFn.apply(x.detach(), w, bsum, g, fused_relu, False)
        else:
            w, bsum = self._packed()
            out = ops.GraphConvFn.apply(x, w, bsum, g, fused_relu, grad_masked and fused_relu)
        if self.activation_fn is not None and not fused_relu:
            out = self.activation_fn(out)
        return out

    def sum_neigh(self, atoms: torch.Tensor, deg_adj_lists, deg_slice=None) -> List[torch.Tensor]:
        """Neighbour sums per degree 1..max (reference :6236-6246; returned as
        tensors on the GPU instead of NumPy arrays)."""
        counts = [0] + [int(a.shape[0]) for a in deg_adj_lists]
        if deg_slice is not None:
            counts[0] = int(deg_slice[0, 1])
        else:
            counts[0] = int(atoms.shape[0]) - sum(counts[1:])
        ds = torch.tensor([[0, c] for c in counts])
        g = BatchGraph.from_layer_inputs(ds, None, list(deg_adj_lists), atoms.device)
        with torch.no_grad():
            s = ops.gather_sum(g, ops.rowmajor(atoms.to(torch.float32)))
        return [s[g.deg_start[d]:g.deg_start[d + 1]] for d in range(1, self.max_degree + 1)]


class GraphPool(nn.Module):
    """Max over {self} U neighbours per atom (reference layers.py:6249-6367)."""

    def __init__(self, min_degree: int = 0, max_degree: int = 10, **kwargs):
        super(GraphPool, self).__init__(**kwargs)
        if min_degree != 0:
            raise ValueError("only min_degree=0 is supported")
        self.min_degree: int = min_degree
        self.max_degree: int = max_degree

    def get_config(self) -> str:
        return f'{self.__class__.__name__}(min_degree:{self.min_degree},max_degree:{self.max_degree})'

    def forward(self, inputs: List[torch.Tensor], graph: Optional[BatchGraph] = None) -> torch.Tensor:
        atom_features = inputs[0]
        _require_cuda(atom_features, "GraphPool")
        g = _graph(inputs, graph)
        return ops.PoolFn.apply(atom_features.to(torch.float32), None, None, None, None, g, False,
                                False, 0.0, 0.0, False)


class GraphGather(nn.Module):
    """Per-molecule ``[sum | max]`` of the atom rows (reference layers.py:6370-6479).
    Always ``batch_size`` rows; empty molecules give (0, -inf)."""

    def __init__(self, batch_size: int, activation_fn: Optional[Callable] = None, **kwargs):
        super(GraphGather, self).__init__(**kwargs)
        self.batch_size: int = batch_size
        self.activation_fn: Optional[Callable] = activation_fn

    def get_config(self) -> str:
        return f'{self.__class__.__name__}(batch_size:{self.batch_size},activation_fn:{self.activation_fn})'

    def forward(self, inputs: List[torch.Tensor], graph: Optional[BatchGraph] = None):
        atom_features = inputs[0]
        _require_cuda(atom_features, "GraphGather")
        assert self.batch_size > 1, "graph_gather requires batches larger than 1"
        g = _graph(inputs, graph)
        fused_tanh = _is_tanh(self.activation_fn)
        out = ops.ReadoutFn.apply(atom_features.to(torch.float32), None, None, None, None, g,
                                  self.batch_size, False, False, 0.0, 0.0, fused_tanh, False)
        if self.activation_fn is not None and not fused_tanh:
            out = self.activation_fn(out)
        return out


from deepchem_amd.models.torch_models.weave_layers import WeaveGather, WeaveLayer  # noqa: E402,F401
from deepchem_amd.models.torch_models.mpnn_layers import EdgeNetwork, GatedRecurrentUnit, SetGather  # noqa: E402,F401

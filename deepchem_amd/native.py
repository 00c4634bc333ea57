"""NativeNet: the whole GraphConv network behind three C calls per training step.

``_GraphConvTorchModel`` keeps the reference's modules and parameter names; this
helper re-points every ``nn.Parameter`` at a slice of ONE flat fp32 arena (in
``parameters()`` order), keeps a same-shaped gradient arena, and drives
``gcmi_model_forward`` / ``gcmi_model_loss_backward`` (include/gcmi.h), which enqueue
every kernel of the step back to back without returning to Python.  The flat
arenas also make the optimizer one launch (``GcmiAdam``) and the data-parallel
gradient exchange one zero-copy all-reduce (``deepchem_amd.dist``).

Numerically this is the same kernel sequence as the autograd path in
``deepchem_amd/ops.py`` (tests compare the two); it exists to remove ~150 Python-
driven launches per step, which is what bounds throughput at the reference's
default batch size of 100 molecules.
"""
import ctypes
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from deepchem_amd import _lib
from deepchem_amd._lib import GcmiModelDesc, GcmiModelIO, MAX_CONV_LAYERS
from deepchem_amd.graph import BatchGraph, _stream


class NativeUnsupported(Exception):
    pass


class NativeNet:

    def __init__(self, module: nn.Module):
        self.module = module
        if getattr(module, "uncertainty", False):
            raise NativeUnsupported("uncertainty head")
        L = len(module.graph_convs)
        if not (1 <= L <= MAX_CONV_LAYERS):
            raise NativeUnsupported("number of GraphConv layers")
        for gc in module.graph_convs:
            from deepchem_amd.models.torch_models.layers import _is_relu
            if not _is_relu(gc.activation_fn) or gc.min_degree != 0 or gc.max_degree != 10:
                raise NativeUnsupported("non-standard GraphConv")
        self.n_layers = L
        self.flat: Optional[torch.Tensor] = None
        self.grad_flat: Optional[torch.Tensor] = None
        self.offsets: Dict[str, int] = {}
        self._ws: Optional[torch.Tensor] = None
        self._out_cache = {}
        self.desc = GcmiModelDesc()
        self.grad_range: Tuple[int, int] = (0, 0)
        self._flatten()

    # ------------------------------------------------------------------ parameter arena
    def _params(self) -> List[nn.Parameter]:
        return list(self.module.parameters())

    def views_intact(self, quick: bool = False) -> bool:
        """Are all parameters still views of the arena?  ``quick`` looks at the first and last
        parameter object only (enough inside a fit loop, where nobody swaps parameters)."""
        if self.flat is None:
            return False
        if quick:
            ps = self._plist
            return ps[0].data_ptr() == self._param_ptrs[0][0] and ps[-1].data_ptr() == self._param_ptrs[-1][0]
        ps = self._params()
        if len(ps) != len(self._param_ptrs) or any(a is not b for a, b in zip(ps, self._plist)):
            return False
        return all(p.data_ptr() == ptr and p.numel() == n for p, (ptr, n) in zip(ps, self._param_ptrs))

    def _flatten(self):
        """Move all parameters into one flat buffer (in parameters() order) and make each
        nn.Parameter a view of its slice.  Values are preserved."""
        ps = self._params()
        if not ps or not ps[0].is_cuda:
            raise NativeUnsupported("parameters are not on a GPU")
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev:
                raise NativeUnsupported("mixed parameter dtype/device")
        total = sum(((p.numel() + 3) // 4) * 4 for p in ps)  # every block 16-byte aligned
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        names = [n for n, _ in self.module.named_parameters()]
        self.offsets = {}
        self._param_ptrs = []
        self._slices = []
        with torch.no_grad():
            for name, p in zip(names, ps):
                n = p.numel()
                view = flat[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = None
                self.offsets[name] = off
                self._slices.append((off, n))
                self._param_ptrs.append((p.data_ptr(), n))
                off += ((n + 3) // 4) * 4
        self.flat, self.grad_flat = flat, grad
        self._plist = ps
        self._build_desc()

    def _build_desc(self):
        m = self.module
        d = self.desc
        L = self.n_layers
        d.n_layers = L
        d.max_deg = 10
        d.n_feat_in = int(m.graph_convs[0].number_input_features)
        for l, gc in enumerate(m.graph_convs):
            d.conv_width[l] = int(gc.out_channel)
            if l > 0 and gc.number_input_features != m.graph_convs[l - 1].out_channel:
                raise NativeUnsupported("GraphConv input width does not chain")
            k, w = int(gc.number_input_features), int(gc.out_channel)
            base = self.offsets["graph_convs.%d.W_list.0" % l]
            for j in range(21):  # the 21 blocks must be back to back in reference order
                if self.offsets["graph_convs.%d.W_list.%d" % (l, j)] != base + j * k * w:
                    raise NativeUnsupported("GraphConv weights are not contiguous")
            bbase = self.offsets["graph_convs.%d.b_list.0" % l]
            for j in range(21):
                if self.offsets["graph_convs.%d.b_list.%d" % (l, j)] != bbase + j * w:
                    raise NativeUnsupported("GraphConv biases are not contiguous")
            d.off_conv_w[l] = base
            d.off_conv_b[l] = bbase
        d.dense_width = int(m.dense.out_features)
        if m.dense.in_features != d.conv_width[L - 1]:
            raise NativeUnsupported("dense input width")
        d.n_tasks = int(m.n_tasks)
        if m.mode == "classification":
            d.mode, d.n_classes = 0, int(m.n_classes)
            head = "reshape_dense"
        else:
            d.mode, d.n_classes = 1, 1
            head = "regression_dense"
        self.head = getattr(m, head)
        has_bn = isinstance(m.batch_norms[0], nn.BatchNorm1d)
        d.batch_norm = 1 if has_bn else 0
        d.grad_mode = 1 if m.grad_mode == "full" else 0
        if has_bn:
            bn0 = m.batch_norms[0]
            d.bn_eps, d.bn_momentum = float(bn0.eps), float(bn0.momentum)
            for i in range(L + 1):
                bn = m.batch_norms[i]
                width = d.conv_width[i] if i < L else d.dense_width
                if not isinstance(bn, nn.BatchNorm1d) or bn.num_features != width or not bn.affine \
                        or not bn.track_running_stats:
                    raise NativeUnsupported("non-standard BatchNorm")
                d.off_bn_gamma[i] = self.offsets["batch_norms.%d.weight" % i]
                d.off_bn_beta[i] = self.offsets["batch_norms.%d.bias" % i]
        d.off_dense_w = self.offsets["dense.weight"]
        d.off_dense_b = self.offsets["dense.bias"]
        d.off_head_w = self.offsets[head + ".weight"]
        d.off_head_b = self.offsets[head + ".bias"]
        d.n_params = self.flat.numel()
        d.storage = {"fp32": 0, "bf16": 1, "bf16+grads": 2}[getattr(m, "activation_storage", "fp32")]

    def refresh(self, quick: bool = False):
        """Re-flatten if someone replaced parameters (``layer.W_list = ...``, ``p.data = ...``) or
        moved the module."""
        if not self.views_intact(quick):
            self._flatten()
        self.desc.grad_mode = 1 if self.module.grad_mode == "full" else 0

    # ------------------------------------------------------------------ buffers
    def _workspace(self, n_atoms: int, n_mols: int) -> torch.Tensor:
        need = int(_lib.load().gcmi_model_workspace_floats(ctypes.byref(self.desc), n_atoms, n_mols))
        if need < 0:
            why = _lib.load().gcmi_last_error()
            raise _lib.GcmiError("gcmi_model_workspace_floats rejected the model description: %s" %
                                 (why.decode() if isinstance(why, bytes) else why))
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.flat.device:
            self._ws = torch.empty(int(need * 1.1) + 1024, dtype=torch.float32, device=self.flat.device)
        return self._ws

    def _io(self, x: torch.Tensor, graph: BatchGraph, want_probs: bool):
        d = self.desc
        dev = self.flat.device
        B = graph.n_mols
        tc = d.n_tasks * d.n_classes
        logits = torch.empty((B, tc), dtype=torch.float32, device=dev)
        probs = torch.empty((B, tc), dtype=torch.float32, device=dev) if (want_probs and d.mode == 0) else None
        fp = torch.empty((B, 2 * d.dense_width), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        io = GcmiModelIO()
        io.d_atom_features = x.data_ptr()
        io.ld_features = int(x.stride(0)) if x.shape[0] > 1 else int(x.shape[1])
        io.d_workspace = self._workspace(graph.n_atoms, B).data_ptr()
        if d.batch_norm:
            for i in range(self.n_layers + 1):
                bn = self.module.batch_norms[i]
                io.d_bn_running_mean[i] = bn.running_mean.data_ptr()
                io.d_bn_running_var[i] = bn.running_var.data_ptr()
                io.d_bn_batches_tracked[i] = bn.num_batches_tracked.data_ptr()
        io.d_logits = logits.data_ptr()
        io.d_probs = probs.data_ptr() if probs is not None else None
        io.d_fingerprint = fp.data_ptr()
        io.d_loss = loss.data_ptr()
        return io, logits, probs, fp, loss

    # ------------------------------------------------------------------ calls
    def forward(self, x: torch.Tensor, graph: BatchGraph, training: bool, want_probs: bool = True):
        """Returns (logits (B, T*C), probs|None, fingerprint (B, 2*dense)), all untrimmed."""
        d = self.desc
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.shape[0] != graph.n_atoms \
                or (x.shape[1] > 1 and x.stride(1) != 1):
            raise ValueError("atom_features must be a float32 CUDA matrix with one row per atom")
        if not (d.n_feat_in <= x.shape[1] < d.n_feat_in + 4):
            raise ValueError("atom_features has %d columns, the model expects %d" % (x.shape[1], d.n_feat_in))
        if graph.mol_runs is None:
            raise ValueError("the graph has no readout plan (set_mols)")
        io, logits, probs, fp, loss = self._io(x, graph, want_probs)
        self._last = (io, logits, probs, fp, loss, x, graph)
        _lib.call("gcmi_model_forward", ctypes.byref(d), graph.ref, ctypes.c_void_p(self.flat.data_ptr()),
                  ctypes.byref(io), 1 if training else 0, _stream())
        return logits, probs, fp

    def loss_backward(self, labels: torch.Tensor, weights: Optional[torch.Tensor], n_rows: int):
        """After ``forward(training=True)``: loss over the first n_rows molecules and the backward
        pass into the gradient arena.  Returns the loss (0-dim device tensor)."""
        io, logits, probs, fp, loss, x, graph = self._last
        d = self.desc
        graph.ensure_rev_pos()
        labels = labels.contiguous().to(torch.float32)
        per_row = d.n_tasks * d.n_classes
        if labels.numel() != n_rows * per_row:
            raise ValueError("labels %s do not match (%d, %d, %d)" % (tuple(labels.shape), n_rows, d.n_tasks,
                                                                      d.n_classes))
        if weights is not None:
            weights = weights.to(torch.float32)
            if d.n_tasks > 1 and weights.numel() * d.n_tasks == labels.numel() // d.n_classes:
                # one weight per molecule ((n,) or (n, 1): what NumpyDataset(X, y) creates): the reference's
                # _StandardLoss broadcasts it over the tasks (torch_model.py:1285-1291)
                weights = weights.reshape(-1, 1).expand(-1, d.n_tasks)
            weights = weights.contiguous()
            if weights.numel() != n_rows * d.n_tasks:
                raise ValueError("weights %s do not match (%d, %d)" % (tuple(weights.shape), n_rows, d.n_tasks))
        lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.call("gcmi_model_loss_backward", ctypes.byref(d), graph.ref,
                  ctypes.c_void_p(self.flat.data_ptr()), ctypes.c_void_p(self.grad_flat.data_ptr()),
                  ctypes.byref(io), ctypes.c_void_p(labels.data_ptr()),
                  ctypes.c_void_p(weights.data_ptr()) if weights is not None else None, int(n_rows),
                  ctypes.byref(lo), ctypes.byref(hi), _stream())
        self.grad_range = (int(lo.value), int(hi.value))
        self._expose_grads()
        return loss

    def _expose_grads(self):
        """p.grad = view of the gradient arena for every trained parameter (None for the rest,
        exactly as the reference leaves them)."""
        lo, hi = self.grad_range
        if getattr(self, "_exposed", None) == (lo, hi, self.grad_flat.data_ptr()):
            return
        self._exposed = (lo, hi, self.grad_flat.data_ptr())
        for p, (off, n) in zip(self._plist, self._slices):
            if lo <= off and off + n <= hi:
                g = p.grad
                if g is None or g.data_ptr() != self.grad_flat.data_ptr() + 4 * off:
                    p.grad = self.grad_flat[off:off + n].view(p.shape)
            else:
                p.grad = None

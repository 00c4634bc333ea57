"""Native batch collation straight to the GPU.

``collate_to_device`` runs ``gcmi_collate`` (C++, multi-threaded; the reference's
``ConvMol.agglomerate_mols``, feat/mol_graphs.py:256-349) over a ``PackedMols``
set into ONE pinned staging arena -- atom features (padded to a multiple of four
columns so every kernel moves 16 bytes per lane), membership, the flattened
neighbour tables and the per-molecule row ranges -- and ships that arena with a
single asynchronous H2D copy instead of the 16 small copies of
``TorchModel._prepare_batch`` (torch_model.py:923-952).

The result is a ``DeviceBatch``: what ``_GraphConvTorchModel.forward`` accepts in
place of the reference's 14-tensor input list.
"""
import ctypes
from typing import Optional

import numpy as np
import torch

from deepchem_amd import _lib
from deepchem_amd.graph import BatchGraph
from deepchem_amd.utils.synthetic import PackedMols


class DeviceBatch:
    """A collated batch resident in HBM.

    atom_features : (N, F_pad) float32, columns >= n_feat are zero
    graph         : BatchGraph (col_idx, membership, mol_runs on the device)
    n_samples     : real molecules in the batch (rows kept by TrimGraphOutput)
    """

    def __init__(self, atom_features: torch.Tensor, graph: BatchGraph, n_samples: int, n_feat: int):
        self.atom_features = atom_features
        self.graph = graph
        self.n_samples = int(n_samples)
        self.n_feat = int(n_feat)

    @property
    def n_atoms(self) -> int:
        return self.graph.n_atoms

    @property
    def n_mols(self) -> int:
        return self.graph.n_mols


class PinnedRing:
    """A few reusable pinned staging buffers (grow-only).  A fresh pinned allocation costs
    milliseconds (hipHostMalloc), far more than collating a 100-molecule batch, so the pipeline
    cycles through ``n`` buffers and waits for a buffer's previous H2D copy before reusing it."""

    def __init__(self, n: int = 3):
        self.bufs = [None] * n
        self.events = [None] * n
        self.i = 0

    def get(self, n_floats: int) -> torch.Tensor:
        k = self.i
        self.i = (self.i + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        buf = self.bufs[k]
        if buf is None or buf.numel() < n_floats:
            buf = torch.empty(int(n_floats * 1.25) + 1024, dtype=torch.float32,
                              pin_memory=torch.cuda.is_available())
            self.bufs[k] = buf
        self._last = k
        return buf

    def mark(self, stream=None):
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        self.events[self._last] = ev


def collate_to_device(packed: PackedMols, sel: Optional[np.ndarray], device: torch.device,
                      n_samples: Optional[int] = None, max_deg: int = 10,
                      pad_features_to: int = 4, ring: Optional[PinnedRing] = None) -> DeviceBatch:
    if sel is None:
        sel = np.arange(packed.n_mols, dtype=np.int64)
    sel = np.ascontiguousarray(sel, np.int64)
    n_sel = int(sel.shape[0])
    if n_sel and (sel.min() < 0 or sel.max() >= packed.n_mols):
        raise IndexError("molecule index outside the set")
    na, ne = ctypes.c_int64(), ctypes.c_int64()
    _lib.call("gcmi_collate_sizes", packed.atom_ptr.ctypes.data, packed.adj_ptr.ctypes.data,
              sel.ctypes.data, n_sel, ctypes.byref(na), ctypes.byref(ne))
    n_atoms, n_edges = int(na.value), int(ne.value)
    n_feat = packed.n_feat
    ld = ((n_feat + pad_features_to - 1) // pad_features_to) * pad_features_to
    n_deg = max_deg + 1
    # one arena: [features | membership | col_idx | mol_runs], every part 16-byte aligned
    def up4(n):
        return (n + 3) // 4 * 4
    off_mem = n_atoms * ld
    off_col = off_mem + up4(n_atoms)
    off_runs = off_col + up4(n_edges)
    total = off_runs + up4(n_sel * n_deg * 2)
    pin = torch.cuda.is_available()
    if ring is not None:
        arena = ring.get(max(total, 4))[:max(total, 4)]
    else:
        arena = torch.empty(max(total, 4), dtype=torch.float32, pin_memory=pin)
    base = arena.data_ptr()
    feats = np.ascontiguousarray(packed.atom_features, np.float32)
    atom_ptr = np.ascontiguousarray(packed.atom_ptr, np.int64)
    adj_ptr = np.ascontiguousarray(packed.adj_ptr, np.int64)
    adj_idx = np.ascontiguousarray(packed.adj_idx, np.int32)
    g = _lib.GcmiGraph()
    _lib.call("gcmi_collate", feats.ctypes.data, n_feat, atom_ptr.ctypes.data, adj_ptr.ctypes.data,
              adj_idx.ctypes.data, sel.ctypes.data, n_sel, max_deg, base, ld, n_atoms,
              base + 4 * off_mem, base + 4 * off_col, n_edges, base + 4 * off_runs, ctypes.byref(g))
    dev_arena = arena.to(device, non_blocking=True)
    if ring is not None:
        ring.mark()
    as_i32 = dev_arena.view(torch.int32)
    x = dev_arena[:n_atoms * ld].view(n_atoms, ld)
    membership = as_i32[off_mem:off_mem + n_atoms]
    col_idx = as_i32[off_col:off_col + n_edges]
    mol_runs = as_i32[off_runs:off_runs + n_sel * n_deg * 2]
    counts = [g.deg_start[d + 1] - g.deg_start[d] for d in range(n_deg)]
    graph = BatchGraph(counts, col_idx, membership, n_mols=n_sel, mol_runs=mol_runs, symmetric=None)
    graph._arena = dev_arena  # keep the storage alive with the graph
    graph.symmetric = _is_symmetric(packed)
    return DeviceBatch(x, graph, n_sel if n_samples is None else n_samples, n_feat)


def _is_symmetric(packed: PackedMols) -> bool:
    """True when every bond is listed from both ends with equal multiplicity (what a
    featurizer produces).  Cached on the set: it is a property of the molecules."""
    cached = getattr(packed, "_symmetric", None)
    if cached is not None:
        return cached
    deg = np.diff(packed.adj_ptr)
    src = np.repeat(np.arange(packed.n_atoms, dtype=np.int64), deg)
    mol_of_atom = np.repeat(np.arange(packed.n_mols, dtype=np.int64), np.diff(packed.atom_ptr))
    dst = packed.adj_idx.astype(np.int64) + packed.atom_ptr[mol_of_atom[src]]
    n = np.int64(packed.n_atoms)
    fwd = np.sort(src * n + dst)
    bwd = np.sort(dst * n + src)
    sym = bool(np.array_equal(fwd, bwd))
    try:
        packed._symmetric = sym
    except Exception:
        pass
    return sym

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
import os
from typing import Optional

import numpy as np
import torch

from deepchem_amd import _lib
from deepchem_amd.graph import BatchGraph
from deepchem_amd.utils.synthetic import PackedMols

# atoms per molecule window of the LDS-staged kernels (gcmi_collate_plans; GCMI_WIN_CAP overrides: tools/ sweeps)
DEFAULT_WIN_CAP = int(os.environ.get("GCMI_WIN_CAP", "96"))


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


class HostBatch:
    """What ``gcmi_collate_plans`` wrote into one (pinned) host arena: float32 words
    [features | membership | col_idx | mol_runs | win_meta | win_edges (uint16) | rev_pos (uint8)],
    every part 16-byte aligned.  ``part(name)`` gives typed views of the arena (host) or of its
    device copy."""

    def __init__(self, arena, offsets, n_atoms, n_edges, n_sel, n_feat, ld, n_deg, deg_counts,
                 symmetric, plan):
        self.arena, self.off = arena, offsets
        self.n_atoms, self.n_edges, self.n_sel = n_atoms, n_edges, n_sel
        self.n_feat, self.ld, self.n_deg = n_feat, ld, n_deg
        self.deg_counts, self.symmetric = deg_counts, symmetric
        self.plan = plan                      # GcmiGraph as gcmi_collate_plans filled it
        self.n_win, self.n_win_big = int(plan.n_win), int(plan.n_win_big)
        self.win_alloc, self.win_ecap = int(plan.win_alloc), int(plan.win_ecap)
        self.win_alloc_big, self.win_ecap_big = int(plan.win_alloc_big), int(plan.win_ecap_big)

    @property
    def n_words(self) -> int:
        return self.off["end"]

    def part(self, name: str, arena: Optional[torch.Tensor] = None) -> torch.Tensor:
        a = self.arena if arena is None else arena
        i32 = a.view(torch.int32)
        o = self.off
        if name == "features":
            return a[:self.n_atoms * self.ld].view(self.n_atoms, self.ld)
        if name == "membership":
            return i32[o["mem"]:o["mem"] + self.n_atoms]
        if name == "col_idx":
            return i32[o["col"]:o["col"] + self.n_edges]
        if name == "mol_runs":
            return i32[o["runs"]:o["runs"] + self.n_sel * self.n_deg * 2]
        if name == "win_meta":
            return i32[o["win"]:o["win"] + self.n_win * _lib.GCMI_WIN_META_INTS]
        if name == "win_edges":
            return a[o["loc"]:o["rev"]].view(torch.int16)
        if name == "rev_pos":
            return a[o["rev"]:o["end"]].view(torch.uint8)[:self.n_edges]
        raise KeyError(name)


def collate_host(packed: PackedMols, sel: Optional[np.ndarray], max_deg: int = 10,
                 pad_features_to: int = 4, ring: Optional[PinnedRing] = None,
                 win_cap: int = DEFAULT_WIN_CAP, pin: Optional[bool] = None) -> HostBatch:
    """Run the native collation into one host arena (no GPU involved)."""
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
    # a set kept as 8-byte atom codes is collated as rows of two "floats" (8 bytes per atom in the arena) and expanded
    # to feature rows on the device (collate_to_device)
    coded = getattr(packed, "atom_codes", None) is not None
    n_feat = 2 if coded else packed.n_feat
    ld = 2 if coded else ((n_feat + pad_features_to - 1) // pad_features_to) * pad_features_to
    n_deg = max_deg + 1

    def up4(n):
        return (n + 3) // 4 * 4
    off = {"mem": up4(n_atoms * ld)}
    off["col"] = off["mem"] + up4(n_atoms)
    off["runs"] = off["col"] + up4(n_edges)
    off["win"] = off["runs"] + up4(n_sel * n_deg * 2)
    off["loc"] = off["win"] + n_sel * _lib.GCMI_WIN_META_INTS   # at most one window per molecule
    off["rev"] = off["loc"] + up4((n_edges + 8 * n_sel + 1) // 2)  # every window padded to 8 entries
    off["end"] = max(off["rev"] + up4((n_edges + 3) // 4), 4)
    total = off["end"]
    if pin is None:
        pin = torch.cuda.is_available()
    if ring is not None:
        arena = ring.get(total)[:total]
    else:
        arena = torch.empty(total, dtype=torch.float32, pin_memory=pin)
    base = arena.data_ptr()
    feats = packed.atom_codes.view(np.float32) if coded else np.ascontiguousarray(packed.atom_features, np.float32)
    atom_ptr = np.ascontiguousarray(packed.atom_ptr, np.int64)
    adj_ptr = np.ascontiguousarray(packed.adj_ptr, np.int64)
    adj_idx = np.ascontiguousarray(packed.adj_idx, np.int32)
    g = _lib.GcmiGraph()
    sym = ctypes.c_int32(1)
    _lib.call("gcmi_collate_plans", feats.ctypes.data, n_feat, atom_ptr.ctypes.data, adj_ptr.ctypes.data,
              adj_idx.ctypes.data, sel.ctypes.data, n_sel, max_deg, base, ld, n_atoms,
              base + 4 * off["mem"], base + 4 * off["col"], n_edges, base + 4 * off["runs"],
              base + 4 * off["rev"], ctypes.byref(sym), int(win_cap), base + 4 * off["win"],
              base + 4 * off["loc"], ctypes.byref(g))
    counts = [g.deg_start[d + 1] - g.deg_start[d] for d in range(n_deg)]
    hb = HostBatch(arena, off, n_atoms, n_edges, n_sel, n_feat, ld, n_deg, counts, bool(sym.value), g)
    hb.coded = coded
    return hb


def collate_to_device(packed: PackedMols, sel: Optional[np.ndarray], device: torch.device,
                      n_samples: Optional[int] = None, max_deg: int = 10,
                      pad_features_to: int = 4, ring: Optional[PinnedRing] = None,
                      win_cap: int = DEFAULT_WIN_CAP) -> DeviceBatch:
    hb = collate_host(packed, sel, max_deg, pad_features_to, ring, win_cap)
    dev_arena = hb.arena.to(device, non_blocking=True)
    if ring is not None:
        ring.mark()
    graph = BatchGraph(hb.deg_counts, hb.part("col_idx", dev_arena), hb.part("membership", dev_arena),
                       n_mols=hb.n_sel, mol_runs=hb.part("mol_runs", dev_arena), symmetric=hb.symmetric)
    graph._arena = dev_arena  # keep the storage alive with the graph
    if hb.symmetric and hb.n_edges:
        graph.attach_rev_pos(hb.part("rev_pos", dev_arena))
    if hb.n_win > 0:
        graph.attach_windows(hb.plan, hb.part("win_meta", dev_arena), hb.part("win_edges", dev_arena))
    feats = hb.part("features", dev_arena)
    n_feat = hb.n_feat
    if getattr(hb, "coded", False):
        from deepchem_amd import ops
        ld_out = ((75 + pad_features_to - 1) // pad_features_to) * pad_features_to
        feats = ops.expand_atom_codes(feats.view(torch.uint8), max(76, (ld_out + 3) // 4 * 4))
        n_feat = 75
    return DeviceBatch(feats, graph, hb.n_sel if n_samples is None else n_samples, n_feat)


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

"""Batches collated BY the GPU from a molecule set that lives in HBM.

``collate_to_device`` (collate.py) builds every batch on the host and ships its whole arena over PCIe; an epoch of
shuffled 65 536-molecule batches (``DiskDataset.iterbatches``, data/datasets.py:1518-1623, feeding
``ConvMol.agglomerate_mols``, feat/mol_graphs.py:256-349) is then bound by the host's collation threads, at about a
third of the rate the training step itself runs at.  Here the set is uploaded once (``ResidentMolSet``: atom rows
or 8-byte atom codes, adjacency, and the per-atom / per-edge tables that do not depend on the batch); per batch the
host does the serial pass over per-molecule degree histograms (``gcmi_collate_plan``) and ``gcmi_collate_rows``
writes the arena on the device -- the same bytes ``gcmi_collate_plans`` writes (tests/test_resident.py).
"""
import ctypes
import os
from typing import Optional

import numpy as np
import torch

from deepchem_amd import _lib
from deepchem_amd.data.collate import DeviceBatch, HostBatch, PinnedRing
from deepchem_amd.graph import BatchGraph
from deepchem_amd.utils.synthetic import PackedMols

# atoms per molecule window of the LDS-staged kernels (gcmi_collate_plans; GCMI_WIN_CAP overrides: tools/ sweeps)
DEFAULT_WIN_CAP = int(os.environ.get("GCMI_WIN_CAP", "96"))

ND = _lib.GCMI_MAX_DEG + 1


def molset_tables(packed: PackedMols, max_deg: int = 10, n_threads: int = 0):
    """(mol_hist [M, 11] int32, rank [A] int32, rev [nnz] uint8, symmetric) -- ``gcmi_molset_tables``."""
    atom_ptr = np.ascontiguousarray(packed.atom_ptr, np.int64)
    adj_ptr = np.ascontiguousarray(packed.adj_ptr, np.int64)
    adj_idx = np.ascontiguousarray(packed.adj_idx, np.int32)
    n_mols = int(atom_ptr.shape[0] - 1)
    n_atoms = int(atom_ptr[-1]) if n_mols else 0
    hist = np.zeros((n_mols, ND), np.int32)
    rank = np.zeros(max(1, n_atoms), np.int32)
    rev = np.zeros(max(1, adj_idx.shape[0]), np.uint8)
    sym = ctypes.c_int32(1)
    _lib.call("gcmi_molset_tables", atom_ptr.ctypes.data, adj_ptr.ctypes.data, adj_idx.ctypes.data, n_mols,
              int(max_deg), hist.ctypes.data, rank.ctypes.data, rev.ctypes.data, ctypes.byref(sym), int(n_threads))
    return hist, rank[:n_atoms], rev[:adj_idx.shape[0]], bool(sym.value)


class BatchPlan:
    """What ``gcmi_collate_plan`` decided for one batch: the staging words the device needs, their offsets, the
    graph descriptor and the arena layout (the one ``HostBatch`` documents)."""

    def __init__(self, staging, offsets, g, n_sel, n_feat, ld, n_deg):
        self.staging, self.offsets, self.g = staging, offsets, g
        self.n_sel, self.n_feat, self.ld, self.n_deg = n_sel, n_feat, ld, n_deg
        self.n_atoms, self.n_edges = int(g.n_atoms), int(g.n_edges)
        self.used = int(offsets[6])

        def up4(n):
            return (n + 3) // 4 * 4
        off = {"mem": up4(self.n_atoms * ld)}
        off["col"] = off["mem"] + up4(self.n_atoms)
        off["runs"] = off["col"] + up4(self.n_edges)
        off["win"] = off["runs"] + up4(n_sel * n_deg * 2)
        off["loc"] = off["win"] + n_sel * _lib.GCMI_WIN_META_INTS
        off["rev"] = off["loc"] + up4((self.n_edges + 8 * n_sel + 1) // 2)
        off["end"] = max(off["rev"] + up4((self.n_edges + 3) // 4), 4)
        self.off = off

    def host_batch(self, arena) -> HostBatch:
        counts = [self.g.deg_start[d + 1] - self.g.deg_start[d] for d in range(self.n_deg)]
        return HostBatch(arena, self.off, self.n_atoms, self.n_edges, self.n_sel, self.n_feat, self.ld, self.n_deg,
                         counts, True, self.g)


def plan_batch(mol_hist: np.ndarray, atom_ptr: np.ndarray, sel: np.ndarray, n_feat: int, ld: int, max_deg: int = 10,
               win_cap: int = DEFAULT_WIN_CAP, staging: Optional[torch.Tensor] = None) -> BatchPlan:
    sel = np.ascontiguousarray(sel, np.int64)
    n_sel = int(sel.shape[0])
    if n_sel and (sel.min() < 0 or sel.max() >= mol_hist.shape[0]):
        raise IndexError("molecule index outside the set")
    words = int(_lib.load().gcmi_collate_plan_words(n_sel))
    if staging is None:
        staging = torch.empty(words, dtype=torch.int32)
    if staging.numel() < words or staging.dtype != torch.int32:
        raise ValueError("staging buffer too small")
    offsets = (ctypes.c_int64 * 8)()
    g = _lib.GcmiGraph()
    _lib.call("gcmi_collate_plan", mol_hist.ctypes.data, atom_ptr.ctypes.data, sel.ctypes.data, n_sel, int(max_deg),
              int(win_cap), staging.data_ptr(), staging.numel(), ctypes.cast(offsets, ctypes.c_void_p),
              ctypes.byref(g))
    return BatchPlan(staging, offsets, g, n_sel, n_feat, ld, max_deg + 1)


class ResidentMolSet:
    """A ``PackedMols`` set uploaded once: features (or atom codes), adjacency and the batch-independent tables."""

    def __init__(self, packed: PackedMols, device: torch.device, max_deg: int = 10, pad_features_to: int = 4):
        self.device = torch.device(device)
        self.max_deg = int(max_deg)
        self.n_mols = packed.n_mols
        self.coded = getattr(packed, "atom_codes", None) is not None
        self.pad_features_to = pad_features_to
        self.mol_hist, rank, rev, self.symmetric = molset_tables(packed, max_deg)
        if not self.symmetric:
            raise ValueError("the set lists some bond from one end only: collate it on the host (collate_to_device)")
        self.atom_ptr = np.ascontiguousarray(packed.atom_ptr, np.int64)
        if self.coded:
            feats = np.ascontiguousarray(packed.atom_codes).view(np.float32).reshape(-1, 2)
            self.n_feat, self.ld = 2, 2
        else:
            feats = np.ascontiguousarray(packed.atom_features, np.float32)
            self.n_feat = int(packed.n_feat)
            self.ld = (self.n_feat + pad_features_to - 1) // pad_features_to * pad_features_to
        dev = self.device
        self.d_feats = torch.from_numpy(feats).to(dev)
        self.d_adj_ptr = torch.from_numpy(np.ascontiguousarray(packed.adj_ptr, np.int64)).to(dev)
        self.d_adj_idx = torch.from_numpy(np.ascontiguousarray(packed.adj_idx, np.int32)).to(dev)
        self.d_rank = torch.from_numpy(np.ascontiguousarray(rank)).to(dev)
        self.d_rev = torch.from_numpy(np.ascontiguousarray(rev)).to(dev)
        torch.cuda.current_stream(dev).synchronize()  # the collating streams start reading right away

    @staticmethod
    def bytes_needed(packed: PackedMols) -> int:
        coded = getattr(packed, "atom_codes", None) is not None
        per_atom = 8 if coded else 4 * int(packed.n_feat)
        return int(packed.n_atoms) * (per_atom + 8 + 4) + int(packed.adj_idx.shape[0]) * 5

    def collate(self, sel: np.ndarray, n_samples: Optional[int] = None, ring: Optional[PinnedRing] = None,
                win_cap: int = DEFAULT_WIN_CAP) -> DeviceBatch:
        """The batch of molecules ``sel`` (in order, repeats allowed), built on the current stream."""
        n_sel = int(np.shape(sel)[0])
        words = int(_lib.load().gcmi_collate_plan_words(n_sel))
        if ring is not None:
            staging = ring.get(words).view(torch.int32)[:words]
        else:
            staging = torch.empty(words, dtype=torch.int32, pin_memory=True)
        plan = plan_batch(self.mol_hist, self.atom_ptr, sel, self.n_feat, self.ld, self.max_deg, win_cap, staging)
        dev = self.device
        d_plan = staging[:max(plan.used, 1)].to(dev, non_blocking=True)
        if ring is not None:
            ring.mark()
        arena = torch.empty(plan.off["end"], dtype=torch.float32, device=dev)
        hb = plan.host_batch(None)
        base = arena.data_ptr()
        off = plan.off
        d_off = plan.offsets
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        # wide float rows are copied by their own launch (lanes along the columns); 8-byte codes by the atom's thread
        src_atom = None if self.n_feat <= 4 else torch.empty(max(plan.n_atoms, 1), dtype=torch.int64, device=dev)
        _lib.call("gcmi_collate_rows", self.d_feats.data_ptr(), self.n_feat, self.d_adj_ptr.data_ptr(),
                  self.d_adj_idx.data_ptr(), self.d_rank.data_ptr(), self.d_rev.data_ptr(), d_plan.data_ptr(),
                  ctypes.cast(d_off, ctypes.c_void_p), ctypes.byref(plan.g), base, self.ld,
                  base + 4 * off["mem"], base + 4 * off["col"], base + 4 * off["runs"], base + 4 * off["rev"],
                  base + 4 * off["loc"], None if src_atom is None else src_atom.data_ptr(), stream)
        graph = BatchGraph(hb.deg_counts, hb.part("col_idx", arena), hb.part("membership", arena), n_mols=n_sel,
                           mol_runs=hb.part("mol_runs", arena), symmetric=True)
        graph._arena = arena
        graph._plan = d_plan  # the window descriptors the kernels read live in the plan's device copy
        if plan.n_edges:
            graph.attach_rev_pos(hb.part("rev_pos", arena))
        if hb.n_win > 0:
            meta = d_plan[int(d_off[4]):int(d_off[4]) + hb.n_win * _lib.GCMI_WIN_META_INTS]
            graph.attach_windows(plan.g, meta, hb.part("win_edges", arena))
        feats = hb.part("features", arena)
        n_feat = self.n_feat
        if self.coded:
            from deepchem_amd import ops
            ld_out = ((75 + self.pad_features_to - 1) // self.pad_features_to) * self.pad_features_to
            feats = ops.expand_atom_codes(feats.view(torch.uint8), max(76, (ld_out + 3) // 4 * 4))
            n_feat = 75
        return DeviceBatch(feats, graph, n_sel if n_samples is None else n_samples, n_feat)

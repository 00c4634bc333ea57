"""Collation on the device over a resident molecule set (deepchem_amd/data/resident.py) against the host collation
(gcmi_collate_plans = ConvMol.agglomerate_mols, feat/mol_graphs.py:256-349, + the kernel plans).  Integer / byte
work: every array must be identical.  The CPU half runs the SAME per-atom routine through gcmi_collate_rows_host."""
import ctypes

import numpy as np
import pytest
import torch

from deepchem_amd import _lib
from deepchem_amd.data.collate import collate_host
from deepchem_amd.data.resident import ND, molset_tables, plan_batch
from deepchem_amd.utils.synthetic import (PackedMols, concat_packed, single_atom_and_edge_cases,
                                          synthetic_molecules)

PARTS = ["features", "membership", "col_idx", "mol_runs", "win_meta", "rev_pos"]


def _pack(seed=0, n=300, n_feat=8, big=False):
    sets = [synthetic_molecules(n, seed=seed, n_feat=n_feat),
            single_atom_and_edge_cases(n_feat, seed),
            synthetic_molecules(6, seed=seed + 1, n_feat=n_feat, mean_atoms=14, max_atoms=40,
                                parent_weights=(1,) * 10, ring_deg=10, ring_p_deg3=1.0, rings_per_atom=0.8)]
    if big:  # molecules above the window capacity get windows of their own
        sets.append(synthetic_molecules(3, seed=seed + 2, n_feat=n_feat, mean_atoms=150, min_atoms=120, max_atoms=200))
    return concat_packed(sets)


def _with_empty_molecules(packed: PackedMols) -> PackedMols:
    """Two molecules without atoms spliced in (a featurizer's empty ConvMol)."""
    ptr = packed.atom_ptr
    k = len(ptr) // 2
    atom_ptr = np.concatenate([ptr[:1], ptr[:k], ptr[k - 1:k], ptr[k:], ptr[-1:]])
    return PackedMols(packed.atom_features, atom_ptr, packed.adj_ptr, packed.adj_idx)


def _win_entries(hb):
    """The window edge entries that belong to windows (the arena part is sized for the worst case)."""
    meta = hb.part("win_meta").numpy().reshape(hb.n_win, _lib.GCMI_WIN_META_INTS)
    n = 0
    for m in meta:
        n = max(n, int(m[2 * ND]) + (int(m[2 * ND + 1]) + 7) // 8 * 8)
    return hb.part("win_edges").numpy()[:n]


def _rows_on_host(packed, sel, win_cap=96):
    hist, rank, rev, sym = molset_tables(packed)
    assert sym
    coded = packed.atom_codes is not None
    n_feat = 2 if coded else packed.n_feat
    ld = 2 if coded else (n_feat + 3) // 4 * 4
    atom_ptr = np.ascontiguousarray(packed.atom_ptr, np.int64)
    plan = plan_batch(hist, atom_ptr, sel, n_feat, ld, win_cap=win_cap)
    arena = torch.zeros(plan.off["end"], dtype=torch.float32)
    base, off = arena.data_ptr(), plan.off
    feats = (np.ascontiguousarray(packed.atom_codes).view(np.float32) if coded
             else np.ascontiguousarray(packed.atom_features, np.float32))
    adj_ptr = np.ascontiguousarray(packed.adj_ptr, np.int64)
    adj_idx = np.ascontiguousarray(packed.adj_idx, np.int32)
    _lib.call("gcmi_collate_rows_host", feats.ctypes.data, n_feat, adj_ptr.ctypes.data, adj_idx.ctypes.data,
              rank.ctypes.data, rev.ctypes.data, plan.staging.data_ptr(), ctypes.cast(plan.offsets, ctypes.c_void_p),
              ctypes.byref(plan.g), base, ld, base + 4 * off["mem"], base + 4 * off["col"], base + 4 * off["runs"],
              base + 4 * off["rev"], base + 4 * off["loc"])
    hb = plan.host_batch(arena)
    o4 = int(plan.offsets[4])
    meta = plan.staging[o4:o4 + hb.n_win * _lib.GCMI_WIN_META_INTS]
    arena.view(torch.int32)[off["win"]:off["win"] + meta.numel()] = meta
    return hb


def _same_batch(a, b):
    for f in ("n_atoms", "n_edges", "n_sel", "n_win", "n_win_big", "win_alloc", "win_ecap", "win_alloc_big",
              "win_ecap_big"):
        assert getattr(a, f) == getattr(b, f), f
    assert list(a.deg_counts) == list(b.deg_counts)
    for name in PARTS:
        if name == "win_meta" and a.n_win == 0:
            continue
        x, y = a.part(name).numpy(), b.part(name).numpy()
        assert x.shape == y.shape, name
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), name
    if a.n_win:
        assert np.array_equal(_win_entries(a), _win_entries(b))


def test_tables_are_the_per_molecule_facts():
    packed = _pack(1, n=120)
    hist, rank, rev, sym = molset_tables(packed)
    assert sym
    deg = np.diff(packed.adj_ptr)
    for m in range(packed.n_mols):
        a0, a1 = int(packed.atom_ptr[m]), int(packed.atom_ptr[m + 1])
        d = deg[a0:a1]
        assert np.array_equal(hist[m], np.bincount(d, minlength=ND))
        seen = {}
        for i, di in enumerate(d):
            assert rank[a0 + i] == seen.get(int(di), 0)
            seen[int(di)] = seen.get(int(di), 0) + 1
        for i in range(a1 - a0):
            e0, e1 = int(packed.adj_ptr[a0 + i]), int(packed.adj_ptr[a0 + i + 1])
            for e in range(e0, e1):
                nb = int(packed.adj_idx[e])
                f0 = int(packed.adj_ptr[a0 + nb])
                assert packed.adj_idx[f0 + int(rev[e])] == i   # the reverse slot points back at the owner
    # thread count does not matter
    for t in (1, 3):
        h2, r2, v2, _ = molset_tables(packed, n_threads=t)
        assert np.array_equal(h2, hist) and np.array_equal(r2, rank) and np.array_equal(v2, rev)


@pytest.mark.parametrize("win_cap", [32, 96, 1000])
def test_rows_routine_writes_the_host_collation(win_cap):
    packed = _pack(3, big=True)
    rng = np.random.RandomState(5)
    sel = rng.permutation(packed.n_mols)
    sel = np.concatenate([sel, sel[:17]])              # repeats = pad_batch tiling (data/datasets.py:204-216)
    _same_batch(_rows_on_host(packed, sel, win_cap), collate_host(packed, sel, win_cap=win_cap, pin=False))


def test_rows_routine_on_ragged_batches():
    packed = _with_empty_molecules(_pack(7, n=60))
    n = packed.n_mols
    for sel in (np.arange(n), np.arange(n)[::-1].copy(), np.array([n // 2]), np.array([n // 2 - 1, n // 2 - 1]),
                np.array([0, n - 1, n // 2, 3, 3, 3])):
        _same_batch(_rows_on_host(packed, sel), collate_host(packed, sel, pin=False))


def _coded_set():
    import os
    import deepchem_amd as dc
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "smiles_sample.txt")) as f:
        smiles = [l.strip() for l in f if l.strip() and not l.startswith("#")]
    packed, _ = dc.feat.ConvMolFeaturizer().featurize_packed(smiles)
    assert packed.atom_codes is not None
    return packed


def test_rows_routine_on_atom_codes_from_real_smiles():
    packed = _coded_set()
    sel = np.random.RandomState(3).permutation(packed.n_mols)
    _same_batch(_rows_on_host(packed, sel), collate_host(packed, sel, pin=False))


def test_one_sided_bonds_and_bad_neighbours_are_reported():
    atom_ptr = np.array([0, 2], np.int64)
    adj_ptr = np.array([0, 1, 1], np.int64)
    packed = PackedMols(np.zeros((2, 4), np.float32), atom_ptr, adj_ptr, np.array([1], np.int32))
    assert molset_tables(packed)[3] is False
    bad = PackedMols(np.zeros((2, 4), np.float32), atom_ptr, adj_ptr, np.array([2], np.int32))
    with pytest.raises(_lib.GcmiError, match="outside its molecule"):
        molset_tables(bad)
    with pytest.raises(_lib.GcmiError, match="max_deg"):
        molset_tables(_pack(0, n=20), max_deg=2)
    small = _pack(0, n=20)
    hist, _, _, _ = molset_tables(small)
    with pytest.raises(_lib.GcmiError, match="max_deg"):   # tables of a wider set planned with a narrower degree range
        plan_batch(hist, np.ascontiguousarray(small.atom_ptr, np.int64), np.arange(small.n_mols), 8, 8, max_deg=2)
    empty = plan_batch(hist, np.ascontiguousarray(small.atom_ptr, np.int64), np.zeros(0, np.int64), 8, 8)
    assert empty.n_atoms == 0 and empty.n_edges == 0 and int(empty.g.n_win) == 0
    with pytest.raises(IndexError):
        hist, _, _, _ = molset_tables(_pack(0, n=20))
        plan_batch(hist, np.ascontiguousarray(_pack(0, n=20).atom_ptr, np.int64), np.array([10 ** 6]), 8, 8)


# ------------------------------------------------------------------------------------------------ GPU


@pytest.mark.gpu
@pytest.mark.parametrize("coded", [False, True])
def test_device_collation_equals_host_collation(coded):
    from deepchem_amd.data.collate import collate_to_device
    from deepchem_amd.data.resident import ResidentMolSet
    dev = torch.device("cuda:0")
    if coded:
        packed = _coded_set()
    else:
        packed = _pack(9, n=2000, n_feat=75, big=True)
    rset = ResidentMolSet(packed, dev)
    rng = np.random.RandomState(2)
    for trial in range(3):
        sel = rng.permutation(packed.n_mols)[:packed.n_mols - 7 * trial]
        sel = np.concatenate([sel, sel[:5]])
        a = rset.collate(sel)
        b = collate_to_device(packed, sel, dev)
        torch.cuda.synchronize()
        assert a.n_feat == b.n_feat and a.n_samples == b.n_samples
        assert torch.equal(a.atom_features, b.atom_features)
        ga, gb = a.graph, b.graph
        assert ga.deg_counts == gb.deg_counts and ga.n_mols == gb.n_mols
        for name in ("col_idx", "membership", "mol_runs", "rev_pos", "win_meta"):
            assert torch.equal(getattr(ga, name), getattr(gb, name)), name
        for f in ("n_win", "n_win_big", "win_alloc", "win_ecap", "win_alloc_big", "win_ecap_big"):
            assert getattr(ga.c, f) == getattr(gb.c, f), f
        meta = ga.win_meta.cpu().numpy().reshape(-1, _lib.GCMI_WIN_META_INTS)
        n = max(int(m[2 * ND]) + (int(m[2 * ND + 1]) + 7) // 8 * 8 for m in meta)
        assert torch.equal(ga.win_edges[:n], gb.win_edges[:n])


@pytest.mark.gpu
def test_pipeline_over_a_resident_set_hands_out_the_same_batches():
    from deepchem_amd.data.packed_dataset import DeviceBatchPipeline
    dev = torch.device("cuda:0")
    packed = _pack(4, n=1500, n_feat=75)
    n = packed.n_mols
    y = np.random.RandomState(0).standard_normal((n, 3))
    w = np.ones((n, 3))
    rng = np.random.RandomState(1)
    batches = [(rng.permutation(n)[:256].astype(np.int64), 256) for _ in range(6)]
    out = {}
    for resident in (False, True):
        pipe = DeviceBatchPipeline(packed, y, w, batches, dev, resident=resident)
        assert (pipe.resident is not None) == resident
        rows = []
        for batch, y_t, w_t in pipe:
            rows.append((batch.atom_features.clone(), batch.graph.col_idx.clone(), batch.graph.membership.clone(),
                         batch.graph.win_edges[:batch.graph.n_edges].clone(), y_t.clone(), w_t.clone()))
        torch.cuda.synchronize()
        out[resident] = rows
    assert len(out[True]) == len(out[False]) == 6
    for ra, rb in zip(out[True], out[False]):
        for x, z in zip(ra, rb):
            assert torch.equal(x, z)


@pytest.mark.gpu
def test_labels_uploaded_by_one_fit_serve_the_next_until_they_change():
    from deepchem_amd.data.packed_dataset import DeviceBatchPipeline
    dev = torch.device("cuda:0")
    packed = _pack(4, n=100, n_feat=75)
    n = packed.n_mols
    y = (np.random.RandomState(0).rand(n, 3) < 0.5).astype(np.float64)
    w = np.ones((n, 3))
    one_hot = lambda a: np.stack([1 - a, a], axis=-1)
    p1 = DeviceBatchPipeline(packed, y, w, [], dev, one_hot, label_key=("one_hot", 3, 2))
    p2 = DeviceBatchPipeline(packed, y, w, [], dev, one_hot, label_key=("one_hot", 3, 2))
    assert p2.y_dev is p1.y_dev and p2.w_dev is p1.w_dev
    assert torch.equal(p1.y_dev.cpu(), torch.as_tensor(one_hot(y).astype(np.float32)))
    y[0, 0] = 1 - y[0, 0]                                  # edited in place: the fingerprint of the bytes changes
    p3 = DeviceBatchPipeline(packed, y, w, [], dev, one_hot, label_key=("one_hot", 3, 2))
    assert p3.y_dev is not p1.y_dev and p3.w_dev is p1.w_dev
    assert torch.equal(p3.y_dev.cpu(), torch.as_tensor(one_hot(y).astype(np.float32)))
    p4 = DeviceBatchPipeline(packed, y, w, [], dev, None)  # another transform of the same labels: its own copy
    assert p4.y_dev.shape == (n, 3)
    p5 = DeviceBatchPipeline(packed, y, w, [], dev, one_hot)  # a transform nobody named is never reused
    assert p5.y_dev is not p3.y_dev


@pytest.mark.gpu
def test_device_collation_at_the_benchmark_batch_size():
    """65 536 shuffled molecules out of 100 000 (BASELINE.json's batch): every array equal to the host collation's."""
    from deepchem_amd.data.collate import collate_to_device
    from deepchem_amd.data.resident import ResidentMolSet
    dev = torch.device("cuda:0")
    packed = synthetic_molecules(100000, seed=21, n_feat=75)
    rset = ResidentMolSet(packed, dev)
    sel = np.random.RandomState(8).permutation(packed.n_mols)[:65536]
    a = rset.collate(sel)
    b = collate_to_device(packed, sel, dev)
    torch.cuda.synchronize()
    assert a.graph.n_atoms == b.graph.n_atoms > 10 ** 6
    assert torch.equal(a.atom_features, b.atom_features)
    for name in ("col_idx", "membership", "mol_runs", "rev_pos", "win_meta"):
        assert torch.equal(getattr(a.graph, name), getattr(b.graph, name)), name
    n = int((a.graph.win_meta.view(-1, _lib.GCMI_WIN_META_INTS)[:, 2 * ND] +
             (a.graph.win_meta.view(-1, _lib.GCMI_WIN_META_INTS)[:, 2 * ND + 1] + 7) // 8 * 8).max().item())
    assert torch.equal(a.graph.win_edges[:n], b.graph.win_edges[:n])


@pytest.mark.gpu
def test_fit_over_the_resident_set_equals_fit_over_host_collation(monkeypatch):
    """GraphConvModel.fit with batches of 1 024 shuffled molecules: collated by the GPU (the default from 512 molecules
    per batch up) and on the host (GCMI_RESIDENT_SET=0).  Same seeds -> the same batches -> predictions before are
    identical, after two optimiser steps equal up to the float atomics of the weight gradients."""
    import deepchem_amd as dc
    from deepchem_amd.models.torch_models import GraphConvModel
    packed = synthetic_molecules(2048, seed=31, n_feat=75)
    y = np.random.RandomState(4).randn(packed.n_mols, 2)
    w = np.ones_like(y)
    first, after = [], []
    for env in ("1", "0"):
        monkeypatch.setenv("GCMI_RESIDENT_SET", env)
        packed.__dict__.pop("_resident_sets", None)
        torch.manual_seed(11)
        np.random.seed(11)
        # a large Adam epsilon keeps the first updates proportional to the gradient: with the default 1e-8 they are
        # +-lr whatever the gradient's size, and the float atomics of the weight gradients decide the sign of the
        # near-zero ones (two runs of the SAME configuration then end 1e-4 apart)
        model = GraphConvModel(2, number_input_features=[75, 64], batch_size=1024, mode="regression",
                               grad_mode="full", device=torch.device("cuda:0"),
                               optimizer=dc.models.optimizers.Adam(learning_rate=1e-3, epsilon=1e-2))
        ds = dc.data.PackedDataset(packed, y, w)
        first.append(model.predict(ds))
        model.fit(ds, nb_epoch=1, deterministic=False, checkpoint_interval=0)
        after.append(model.predict(ds))
        built = packed.__dict__.get("_resident_sets", {})
        assert (len(built) > 0 and all(v is not None for v in built.values())) == (env == "1")
    np.testing.assert_array_equal(first[0], first[1])
    moved = np.abs(after[0] - first[0]).max()
    apart = np.abs(after[0] - after[1]).max()
    print("moved %.3e apart %.3e" % (moved, apart))
    assert apart < 2e-5 and moved > 1e-3


@pytest.mark.gpu
def test_device_collation_with_unpadded_feature_rows():
    """Rows of 75 floats (no padding to a multiple of four): the element-wise copy launch instead of the 16-byte one."""
    from deepchem_amd.data.collate import collate_to_device
    from deepchem_amd.data.resident import ResidentMolSet
    dev = torch.device("cuda:0")
    packed = _pack(13, n=500, n_feat=75)
    rset = ResidentMolSet(packed, dev, pad_features_to=1)
    assert rset.ld == 75
    sel = np.random.RandomState(6).permutation(packed.n_mols)
    a = rset.collate(sel)
    b = collate_to_device(packed, sel, dev, pad_features_to=1)
    torch.cuda.synchronize()
    assert a.atom_features.shape == b.atom_features.shape == (a.graph.n_atoms, 75)
    assert torch.equal(a.atom_features, b.atom_features)
    assert torch.equal(a.graph.col_idx, b.graph.col_idx) and torch.equal(a.graph.membership, b.graph.membership)


@pytest.mark.gpu
def test_one_sided_sets_stay_on_the_host_collation():
    """A set that lists some bond from one end only cannot use the reverse-slot tables: the pipeline collates it on the
    host (atomic backward kernels) when left to choose, and says so when the resident set is demanded."""
    from deepchem_amd.data.packed_dataset import DeviceBatchPipeline
    dev = torch.device("cuda:0")
    good = synthetic_molecules(600, seed=2, n_feat=75)
    adj_ptr = np.array([0, 1, 1], np.int64)  # atom 0 -> atom 1, never back
    lone = PackedMols(np.zeros((2, 75), np.float32), np.array([0, 2], np.int64), adj_ptr, np.array([1], np.int32))
    packed = concat_packed([good, lone])
    n = packed.n_mols
    y, w = np.zeros((n, 1)), np.ones((n, 1))
    batches = [(np.arange(n, dtype=np.int64), n)]
    pipe = DeviceBatchPipeline(packed, y, w, batches, dev)
    out = list(pipe)
    assert len(out) == 1 and out[0][0].graph.n_mols == n and pipe.resident is None
    assert out[0][0].graph.rev_pos is None       # not symmetric: no reverse slots attached
    with pytest.raises(ValueError, match="one end only"):
        DeviceBatchPipeline(packed, y, w, batches, dev, resident=True)

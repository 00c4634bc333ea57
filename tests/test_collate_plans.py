"""CPU checks of the plans gcmi_collate_plans emits beside the reference layout
(ConvMol.agglomerate_mols, feat/mol_graphs.py:256-349): reverse edge slots and the LDS
windows.  Integer work: every property is exact."""
import numpy as np
import pytest

from deepchem_amd.data.collate import collate_host
from deepchem_amd.feat.mol_graphs import collate_packed
from deepchem_amd.utils.synthetic import (concat_packed, single_atom_and_edge_cases,
                                          synthetic_molecules)


def _pack(seed=0, n=200):
    return concat_packed([synthetic_molecules(n, seed=seed, n_feat=8),
                          single_atom_and_edge_cases(8, seed),
                          synthetic_molecules(6, seed=seed + 1, n_feat=8, mean_atoms=14, max_atoms=40,
                                              parent_weights=(1,) * 10, ring_deg=10, ring_p_deg3=1.0,
                                              rings_per_atom=0.8)])


def _row_ptr(hb):
    deg_start = np.concatenate([[0], np.cumsum(hb.deg_counts)])
    edge_start = np.concatenate([[0], np.cumsum(np.asarray(hb.deg_counts) * np.arange(hb.n_deg))])
    deg = np.repeat(np.arange(hb.n_deg), hb.deg_counts)
    rows = np.arange(hb.n_atoms)
    ptr = edge_start[deg] + (rows - deg_start[deg]) * deg
    return deg, ptr, deg_start


@pytest.mark.parametrize("win_cap", [32, 128, 1000])
def test_plans_match_layout(win_cap):
    packed = _pack(3)
    hb = collate_host(packed, None, win_cap=win_cap, pin=False)
    multi = collate_packed(packed)
    # the reference layout itself
    assert np.array_equal(hb.part("membership").numpy(), multi.membership)
    feats = hb.part("features").numpy()
    assert np.array_equal(feats[:, :8], multi.get_atom_features().astype(np.float32))
    col = hb.part("col_idx").numpy()
    ref_col = np.concatenate([a.reshape(-1) for a in multi.get_deg_adjacency_lists()[1:]])
    assert np.array_equal(col, ref_col)
    deg, ptr, deg_start = _row_ptr(hb)
    src = np.repeat(np.arange(hb.n_atoms), deg)         # owner row of every edge slot
    # reverse slots: the slot rev[e] of the neighbour's list points back at the owner
    assert hb.symmetric
    rev = hb.part("rev_pos").numpy().astype(np.int64)
    assert (rev < deg[col]).all()
    assert np.array_equal(col[ptr[col] + rev], src)
    # ... and the pairing is an involution (slot pairs, also for multi-bonds)
    back = ptr[col] + rev
    slot_in_owner = np.arange(hb.n_edges) - ptr[src]
    assert np.array_equal(rev[back], slot_in_owner)
    # windows: a partition of every degree block into consecutive row ranges
    ND = hb.n_deg
    assert ND == 11
    meta = hb.part("win_meta").numpy().reshape(hb.n_win, 24)
    sb = np.concatenate([np.zeros((hb.n_win, 1), np.int64), meta[:, 11:22]], 1)      # slot starts
    cnt = np.diff(sb, axis=1)
    begin = meta[:, :11] + sb[:, :11]
    size = sb[:, 11]
    assert (cnt >= 0).all() and (size > 0).all()
    # ordinary windows first, oversized ones (a single molecule above the cap) last
    n_norm = hb.n_win - hb.n_win_big
    assert (size[:n_norm] <= win_cap).all() and (size[n_norm:] > win_cap).all()
    assert size[:n_norm].max() == hb.win_alloc
    assert (size[n_norm:].max() if hb.n_win_big else 0) == hb.win_alloc_big
    n_mol_atoms = np.bincount(multi.membership)
    assert hb.n_win_big == int((n_mol_atoms > win_cap).sum())
    # in batch order the windows tile every degree block
    order = np.argsort(begin.sum(1), kind="stable")
    begin, cnt, sb, size, meta = begin[order], cnt[order], sb[order], size[order], meta[order]
    for d in range(ND):
        assert begin[0, d] == deg_start[d]
        assert np.array_equal(begin[1:, d], begin[:-1, d] + cnt[:-1, d])
        assert begin[-1, d] + cnt[-1, d] == deg_start[d + 1]
    # edge entries: window-major, padded to 8, slot | rev << 12
    ne = meta[:, 23]
    eoff = meta[:, 22]
    assert np.array_equal(ne, (cnt * np.arange(ND)).sum(1))
    padded = (ne + 7) // 8 * 8
    assert np.array_equal(eoff, np.concatenate([[0], np.cumsum(padded)[:-1]]))
    assert padded.max() == max(hb.win_ecap, hb.win_ecap_big)
    ent = hb.part("win_edges").numpy().astype(np.uint16).astype(np.int64)
    win_of_row = np.empty(hb.n_atoms, np.int64)
    slot_of_row = np.empty(hb.n_atoms, np.int64)
    ent_of_edge = np.empty(hb.n_edges, np.int64)     # global edge slot -> position in win_edges
    for w in range(hb.n_win):
        eb = 0
        for d in range(ND):
            b, c = begin[w, d], cnt[w, d]
            win_of_row[b:b + c] = w
            slot_of_row[b:b + c] = sb[w, d] + np.arange(c)
            if d:
                g0 = ptr[b] if c else 0
                ent_of_edge[g0:g0 + c * d] = eoff[w] + eb + np.arange(c * d)
            eb += c * d
        assert (ent[eoff[w] + ne[w]:eoff[w] + padded[w]] == 0).all()
    assert np.array_equal(win_of_row[col], win_of_row[src])      # neighbours share the window
    assert np.array_equal(ent[ent_of_edge] & 4095, slot_of_row[col])
    assert np.array_equal(ent[ent_of_edge] >> 12, rev)
    # molecules are never split across windows
    mem = multi.membership
    first = {}
    for r in range(hb.n_atoms):
        assert first.setdefault(int(mem[r]), int(win_of_row[r])) == win_of_row[r]


def test_asymmetric_adjacency_is_flagged():
    from deepchem_amd.utils.synthetic import PackedMols
    feats = np.zeros((3, 8), np.float32)
    packed = PackedMols(feats, np.array([0, 3]), np.array([0, 1, 2, 2]), np.array([1, 0], np.int32))
    assert collate_host(packed, None, pin=False).symmetric
    packed = PackedMols(feats, np.array([0, 3]), np.array([0, 1, 2, 3]), np.array([1, 0, 0], np.int32))
    assert not collate_host(packed, None, pin=False).symmetric


def test_selection_order_and_thread_independence():
    packed = _pack(5, n=3000)  # enough molecules for the threaded path
    sel = np.random.RandomState(0).permutation(packed.n_mols)[:2500]
    a = collate_host(packed, sel, pin=False)
    b = collate_host(packed, sel, pin=False)
    for name in ("membership", "col_idx", "mol_runs", "win_meta", "rev_pos"):
        assert np.array_equal(a.part(name).numpy(), b.part(name).numpy()), name
    ref = collate_packed(packed.select(sel))
    assert np.array_equal(a.part("membership").numpy(), ref.membership)
    assert np.array_equal(a.part("col_idx").numpy(),
                          np.concatenate([t.reshape(-1) for t in ref.get_deg_adjacency_lists()[1:]]))

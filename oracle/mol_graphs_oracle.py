"""ORACLE (test infrastructure, not product code): batch collation.

Definitional restatement of the reference's host-side graph collation, written
as plain Python over tuples so that it is obviously right on small inputs:

* ``conv_mol``      follows deepchem/feat/mol_graphs.py:48-98, :113-185
  (``ConvMol.__init__`` / ``_deg_sort``): atoms of one molecule in order of
  (degree, original index); neighbour ids renamed to the new positions, row
  order of each neighbour list kept.
* ``agglomerate``   follows deepchem/feat/mol_graphs.py:256-349
  (``ConvMol.agglomerate_mols``): batch order (degree, molecule, in-molecule
  sorted position); ``membership``; per-degree ``(n_d, d)`` int32 tables of
  batch positions; ``deg_slice[d] = (running start, size)``.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Pinned by: the known answers of the reference's
deepchem/feat/tests/test_mol_graphs.py:21-142 (tests/test_oracle_golden.py) and
by tests/golden/collate_*.npz generated from the reference itself
(oracle/gen_golden.py).
"""
import numpy as np


def conv_mol(atom_features, adj_list, max_deg=10, min_deg=0):
    atom_features = np.asarray(atom_features)
    n = atom_features.shape[0]
    deg = [len(nb) for nb in adj_list]
    for d in deg:
        if d > max_deg or d < min_deg:
            raise ValueError("degree out of range")
    new_order = sorted(range(n), key=lambda i: (deg[i], i))
    new_pos = {old: new for new, old in enumerate(new_order)}
    feats = atom_features[new_order, :] if n else atom_features
    adj = [[new_pos[j] for j in adj_list[old]] for old in new_order]
    sdeg = [deg[old] for old in new_order]
    n_deg = max_deg - min_deg + 1
    tables = []
    deg_slice = np.zeros((n_deg, 2), np.int32)
    start = 0
    for k in range(n_deg):
        d = k + min_deg
        rows = [adj[i] for i in range(n) if sdeg[i] == d]
        tables.append(np.array(rows, dtype=np.int32).reshape(len(rows), d))
        deg_slice[k] = (start if rows else 0, len(rows))  # start zeroed when empty (:184)
        start += len(rows)
    return dict(atom_features=feats, adj=adj, deg=sdeg, deg_adj_lists=tables,
                deg_slice=deg_slice, n_atoms=n)


def agglomerate(mols, max_deg=10, min_deg=0):
    """``mols``: list of dicts from :func:`conv_mol`."""
    n_deg = max_deg - min_deg + 1
    keys = []  # (degree, molecule, position in molecule)
    for m, mol in enumerate(mols):
        for i in range(mol["n_atoms"]):
            keys.append((mol["deg"][i], m, i))
    keys.sort()
    batch_pos = {(m, i): p for p, (_, m, i) in enumerate(keys)}
    n_total = len(keys)
    if n_total:
        feats = np.stack([mols[m]["atom_features"][i] for (_, m, i) in keys])
    else:
        feats = np.zeros((0, 0))
    membership = np.array([m for (_, m, _) in keys], dtype=np.int32)
    tables = []
    deg_slice = np.zeros((n_deg, 2), np.int64)
    start = 0
    for k in range(n_deg):
        d = k + min_deg
        rows = [[batch_pos[(m, j)] for j in mols[m]["adj"][i]] for (dd, m, i) in keys if dd == d]
        tables.append(np.array(rows, dtype=np.int32).reshape(len(rows), d))
        deg_slice[k] = (start, len(rows))  # running start, NOT zeroed (:300-305)
        start += len(rows)
    return dict(atom_features=feats, deg_adj_lists=tables, deg_slice=deg_slice,
                membership=membership, num_mols=len(mols), num_atoms=n_total)

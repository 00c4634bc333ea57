"""SMILES -> ConvMol / WeaveMol featurizers on the native reader.

Mirror of the reference's ``ConvMolFeaturizer`` / ``WeaveFeaturizer`` (deepchem/feat/graph_features.py:698-914,
:921-1078) for SMILES strings.  The reference parses with rdkit; here ``gcmi_smiles_sizes`` /
``gcmi_smiles_featurize`` (csrc/featurize.cpp, host C++ with worker threads) read the SMILES and fill flat
arrays.  ``featurize`` returns the same objects the reference does (``ConvMol`` / ``WeaveMol``, an empty array
where a molecule cannot be read -- feat/base_classes.py:318-328); ``featurize_packed`` skips the per-molecule
objects and returns the ``PackedMols`` the native collation reads, which is what ``fit`` wants for big datasets.

Differences from the reference, by construction: atoms keep their SMILES order (the reference renumbers by
rdkit's canonical ranking; the models are invariant to it); options that need more of rdkit than a SMILES reader
(``use_chirality``, ``atom_properties``, euclidean pair distances, ``max_pair_distance``) raise.
Parity with rdkit is pinned only by the vectors listed in oracle/smiles_oracle.py (PARITY UNPINNED otherwise):
with rdkit installed, DeepChem's own featurizers produce objects every model here accepts.
"""
import ctypes
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from deepchem_amd import _lib
from deepchem_amd.feat.mol_graphs import ConvMol
from deepchem_amd.utils.synthetic import PackedMols

ATOM_FEATURES = 75
BOND_FEATURES = 6
PAIR_FEATURES = 14


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _default_threads() -> int:
    return max(1, min(16, (os.cpu_count() or 1)))


def read_smiles(smiles: Sequence[str], atoms: bool = True, adjacency: bool = True, bonds: bool = False,
                pairs: bool = False, props: bool = False, n_threads: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Flat arrays for a list of SMILES.  Always: ``valid`` (bool per molecule), ``n_atoms``, ``n_bonds``,
    ``atom_off`` / ``bond_off`` (prefix sums, zero-width for unreadable molecules).  Optional blocks:
    ``atom_features`` (A,75); ``adj_degree`` (A,), ``adj_idx`` (2B,); ``bond_atoms`` (B,2), ``bond_features``
    (B,6); ``pair_off``, ``pair_features`` (sum n^2,14); ``atom_props`` (A,8)."""
    lib = _lib.load()
    smiles = list(smiles)
    for s in smiles:
        if not isinstance(s, str):
            raise TypeError("read_smiles takes SMILES strings; got %r" % type(s))
    n = len(smiles)
    n_threads = n_threads or _default_threads()
    raw = [s.encode("utf-8") for s in smiles]
    arr = (ctypes.c_char_p * max(n, 1))(*raw)
    n_atoms = np.zeros(n, np.int32)
    n_bonds = np.zeros(n, np.int32)
    _lib.check(lib.gcmi_smiles_sizes(arr, n, n_atoms.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                     n_bonds.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n_threads),
               "gcmi_smiles_sizes")
    valid = n_atoms >= 0
    na = np.where(valid, n_atoms, 0).astype(np.int64)
    atom_off = np.zeros(n + 1, np.int64)
    np.cumsum(na, out=atom_off[1:])
    bond_off = np.zeros(n + 1, np.int64)
    np.cumsum(n_bonds.astype(np.int64), out=bond_off[1:])
    A, B = int(atom_off[-1]), int(bond_off[-1])
    out: Dict[str, np.ndarray] = {"valid": valid, "n_atoms": na.astype(np.int32), "n_bonds": n_bonds,
                                  "atom_off": atom_off, "bond_off": bond_off}
    if atoms:
        out["atom_features"] = np.empty((A, ATOM_FEATURES), np.float32)
    if adjacency:
        out["adj_degree"] = np.empty(A, np.int32)
        out["adj_idx"] = np.empty(2 * B, np.int32)
    if bonds:
        out["bond_atoms"] = np.empty((B, 2), np.int32)
        out["bond_features"] = np.empty((B, BOND_FEATURES), np.float32)
    pair_off = None
    if pairs:
        pair_off = np.zeros(n + 1, np.int64)
        np.cumsum(na * na, out=pair_off[1:])
        out["pair_off"] = pair_off
        out["pair_features"] = np.empty((int(pair_off[-1]), PAIR_FEATURES), np.float32)
    if props:
        out["atom_props"] = np.empty((A, 8), np.int32)
    _lib.check(lib.gcmi_smiles_featurize(arr, n, atom_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                         bond_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _ptr(pair_off),
                                         _ptr(out.get("atom_features")), _ptr(out.get("adj_degree")),
                                         _ptr(out.get("adj_idx")), _ptr(out.get("bond_atoms")),
                                         _ptr(out.get("bond_features")), _ptr(out.get("pair_features")),
                                         _ptr(out.get("atom_props")), n_threads), "gcmi_smiles_featurize")
    return out


def why_unreadable(smiles: str) -> Optional[str]:
    """None when the SMILES can be featurized, else the reason."""
    msg = _lib.load().gcmi_smiles_check(smiles.encode("utf-8"))
    return None if msg is None else msg.decode()


def _as_list(datapoints) -> List[str]:
    if isinstance(datapoints, str):
        return [datapoints]
    return list(datapoints)


class _SmilesFeaturizer(object):
    """The part of ``MolecularFeaturizer`` (feat/base_classes.py:209-330) that applies to SMILES input."""

    def __call__(self, datapoints, **kwargs):
        return self.featurize(datapoints, **kwargs)


class ConvMolFeaturizer(_SmilesFeaturizer):
    """``dc.feat.ConvMolFeaturizer`` for SMILES strings (graph_features.py:698-914)."""
    name = ['conv_mol']

    def __init__(self, master_atom: bool = False, use_chirality: bool = False, atom_properties: Iterable[str] = (),
                 per_atom_fragmentation: bool = False, n_threads: Optional[int] = None):
        if use_chirality or list(atom_properties):
            raise NotImplementedError("use_chirality / atom_properties read rdkit properties; featurize with "
                                      "DeepChem + rdkit and pass the ConvMol objects instead")
        self.dtype = object
        self.master_atom = master_atom
        self.use_chirality = False
        self.atom_properties: List[str] = []
        self.per_atom_fragmentation = per_atom_fragmentation
        self.n_threads = n_threads

    def feature_length(self) -> int:
        return ATOM_FEATURES

    def featurize_packed(self, smiles: Sequence[str]) -> Tuple[PackedMols, np.ndarray]:
        """(PackedMols of the readable molecules, their positions in ``smiles``)."""
        if self.master_atom or self.per_atom_fragmentation:
            raise NotImplementedError("featurize_packed covers the plain featurization; use featurize()")
        # the float rows are never built here: the property rows become 8-byte atom codes (feat/atom_codes.py),
        # which is what the collation moves and the GPU expands
        from deepchem_amd.feat.atom_codes import codes_from_props
        r = read_smiles(_as_list(smiles), atoms=False, props=True, n_threads=self.n_threads)
        keep = np.nonzero(r["valid"])[0]
        atom_ptr = np.zeros(keep.shape[0] + 1, np.int64)
        np.cumsum(r["n_atoms"][keep], out=atom_ptr[1:])  # unreadable molecules own no rows, so offsets just close up
        adj_ptr = np.zeros(r["adj_degree"].shape[0] + 1, np.int64)
        np.cumsum(r["adj_degree"], out=adj_ptr[1:])
        return PackedMols(None, atom_ptr, adj_ptr, r["adj_idx"], codes_from_props(r["atom_props"])), keep

    def featurize(self, datapoints, log_every_n: int = 1000, **kwargs) -> np.ndarray:
        smiles = _as_list(datapoints)
        r = read_smiles(smiles, n_threads=self.n_threads)
        out = np.empty(len(smiles), dtype=object)
        adj_ptr = np.zeros(r["adj_degree"].shape[0] + 1, np.int64)
        np.cumsum(r["adj_degree"], out=adj_ptr[1:])
        feats = r["atom_features"].astype(np.float64)  # the reference's node matrix is float64 (:887)
        for i in range(len(smiles)):
            if not r["valid"][i]:
                out[i] = np.array([])
                continue
            a0, a1 = int(r["atom_off"][i]), int(r["atom_off"][i + 1])
            nodes = feats[a0:a1]
            ptr = adj_ptr[a0:a1 + 1]
            adj = [r["adj_idx"][ptr[k]:ptr[k + 1]].tolist() for k in range(a1 - a0)]
            if self.master_atom:
                nodes = np.concatenate([nodes, nodes.mean(axis=0, keepdims=True)], axis=0)
                fake = a1 - a0
                adj = [nb + [fake] for nb in adj] + [[]]
            if self.per_atom_fragmentation:
                out[i] = [ConvMol(n_, a_) for n_, a_ in _per_atom_fragments(nodes, adj)]
            else:
                out[i] = ConvMol(nodes, adj)
        if self.per_atom_fragmentation:
            out = np.array([m for m in out if isinstance(m, list) and len(m)], dtype=object)
        return out

    def __hash__(self):
        return hash((self.master_atom, self.use_chirality, tuple(self.atom_properties)))

    def __eq__(self, other):
        return (isinstance(other, ConvMolFeaturizer) and self.master_atom == other.master_atom
                and self.use_chirality == other.use_chirality)


def _per_atom_fragments(nodes: np.ndarray, adj: List[List[int]]):
    """Every molecule-minus-one-atom (graph_features.py:851-877)."""
    for i in range(nodes.shape[0]):
        new_n = np.delete(nodes, i, axis=0)
        new_a = [[v if v < i else v - 1 for v in nb if v != i] for j, nb in enumerate(adj) if j != i]
        yield new_n, new_a


class WeaveFeaturizer(_SmilesFeaturizer):
    """``dc.feat.WeaveFeaturizer`` for SMILES strings, all pairs (graph_features.py:921-1078)."""
    name = ['weave_mol']

    def __init__(self, graph_distance: bool = True, explicit_H: bool = False, use_chirality: bool = False,
                 max_pair_distance: Optional[int] = None, n_threads: Optional[int] = None):
        if not graph_distance or explicit_H or use_chirality:
            raise NotImplementedError("euclidean distances / explicit hydrogens / chirality need rdkit")
        if max_pair_distance is not None:
            if isinstance(max_pair_distance, int) and max_pair_distance <= 0:
                raise ValueError("max_pair_distance must either be a positive integer or None")
            raise NotImplementedError("max_pair_distance is not implemented; all pairs are produced")
        self.graph_distance = True
        self.dtype = object
        self.explicit_H = False
        self.use_chirality = False
        self.max_pair_distance = None
        self.bt_len = BOND_FEATURES
        self.n_threads = n_threads

    def featurize(self, datapoints, log_every_n: int = 1000, **kwargs) -> np.ndarray:
        from deepchem_amd.models.torch_models.weavemodel_pytorch import WeaveMol
        smiles = _as_list(datapoints)
        r = read_smiles(smiles, adjacency=False, pairs=True, n_threads=self.n_threads)
        out = np.empty(len(smiles), dtype=object)
        feats = r["atom_features"].astype(np.float64)
        for i in range(len(smiles)):
            if not r["valid"][i]:
                out[i] = np.array([])
                continue
            a0, a1 = int(r["atom_off"][i]), int(r["atom_off"][i + 1])
            n = a1 - a0
            pairs = r["pair_features"][r["pair_off"][i]:r["pair_off"][i + 1]].astype(np.float64)
            edges = np.stack([np.repeat(np.arange(n), n), np.tile(np.arange(n), n)])
            out[i] = WeaveMol(feats[a0:a1], pairs, edges)
        return out

"""Seeded synthetic molecule sets with Tox21-like statistics.

There is no rdkit and no network on the build or GPU boxes, so the benchmark
and the parity tests run on synthetic molecular graphs whose size and degree
statistics follow the recipe of SURVEY.md section 8(d):

* atoms per molecule  n ~ clip(round(LogNormal(ln m, 0.5)), 2, n_max)
* bonds: a random spanning tree (parent drawn among earlier atoms, weighted
  by degree) plus ~n/10 ring-closure bonds between atoms of degree
  < ``ring_deg`` -- tuned to E/N ~= 2.08 and a degree mix of
  {1: 25 %, 2: 44 %, 3: 28 %, 4: 3 %} (Tox21: 24/46/27/3 %)
* atom features: 75 Bernoulli(0.1) columns stored as float32
* labels Bernoulli(0.075) (classification) or N(0,1) (regression); weights 1
  with 17 % zeros.

The result is a :class:`PackedMols`: every molecule of the set in four flat
arrays (molecule-major, original atom order, molecule-local neighbour ids).
That is also the on-disk / in-memory molecule-set format the native collation
(``gcmi_collate``) consumes, so a set of 10^5 molecules never exists as 10^5
Python objects.
"""
from typing import List, Optional, Tuple

import numpy as np


class PackedMols:
    """A set of molecular graphs in four flat arrays.

    atom_features : (A, F) float32, molecule-major, original atom order
    atom_ptr      : (M+1,) int64, molecule m owns atoms atom_ptr[m]:atom_ptr[m+1]
    adj_ptr       : (A+1,) int64, CSR row pointer over atoms
    adj_idx       : (nnz,) int32, neighbour ids LOCAL to the molecule
    atom_codes    : optional (A, 8) uint8 (deepchem_amd/feat/atom_codes.py): when the feature rows are the
                    75-column one-hot rows of the reference's featurizer they are kept as 8-byte codes; the float
                    rows are then produced on demand (``atom_features``, ``molecule``) and, in the training
                    pipeline, on the GPU.
    """

    def __init__(self, atom_features: Optional[np.ndarray], atom_ptr: np.ndarray, adj_ptr: np.ndarray,
                 adj_idx: np.ndarray, atom_codes: Optional[np.ndarray] = None):
        if atom_features is None and atom_codes is None:
            raise ValueError("PackedMols needs atom_features or atom_codes")
        self._features = atom_features
        self.atom_ptr, self.adj_ptr, self.adj_idx = atom_ptr, adj_ptr, adj_idx
        self.atom_codes = None if atom_codes is None else np.ascontiguousarray(atom_codes, np.uint8).reshape(-1, 8)

    def __getstate__(self):
        # the copies the training pipeline keeps in HBM (resident set, labels) belong to this process and device
        state = dict(self.__dict__)
        state.pop("_resident_sets", None)
        state.pop("_label_cache", None)
        return state

    @property
    def atom_features(self) -> np.ndarray:
        if self._features is not None:
            return self._features
        from deepchem_amd.feat.atom_codes import features_from_codes
        return features_from_codes(self.atom_codes)

    @property
    def n_mols(self) -> int:
        return int(self.atom_ptr.shape[0] - 1)

    @property
    def n_atoms(self) -> int:
        return int(self.adj_ptr.shape[0] - 1)

    @property
    def n_feat(self) -> int:
        return 75 if self._features is None else int(self._features.shape[1])

    def features_of(self, a0: int, a1: int) -> np.ndarray:
        if self._features is not None:
            return self._features[a0:a1]
        from deepchem_amd.feat.atom_codes import features_from_codes
        return features_from_codes(self.atom_codes[a0:a1])

    def drop_float_features(self) -> "PackedMols":
        """Keep only the codes (when there are codes)."""
        if self.atom_codes is not None:
            self._features = None
        return self

    def molecule(self, m: int) -> Tuple[np.ndarray, List[List[int]]]:
        """(atom_features (n,F), adj_list) of molecule ``m`` -- the two
        arguments of ``ConvMol(atom_features, adj_list)``."""
        a0, a1 = int(self.atom_ptr[m]), int(self.atom_ptr[m + 1])
        feats = self.features_of(a0, a1)
        adj = [
            self.adj_idx[self.adj_ptr[a]:self.adj_ptr[a + 1]].tolist()
            for a in range(a0, a1)
        ]
        return feats, adj

    def select(self, idx: np.ndarray) -> "PackedMols":
        """Sub-set (with repetition allowed) in the order given by ``idx``."""
        idx = np.asarray(idx, dtype=np.int64)
        a0 = self.atom_ptr[idx]
        n = self.atom_ptr[idx + 1] - a0
        new_ptr = np.zeros(idx.shape[0] + 1, np.int64)
        np.cumsum(n, out=new_ptr[1:])
        # atom gather list
        rep = np.repeat(a0 - new_ptr[:-1], n)
        atoms = np.arange(new_ptr[-1], dtype=np.int64) + rep
        deg = (self.adj_ptr[atoms + 1] - self.adj_ptr[atoms])
        new_adj_ptr = np.zeros(atoms.shape[0] + 1, np.int64)
        np.cumsum(deg, out=new_adj_ptr[1:])
        erep = np.repeat(self.adj_ptr[atoms] - new_adj_ptr[:-1], deg)
        edges = np.arange(new_adj_ptr[-1], dtype=np.int64) + erep
        return PackedMols(None if self._features is None else self._features[atoms], new_ptr, new_adj_ptr,
                          self.adj_idx[edges], None if self.atom_codes is None else self.atom_codes[atoms])


def _gen_chunk(rng: np.random.RandomState, sizes: np.ndarray,
               parent_weights: Tuple[float, ...], ring_deg: int,
               ring_p_deg3: float, rings_per_atom: float):
    """Topology for one chunk of molecules; returns (deg[M,nmax], nbr[M,nmax,ring_deg]).

    Spanning tree: atom k bonds to an earlier atom drawn with weight
    ``parent_weights[degree]`` (weighted draw = argmax of u**(1/w)); the
    weights favour chain extension, which is what gives real molecules their
    46 % share of degree-2 atoms.  Ring closures: floor(n*rings_per_atom + u)
    extra bonds between distinct non-bonded atoms of degree < ring_deg, a
    degree-3 end being accepted with probability ring_p_deg3.
    """
    M = sizes.shape[0]
    nmax = int(sizes.max())
    deg = np.zeros((M, nmax), np.int16)
    nbr = np.full((M, nmax, ring_deg), -1, np.int16)
    wt = np.zeros(ring_deg + 1)
    wt[:min(len(parent_weights), ring_deg)] = parent_weights[:ring_deg]
    for k in range(1, nmax):
        act = np.nonzero(sizes > k)[0]
        if act.size == 0:
            break
        w = wt[deg[act, :k]]
        r = rng.random_sample((act.size, k))**(1.0 / np.maximum(w, 1e-9))
        r[w <= 0] = -1.0
        parent = np.argmax(r, axis=1)
        slot = deg[act, parent]
        nbr[act, parent, slot] = k
        deg[act, parent] += 1
        nbr[act, k, 0] = parent
        deg[act, k] = 1
    n_rings = np.floor(sizes * rings_per_atom + rng.random_sample(M)).astype(np.int64)
    for r_i in range(int(n_rings.max()) if M else 0):
        act = np.nonzero(n_rings > r_i)[0]
        for _attempt in range(4):
            if act.size == 0:
                break
            n_act = sizes[act]
            a = np.minimum((rng.random_sample(act.size) * n_act).astype(np.int64), n_act - 1)
            b = np.minimum((rng.random_sample(act.size) * n_act).astype(np.int64), n_act - 1)
            da = deg[act, a]
            db = deg[act, b]
            ok = (a != b) & (da < ring_deg) & (db < ring_deg)
            ok &= ~((da >= 3) & (rng.random_sample(act.size) > ring_p_deg3))
            ok &= ~((db >= 3) & (rng.random_sample(act.size) > ring_p_deg3))
            ok &= ~(nbr[act, a, :] == b[:, None].astype(np.int16)).any(axis=1)
            m_ok, a_ok, b_ok = act[ok], a[ok], b[ok]
            nbr[m_ok, a_ok, deg[m_ok, a_ok]] = b_ok
            deg[m_ok, a_ok] += 1
            nbr[m_ok, b_ok, deg[m_ok, b_ok]] = a_ok
            deg[m_ok, b_ok] += 1
            act = act[~ok]
    return deg, nbr


def synthetic_molecules(n_mols: int,
                        mean_atoms: float = 18.5,
                        max_atoms: int = 132,
                        n_feat: int = 75,
                        seed: int = 0,
                        parent_weights: Tuple[float, ...] = (1.0, 3.3, 0.8, 0.15),
                        ring_deg: int = 4,
                        ring_p_deg3: float = 0.3,
                        rings_per_atom: float = 0.1,
                        min_atoms: int = 2,
                        single_atom_frac: float = 0.01,
                        feature_p: float = 0.1,
                        chunk: int = 4096) -> PackedMols:
    """Seeded random molecule set (see module docstring for the recipe)."""
    rng = np.random.RandomState(seed)
    # mu = ln(m) - sigma^2/2 so that the MEAN (not the median) is mean_atoms
    sizes = np.clip(
        np.rint(rng.lognormal(np.log(mean_atoms) - 0.125, 0.5, size=n_mols)),
        min_atoms, max_atoms).astype(np.int64)
    # lone atoms (ions, salts): the only source of degree-0 rows (Tox21: 0.05 % of atoms)
    sizes[rng.random_sample(n_mols) < single_atom_frac] = 1
    atom_ptr = np.zeros(n_mols + 1, np.int64)
    np.cumsum(sizes, out=atom_ptr[1:])
    A = int(atom_ptr[-1])
    deg_flat = np.empty(A, np.int64)
    idx_parts = []
    for c0 in range(0, n_mols, chunk):
        sz = sizes[c0:c0 + chunk]
        deg, nbr = _gen_chunk(rng, sz, parent_weights, ring_deg, ring_p_deg3,
                              rings_per_atom)
        valid = np.arange(deg.shape[1])[None, :] < sz[:, None]
        d = deg[valid].astype(np.int64)  # molecule-major order
        deg_flat[atom_ptr[c0]:atom_ptr[c0] + d.shape[0]] = d
        nb = nbr[valid]  # (atoms_in_chunk, ring_deg)
        keep = np.arange(ring_deg)[None, :] < d[:, None]
        idx_parts.append(nb[keep].astype(np.int32))
    adj_ptr = np.zeros(A + 1, np.int64)
    np.cumsum(deg_flat, out=adj_ptr[1:])
    adj_idx = np.concatenate(idx_parts) if idx_parts else np.zeros(0, np.int32)
    feats = (rng.random_sample((A, n_feat)) < feature_p).astype(np.float32)
    return PackedMols(feats, atom_ptr, adj_ptr, adj_idx)


def synthetic_labels(n_mols: int,
                     n_tasks: int,
                     mode: str = "classification",
                     seed: int = 0,
                     pos_rate: float = 0.075,
                     missing_rate: float = 0.17) -> Tuple[np.ndarray, np.ndarray]:
    """(y, w) of shape (n_mols, n_tasks), float64 like a DeepChem dataset holds them."""
    rng = np.random.RandomState(seed + 7919)
    if mode == "classification":
        y = (rng.random_sample((n_mols, n_tasks)) < pos_rate).astype(np.float64)
    else:
        y = rng.standard_normal((n_mols, n_tasks))
    w = (rng.random_sample((n_mols, n_tasks)) >= missing_rate).astype(np.float64)
    return y, w


def single_atom_and_edge_cases(n_feat: int = 75, seed: int = 0) -> PackedMols:
    """A tiny set that exercises the awkward shapes: an isolated atom
    (degree 0), a two-atom molecule, a chain, a ring, a star of degree 10
    and a clique-ish high-degree blob."""
    rng = np.random.RandomState(seed)
    adjs: List[List[List[int]]] = []
    adjs.append([[]])  # methane-like: one atom, degree 0
    adjs.append([[1], [0]])  # two atoms
    adjs.append([[1], [0, 2], [1]])  # chain of 3
    adjs.append([[1, 5], [0, 2], [1, 3], [2, 4], [3, 5], [4, 0]])  # ring of 6
    adjs.append([list(range(1, 11))] + [[0] for _ in range(10)])  # star, degree 10
    # 7 atoms, every atom bonded to the 6 others (degree 6)
    adjs.append([[j for j in range(7) if j != i] for i in range(7)])
    sizes = np.array([len(a) for a in adjs], np.int64)
    atom_ptr = np.zeros(len(adjs) + 1, np.int64)
    np.cumsum(sizes, out=atom_ptr[1:])
    deg = np.array([len(nb) for a in adjs for nb in a], np.int64)
    adj_ptr = np.zeros(deg.shape[0] + 1, np.int64)
    np.cumsum(deg, out=adj_ptr[1:])
    adj_idx = np.array([j for a in adjs for nb in a for j in nb], np.int32)
    feats = rng.standard_normal((int(atom_ptr[-1]), n_feat)).astype(np.float32)
    return PackedMols(feats, atom_ptr, adj_ptr, adj_idx)


def concat_packed(sets: List[PackedMols]) -> PackedMols:
    coded = all(s.atom_codes is not None for s in sets)
    floats = all(s._features is not None for s in sets)
    feats = np.concatenate([s.atom_features for s in sets]) if (floats or not coded) else None
    ap = [np.zeros(1, np.int64)]
    jp = [np.zeros(1, np.int64)]
    a_off = 0
    e_off = 0
    for s in sets:
        ap.append(s.atom_ptr[1:] + a_off)
        jp.append(s.adj_ptr[1:] + e_off)
        a_off += s.n_atoms
        e_off += int(s.adj_ptr[-1])
    return PackedMols(feats, np.concatenate(ap), np.concatenate(jp),
                      np.concatenate([s.adj_idx for s in sets]),
                      np.concatenate([s.atom_codes for s in sets]) if coded else None)

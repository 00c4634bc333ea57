"""CPU restatement of SMILES -> ConvMol / WeaveMol featurization (TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this).

What it restates
----------------
* ``atom_features``                 deepchem/feat/graph_features.py:282-391  (75 columns)
* ``bond_features``                 graph_features.py:394-459               (6 columns)
* ``pair_features`` / ``find_distance``  graph_features.py:532-695          (14 columns, max_pair_distance=None)
* ``ConvMolFeaturizer._featurize``  graph_features.py:845-914  (node matrix + adjacency lists)
* ``WeaveFeaturizer._featurize``    graph_features.py:1037-1078

The reference reads every chemical property from an rdkit ``Mol`` (``feat/base_classes.py:271-305``).  rdkit is
not in this image and is not part of ``/root/reference``, so the rules rdkit applies are restated here from its
published behaviour (``MolFromSmiles`` = parse, remove explicit hydrogens, clean up nitro-type groups, Kekulize, assign radicals,
perceive aromaticity, mark conjugation, set hybridisation):

* Kekulisation: a perfect matching of double bonds over the aromatic atoms that have a free valence.
* Valence: explicit valence from the Kekule form, checked against the element's largest allowed valence;
  implicit hydrogens only on non-bracket atoms (smallest allowed valence that fits).
* Radicals on bracket atoms from the octet rule (min of the "late" and "early" element counts).
* Rings: the relevant cycles (every cycle that is not a GF(2) sum of strictly smaller ones) -- what rdkit's
  symmetrised SSSR yields for ordinary molecules.
* Aromaticity: rdkit's default model (electron donor types, exocyclic bonds to more electronegative atoms take
  the electron, 4n+2 on single rings, then on fused pairs and triples along their outer bonds).
* Conjugation and hybridisation: rdkit's ``ConjugHybrid`` rules.

PARITY UNPINNED (no rdkit to generate vectors with).  Pinned pieces: the reference's own known answers for
``'C'``, ``'CCC'``, ``'C[N+](C)(C)C'`` (feat/tests/test_graph_features.py:14-104) and the hand-derived carbon
vectors of SURVEY.md 8c; everything else is textbook chemistry checked by hand in tests/test_oracle_smiles.py.
Atoms keep their order of appearance in the SMILES (the reference renumbers by rdkit's canonical ranking, a
row permutation the models are invariant to).

Written for clarity, not speed: brute-force cycle enumeration, so keep inputs to ordinary drug-size molecules.
"""
import itertools
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
           "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu "
           "Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr").split()
Z_OF = {s: i + 1 for i, s in enumerate(SYMBOLS)}

# allowed valences (-1 = anything goes); elements not listed: (-1,)
VALENCES = {1: (1,), 2: (0,), 3: (1, -1), 4: (2,), 5: (3,), 6: (4,), 7: (3,), 8: (2,), 9: (1,), 10: (0,),
            11: (1, -1), 12: (2, -1), 13: (3, 6), 14: (4, 6), 15: (3, 5, 7), 16: (2, 4, 6), 17: (1,), 18: (0,),
            19: (1, -1), 20: (2, -1), 31: (3,), 32: (4,), 33: (3, 5, 7), 34: (2, 4, 6), 35: (1,), 36: (0,),
            37: (1,), 38: (2,), 49: (3,), 50: (2, 4), 51: (3, 5, 7), 52: (2, 4, 6), 53: (1, 3, 5), 54: (0, 2, 4, 6),
            55: (1,), 56: (2,), 81: (3,), 82: (2, 4), 83: (3, 5, 7), 84: (2, 4, 6), 85: (1, 3, 5), 86: (0,)}


def valence_list(z: int) -> Tuple[int, ...]:
    return VALENCES.get(z, (-1,))


def default_valence(z: int) -> int:
    return valence_list(z)[0] if z >= 1 else -1


def n_outer_elecs(z: int) -> int:
    """Outer-shell electron count (s+p for main groups, s+d for the d block, 2 for Zn/Cd/Hg)."""
    if z <= 0:
        return 0
    if z <= 2:
        return z
    for start, length in ((3, 8), (11, 8)):
        if start <= z < start + length:
            return z - start + 1
    for start in (19, 37):
        if start <= z < start + 18:
            k = z - start + 1  # 1..18
            if k <= 2:
                return k
            if k <= 11:
                return k
            if k == 12:
                return 2
            return k - 10
    for start in (55, 87):
        if start <= z < start + 32:
            k = z - start + 1
            if k <= 2:
                return k
            if k <= 17:  # La..Lu / Ac..Lr
                return 3
            k -= 14  # now like the 18-wide rows
            if k <= 11:
                return k
            if k == 12:
                return 2
            return k - 10
    return 0


EARLY_ATOMS = {3, 4, 5, 11, 12, 13, 19, 20, 31, 37, 38, 49, 55, 56, 81}
AROMATIC_SYMBOLS = {"b": 5, "c": 6, "n": 7, "o": 8, "p": 15, "s": 16, "se": 34, "as": 33, "te": 52}
ORGANIC_SUBSET = {"B": 5, "C": 6, "N": 7, "O": 8, "P": 15, "S": 16, "F": 9, "Cl": 17, "Br": 35, "I": 53}

ATOM_SYMBOL_LIST = ['C', 'N', 'O', 'S', 'F', 'Si', 'P', 'Cl', 'Br', 'Mg', 'Na', 'Ca', 'Fe', 'As', 'Al', 'I', 'B', 'V',
                    'K', 'Tl', 'Yb', 'Sb', 'Sn', 'Ag', 'Pd', 'Co', 'Se', 'Ti', 'Zn', 'H', 'Li', 'Ge', 'Cu', 'Au', 'Ni',
                    'Cd', 'In', 'Mn', 'Zr', 'Cr', 'Pt', 'Hg', 'Pb', 'Unknown']
HYBRIDIZATIONS = ["SP", "SP2", "SP3", "SP3D", "SP3D2"]


class SmilesError(ValueError):
    pass


class Atom(object):

    def __init__(self, z, aromatic=False, charge=0, explicit_h=0, bracket=False, isotope=0):
        self.z, self.written_aromatic, self.charge = z, aromatic, charge
        self.explicit_h, self.bracket, self.isotope = explicit_h, bracket, isotope
        self.bonds: List[int] = []  # bond indices
        self.implicit_h = 0
        self.radicals = 0
        self.aromatic = False
        self.hybridization = "UNSPECIFIED"

    @property
    def symbol(self):
        return SYMBOLS[self.z - 1]


class Bond(object):

    def __init__(self, a, b, order, written_aromatic):
        self.a, self.b = a, b
        self.order = order  # 1, 2, 3; None while an aromatic bond waits for kekulisation
        self.written_aromatic = written_aromatic
        self.aromatic = False
        self.conjugated = False
        self.in_ring = False

    def other(self, i):
        return self.b if i == self.a else self.a


class Mol(object):

    def __init__(self):
        self.atoms: List[Atom] = []
        self.bonds: List[Bond] = []
        self.rings: List[Tuple[int, ...]] = []  # atom index tuples in ring order

    def neighbors(self, i) -> List[int]:
        return [self.bonds[b].other(i) for b in self.atoms[i].bonds]

    def degree(self, i) -> int:
        return len(self.atoms[i].bonds)

    def total_h(self, i) -> int:
        return self.atoms[i].implicit_h + self.atoms[i].explicit_h

    def explicit_valence(self, i) -> int:
        return sum(self.bonds[b].order for b in self.atoms[i].bonds) + self.atoms[i].explicit_h

    def total_valence(self, i) -> int:
        return self.explicit_valence(i) + self.atoms[i].implicit_h


# ----------------------------------------------------------------------------------------------- parsing

def _read_bracket(body: str) -> Atom:
    k = 0
    iso = ""
    while k < len(body) and body[k].isdigit():
        iso += body[k]
        k += 1
    if k >= len(body):
        raise SmilesError("empty bracket atom")
    aromatic = False
    if body[k:k + 2] in ("se", "as", "te"):
        z, aromatic, k = AROMATIC_SYMBOLS[body[k:k + 2]], True, k + 2
    elif body[k] in "bcnops":
        z, aromatic, k = AROMATIC_SYMBOLS[body[k]], True, k + 1
    else:
        sym = body[k]
        k += 1
        if k < len(body) and body[k].islower():
            sym += body[k]
            k += 1
        if sym not in Z_OF:
            raise SmilesError("unknown element %r" % sym)
        z = Z_OF[sym]
    if k < len(body) and body[k] == "@":
        k += 1
        if k < len(body) and body[k] == "@":
            k += 1
        elif body[k:k + 2] in ("TH", "AL", "SP", "TB", "OH"):
            k += 2
            while k < len(body) and body[k].isdigit():
                k += 1
    h = 0
    if k < len(body) and body[k] == "H":
        k += 1
        num = ""
        while k < len(body) and body[k].isdigit():
            num += body[k]
            k += 1
        h = int(num) if num else 1
    charge = 0
    if k < len(body) and body[k] in "+-":
        sign = 1 if body[k] == "+" else -1
        run = 0
        while k < len(body) and body[k] == ("+" if sign > 0 else "-"):
            run += 1
            k += 1
        num = ""
        while k < len(body) and body[k].isdigit():
            num += body[k]
            k += 1
        charge = sign * (int(num) if num else run)
    if k < len(body) and body[k] == ":":
        k += 1
        while k < len(body) and body[k].isdigit():
            k += 1
    if k != len(body):
        raise SmilesError("cannot read bracket atom [%s]" % body)
    return Atom(z, aromatic, charge, h, True, int(iso) if iso else 0)


def parse(smiles: str) -> Mol:
    mol = Mol()
    s = smiles.strip()
    stack: List[int] = []
    open_rings: Dict[int, Tuple[int, Optional[int], bool]] = {}
    prev: Optional[int] = None
    pending: Optional[Tuple[Optional[int], bool]] = None  # (order, aromatic ':')
    i = 0

    def add_bond(a, b, spec):
        if a == b:
            raise SmilesError("bond from an atom to itself")
        if any(mol.bonds[x].other(a) == b for x in mol.atoms[a].bonds):
            raise SmilesError("two bonds between the same atoms")
        if spec is None:
            arom = mol.atoms[a].written_aromatic and mol.atoms[b].written_aromatic
            bond = Bond(a, b, None if arom else 1, arom)
        else:
            order, arom = spec
            bond = Bond(a, b, None if arom else order, arom)
        mol.bonds.append(bond)
        mol.atoms[a].bonds.append(len(mol.bonds) - 1)
        mol.atoms[b].bonds.append(len(mol.bonds) - 1)

    while i < len(s):
        ch = s[i]
        if ch == "(":
            if prev is None:
                raise SmilesError("branch before any atom")
            stack.append(prev)
            i += 1
        elif ch == ")":
            if not stack:
                raise SmilesError("unbalanced ')'")
            prev = stack.pop()
            i += 1
        elif ch in "-=#:/\\":
            pending = {"-": (1, False), "=": (2, False), "#": (3, False), ":": (None, True), "/": (1, False),
                       "\\": (1, False)}[ch]
            i += 1
        elif ch == ".":
            if pending is not None:
                raise SmilesError("bond symbol before '.'")
            prev = None
            i += 1
        elif ch.isdigit() or ch == "%":
            if ch == "%":
                if not s[i + 1:i + 3].isdigit() or len(s[i + 1:i + 3]) != 2:
                    raise SmilesError("bad %nn ring closure")
                num, i = int(s[i + 1:i + 3]), i + 3
            else:
                num, i = int(ch), i + 1
            if prev is None:
                raise SmilesError("ring closure before any atom")
            if num in open_rings:
                other, spec = open_rings.pop(num)
                add_bond(other, prev, pending if pending is not None else spec)
            else:
                open_rings[num] = (prev, pending)
            pending = None
        else:
            if ch == "[":
                j = s.find("]", i)
                if j < 0:
                    raise SmilesError("unclosed bracket atom")
                atom, i = _read_bracket(s[i + 1:j]), j + 1
            elif s[i:i + 2] in ("Cl", "Br"):
                atom, i = Atom(ORGANIC_SUBSET[s[i:i + 2]]), i + 2
            elif ch in ORGANIC_SUBSET:
                atom, i = Atom(ORGANIC_SUBSET[ch]), i + 1
            elif ch in "bcnops":
                atom, i = Atom(AROMATIC_SYMBOLS[ch], aromatic=True), i + 1
            else:
                raise SmilesError("unexpected character %r" % ch)
            mol.atoms.append(atom)
            idx = len(mol.atoms) - 1
            if prev is not None:
                add_bond(prev, idx, pending)
            elif pending is not None:
                raise SmilesError("bond symbol without a left atom")
            pending = None
            prev = idx
    if stack:
        raise SmilesError("unbalanced '('")
    if open_rings:
        raise SmilesError("unclosed ring bond")
    if pending is not None:
        raise SmilesError("dangling bond symbol")
    if not mol.atoms:
        raise SmilesError("no atoms")
    return mol


def remove_explicit_hydrogens(mol: Mol) -> Mol:
    """rdkit's RemoveHs as MolFromSmiles applies it: a plain [H] with one bond to a heavy atom becomes a hydrogen
    count on that atom (isotopes, charged H and H-H stay atoms)."""
    drop = set()
    for i, a in enumerate(mol.atoms):
        if a.z == 1 and a.isotope == 0 and a.charge == 0 and len(a.bonds) == 1 and a.explicit_h == 0:
            bond = mol.bonds[a.bonds[0]]
            nb = bond.other(i)
            if mol.atoms[nb].z != 1 and bond.order == 1:
                drop.add(i)
    if not drop:
        return mol
    out = Mol()
    remap = {}
    for i, a in enumerate(mol.atoms):
        if i in drop:
            continue
        remap[i] = len(out.atoms)
        na = Atom(a.z, a.written_aromatic, a.charge, a.explicit_h, a.bracket, a.isotope)
        out.atoms.append(na)
    for b in mol.bonds:
        if b.a in drop or b.b in drop:
            heavy = b.b if b.a in drop else b.a
            if mol.atoms[heavy].bracket:
                out.atoms[remap[heavy]].explicit_h += 1
            continue
        nb = Bond(remap[b.a], remap[b.b], b.order, b.written_aromatic)
        out.bonds.append(nb)
        out.atoms[nb.a].bonds.append(len(out.bonds) - 1)
        out.atoms[nb.b].bonds.append(len(out.bonds) - 1)
    return out


# ------------------------------------------------------------------------------------------------ cleanup

def _raw_valence(mol: Mol, i: int) -> int:
    """Explicit valence before kekulisation: waiting aromatic bonds count 1.5."""
    v = sum(1.5 if mol.bonds[b].order is None else mol.bonds[b].order for b in mol.atoms[i].bonds)
    return int(v + mol.atoms[i].explicit_h + 0.1)


def cleanup(mol: Mol) -> None:
    """rdkit's first sanitisation step: hypervalent neutral N, P and halogens written with double bonds become
    charge-separated (CN(=O)=O -> C[N+](=O)[O-], CN=N#N -> CN=[N+]=[N-], C=P(=O)(C)C -> C=[P+]([O-])(C)C,
    OCl(=O)(=O)=O -> O[Cl+3]([O-])([O-])[O-])."""
    for i, a in enumerate(mol.atoms):
        if a.charge != 0:
            continue
        if a.z == 7 and _raw_valence(mol, i) == 5:
            for b in a.bonds:
                bond = mol.bonds[b]
                nb = mol.atoms[bond.other(i)]
                if nb.z == 8 and nb.charge == 0 and bond.order == 2:
                    bond.order, a.charge, nb.charge = 1, 1, -1
                    break
                if nb.z == 7 and nb.charge == 0 and bond.order == 3:
                    bond.order, a.charge, nb.charge = 2, 1, -1
                    break
        elif a.z == 15 and _raw_valence(mol, i) == 5:
            dbl_o, to_c_or_n = None, False
            for b in a.bonds:
                bond = mol.bonds[b]
                j = bond.other(i)
                nb = mol.atoms[j]
                if nb.z == 8 and nb.charge == 0 and bond.order == 2:
                    dbl_o = b
                elif nb.z in (6, 7) and len(nb.bonds) >= 2 and bond.order == 2:
                    to_c_or_n = True
            if dbl_o is not None and to_c_or_n:
                bond = mol.bonds[dbl_o]
                bond.order, a.charge = 1, 1
                mol.atoms[bond.other(i)].charge = -1
        elif a.z in (17, 35, 53) and _raw_valence(mol, i) in (3, 5, 7):
            if all(mol.atoms[mol.bonds[b].other(i)].z == 8 for b in a.bonds):
                for b in a.bonds:
                    bond = mol.bonds[b]
                    if bond.order == 2:
                        bond.order = 1
                        mol.atoms[bond.other(i)].charge = -1
                        a.charge += 1


# ------------------------------------------------------------------------------------------------- rings

def _connected_without(mol: Mol, skip_bond: int, a: int, b: int) -> bool:
    seen, todo = {a}, [a]
    while todo:
        u = todo.pop()
        for bi in mol.atoms[u].bonds:
            if bi == skip_bond:
                continue
            v = mol.bonds[bi].other(u)
            if v == b:
                return True
            if v not in seen:
                seen.add(v)
                todo.append(v)
    return False


def mark_ring_bonds(mol: Mol) -> None:
    for bi, b in enumerate(mol.bonds):
        b.in_ring = _connected_without(mol, bi, b.a, b.b)


def _all_simple_cycles(mol: Mol, max_len: int) -> List[Tuple[int, ...]]:
    """Every simple cycle of the ring-bond subgraph as an atom tuple (smallest atom first, then the smaller of its
    two ring neighbours), by depth-first search."""
    adj = {i: [] for i in range(len(mol.atoms))}
    for b in mol.bonds:
        if b.in_ring:
            adj[b.a].append(b.b)
            adj[b.b].append(b.a)
    found = set()

    def walk(start, path, on_path):
        u = path[-1]
        for v in adj[u]:
            if v == start and len(path) >= 3:
                if path[1] < path[-1]:
                    found.add(tuple(path))
            elif v > start and v not in on_path and len(path) < max_len:
                path.append(v)
                on_path.add(v)
                walk(start, path, on_path)
                path.pop()
                on_path.discard(v)

    for s0 in adj:
        if adj[s0]:
            walk(s0, [s0], {s0})
    return sorted(found, key=lambda c: (len(c), c))


def _bond_index(mol: Mol) -> Dict[Tuple[int, int], int]:
    return {(min(b.a, b.b), max(b.a, b.b)): i for i, b in enumerate(mol.bonds)}


def _cycle_bits(cycle: Sequence[int], bidx: Dict[Tuple[int, int], int]) -> int:
    bits = 0
    for k in range(len(cycle)):
        u, v = cycle[k], cycle[(k + 1) % len(cycle)]
        bits |= 1 << bidx[(min(u, v), max(u, v))]
    return bits


class _GF2Basis(object):

    def __init__(self):
        self.rows: Dict[int, int] = {}  # leading bit -> vector

    def reduce(self, v: int) -> int:
        while v:
            top = v.bit_length() - 1
            if top not in self.rows:
                return v
            v ^= self.rows[top]
        return 0

    def add(self, v: int) -> bool:
        v = self.reduce(v)
        if v:
            self.rows[v.bit_length() - 1] = v
            return True
        return False


def relevant_cycles(mol: Mol, max_len: int = 100) -> List[Tuple[int, ...]]:
    """Cycles that are not a GF(2) sum of strictly shorter cycles."""
    cycles = _all_simple_cycles(mol, max_len)
    bidx = _bond_index(mol)
    n_ring_bonds = sum(1 for b in mol.bonds if b.in_ring)
    ring_atoms = {x for b in mol.bonds if b.in_ring for x in (b.a, b.b)}
    # cyclomatic number of the ring subgraph
    comp = _components(mol, ring_only=True)
    rank = n_ring_bonds - len(ring_atoms) + comp
    basis = _GF2Basis()
    out = []
    k = 0
    while k < len(cycles) and len(basis.rows) < rank:
        size = len(cycles[k])
        same = []
        while k < len(cycles) and len(cycles[k]) == size:
            same.append(cycles[k])
            k += 1
        vecs = [_cycle_bits(c, bidx) for c in same]
        keep = [c for c, v in zip(same, vecs) if basis.reduce(v)]
        for v in vecs:
            basis.add(v)
        out.extend(keep)
    # canonical order (size, sorted bond ids): the fused-ring search below walks ring combinations in this order
    out.sort(key=lambda c: (len(c), sorted(i for i in range(len(mol.bonds)) if _cycle_bits(c, bidx) >> i & 1)))
    return out


def _components(mol: Mol, ring_only: bool) -> int:
    nodes = set()
    adj: Dict[int, List[int]] = {}
    for b in mol.bonds:
        if ring_only and not b.in_ring:
            continue
        adj.setdefault(b.a, []).append(b.b)
        adj.setdefault(b.b, []).append(b.a)
        nodes.update((b.a, b.b))
    seen, n = set(), 0
    for s0 in nodes:
        if s0 in seen:
            continue
        n += 1
        todo = [s0]
        seen.add(s0)
        while todo:
            u = todo.pop()
            for v in adj[u]:
                if v not in seen:
                    seen.add(v)
                    todo.append(v)
    return n


# ---------------------------------------------------------------------------------------------- kekulise

def _charge_adjusted_valences(a: Atom) -> List[int]:
    chg = a.charge
    if a.z in EARLY_ATOMS:
        chg = -chg
    if a.z == 6 and chg > 0:
        chg = -chg
    return [v + chg for v in valence_list(a.z) if v >= 0]


def kekulize(mol: Mol) -> None:
    """Give every aromatic-written bond an order 1 or 2 such that each aromatic atom with a spare valence gets
    exactly one double bond."""
    for a_i, a in enumerate(mol.atoms):
        if a.written_aromatic and not any(mol.bonds[b].in_ring for b in a.bonds):
            raise SmilesError("non-ring atom %d marked aromatic" % a_i)
    for b in mol.bonds:
        if b.order is None and not b.in_ring:
            b.order, b.written_aromatic = 1, False  # biphenyl's link
    arom_bonds = [i for i, b in enumerate(mol.bonds) if b.order is None]
    if not arom_bonds:
        return
    needs = set()
    for i, a in enumerate(mol.atoms):
        if not any(mol.bonds[b].order is None for b in a.bonds):
            continue
        sigma = sum(1 if mol.bonds[b].order is None else mol.bonds[b].order for b in a.bonds) + a.explicit_h
        allowed = _charge_adjusted_valences(a)
        target = next((v for v in allowed if v >= sigma), None)
        if target is not None and target - sigma >= 1:
            needs.add(i)
    options = {i: [b for b in mol.atoms[i].bonds if mol.bonds[b].order is None and mol.bonds[b].other(i) in needs]
               for i in needs}
    chosen: List[int] = []

    def solve(free: set) -> bool:
        if not free:
            return True
        best, best_opts = None, None
        for i in sorted(free):  # first atom with the fewest choices, so the Kekule form is reproducible
            opts = [b for b in options[i] if mol.bonds[b].other(i) in free]
            if best is None or len(opts) < len(best_opts):
                best, best_opts = i, opts
                if not opts:
                    return False
        for b in best_opts:
            j = mol.bonds[b].other(best)
            chosen.append(b)
            if solve(free - {best, j}):
                return True
            chosen.pop()
        return False

    if not solve(set(needs)):
        raise SmilesError("cannot kekulize")
    for b in arom_bonds:
        mol.bonds[b].order = 1
    for b in chosen:
        mol.bonds[b].order = 2


# --------------------------------------------------------------------------------------- valence, radicals

def assign_valence(mol: Mol) -> None:
    for i, a in enumerate(mol.atoms):
        ev = mol.explicit_valence(i)
        valens = valence_list(a.z)
        effective = ev - a.charge if n_outer_elecs(a.z) >= 4 else ev + a.charge
        if valens[-1] > 0 and effective > valens[-1]:
            raise SmilesError("valence %d of atom %d (%s) too high" % (ev, i, a.symbol))
        if a.bracket:
            a.implicit_h = 0
            continue
        target = next((v for v in _charge_adjusted_valences(a) if v >= ev), None)
        if target is None:
            if valens[-1] == -1:
                a.implicit_h = 0
                continue
            raise SmilesError("valence of atom %d too high" % i)
        a.implicit_h = target - ev


def assign_radicals(mol: Mol) -> None:
    for i, a in enumerate(mol.atoms):
        a.radicals = 0
        if not a.bracket:
            continue
        valens = valence_list(a.z)
        if valens == (-1,):
            continue
        n_outer = n_outer_elecs(a.z)
        total = mol.explicit_valence(i)
        base = 2 if a.z <= 2 else 8
        late = base - n_outer - total + a.charge
        if late < 0:
            late = 0
            if len(valens) > 1:
                for v in valens:
                    if v - total + a.charge >= 0:
                        late = v - total + a.charge
                        break
        early = n_outer - total - a.charge
        if early >= 0:
            late = min(late, early)
        a.radicals = late


# ------------------------------------------------------------------------------------------- aromaticity

VACANT, ONE, TWO, NONE = "vacant", "one", "two", "none"


def count_atom_elec(mol: Mol, i: int) -> int:
    a = mol.atoms[i]
    dv = default_valence(a.z)
    if dv <= 1:
        return 0
    degree = mol.degree(i) + mol.total_h(i)
    if degree > 3:
        return -1
    nlp = max(n_outer_elecs(a.z) - dv - a.charge, 0)
    res = (dv - degree) + nlp - a.radicals
    if res > 1:
        if mol.explicit_valence(i) - mol.degree(i) > 1:
            res = 1
    return res


def more_electronegative(z1: int, z2: int) -> bool:
    n1, n2 = n_outer_elecs(z1), n_outer_elecs(z2)
    return n1 > n2 or (n1 == n2 and z1 < z2)


def _donor_type(mol: Mol, i: int) -> str:
    a = mol.atoms[i]
    nelec = count_atom_elec(mol, i)
    exo = [mol.bonds[b].other(i) for b in a.bonds if not mol.bonds[b].in_ring and mol.bonds[b].order >= 2]
    cyc = any(mol.bonds[b].in_ring and mol.bonds[b].order >= 2 for b in a.bonds)
    multiple = any(mol.bonds[b].order >= 2 for b in a.bonds)
    if nelec < 0:
        return NONE
    if nelec == 0:
        if exo:
            return VACANT
        if cyc:
            return ONE
        return NONE
    if nelec == 1:
        if exo:
            return VACANT if more_electronegative(mol.atoms[exo[0]].z, a.z) else ONE
        if multiple:
            return ONE
        if a.charge == 1:
            return VACANT
        return NONE
    if exo and more_electronegative(mol.atoms[exo[0]].z, a.z):
        nelec -= 1
    return ONE if nelec % 2 == 1 else TWO


def _arom_candidate(mol: Mol, i: int, donor: str) -> bool:
    a = mol.atoms[i]
    if a.z > 18 and a.z not in (34, 52):
        return False
    if donor == NONE:
        return False
    dv = default_valence(a.z)
    if dv > 0 and mol.total_valence(i) > default_valence(a.z - a.charge):
        return False
    n_mult = sum(1 for b in a.bonds if mol.bonds[b].order >= 2)
    if mol.explicit_valence(i) - mol.degree(i) > 1 and n_mult > 1:
        return False
    return True


def _huckel(donors: Sequence[str]) -> bool:
    n = sum({VACANT: 0, ONE: 1, TWO: 2}[d] for d in donors)
    if n >= 6:
        return (n - 2) % 4 == 0
    return n == 2


def perceive_aromaticity(mol: Mol) -> None:
    for a in mol.atoms:
        a.aromatic = False
    for b in mol.bonds:
        b.aromatic = False
    donors = [_donor_type(mol, i) for i in range(len(mol.atoms))]
    cand = [_arom_candidate(mol, i, donors[i]) for i in range(len(mol.atoms))]
    bidx = _bond_index(mol)
    rings = [r for r in mol.rings if all(cand[x] for x in r)]
    ring_bonds = []
    for r in rings:
        ring_bonds.append({bidx[(min(r[k], r[(k + 1) % len(r)]), max(r[k], r[(k + 1) % len(r)]))]
                           for k in range(len(r))})
    n = len(rings)
    fused = [[j for j in range(n) if j != i and ring_bonds[i] & ring_bonds[j]] for i in range(n)]
    done = set()

    def try_combo(combo):
        count: Dict[int, int] = {}
        for r in combo:
            for b in ring_bonds[r]:
                count[b] = count.get(b, 0) + 1
        outer = [b for b, c in count.items() if c == 1]
        atoms = sorted({x for b in outer for x in (mol.bonds[b].a, mol.bonds[b].b)})
        if _huckel([donors[x] for x in atoms]):
            for b in outer:
                mol.bonds[b].aromatic = True
            for x in atoms:
                mol.atoms[x].aromatic = True
            done.update(combo)

    for size in (1, 2, 3):
        if len(done) == n:
            break
        for combo in itertools.combinations(range(n), size):
            if size > 1:
                # connected through shared bonds
                seen, todo = {combo[0]}, [combo[0]]
                while todo:
                    u = todo.pop()
                    for v in fused[u]:
                        if v in combo and v not in seen:
                            seen.add(v)
                            todo.append(v)
                if len(seen) != size:
                    continue
                if all(r in done for r in combo):
                    continue
            try_combo(combo)


# --------------------------------------------------------------------------- conjugation, hybridisation

def _bond_contrib(b: Bond) -> float:
    return 1.5 if b.aromatic else float(b.order)


def _conj_candidate(mol: Mol, i: int) -> bool:
    a = mol.atoms[i]
    n_outer = n_outer_elecs(a.z)
    return (a.z <= 10 or (n_outer != 5 and n_outer != 6)) and count_atom_elec(mol, i) > 0


def mark_conjugation(mol: Mol) -> None:
    for b in mol.bonds:
        b.conjugated = b.aromatic
    for i, a in enumerate(mol.atoms):
        sbo = mol.degree(i) + mol.total_h(i)
        if sbo < 2 or sbo > 3:
            continue
        for b1 in a.bonds:
            if _bond_contrib(mol.bonds[b1]) < 1.5:
                continue
            for b2 in a.bonds:
                if b1 == b2:
                    continue
                j = mol.bonds[b2].other(i)
                if mol.degree(j) + mol.total_h(j) > 3:
                    continue
                if _conj_candidate(mol, j):
                    mol.bonds[b1].conjugated = True
                    mol.bonds[b2].conjugated = True


def set_hybridization(mol: Mol) -> None:
    for i, a in enumerate(mol.atoms):
        deg = mol.degree(i) + mol.total_h(i)
        if a.z <= 1:
            norbs = deg
        else:
            n_outer = n_outer_elecs(a.z)
            tv = mol.total_valence(i)
            free = n_outer - (tv + a.charge)
            if tv + n_outer - a.charge < 8:
                norbs = deg + int((free - a.radicals) / 2) + a.radicals  # C-style truncation
            else:
                norbs = deg + int(free / 2)
        if norbs <= 1:
            hyb = "S"
        elif norbs == 2:
            hyb = "SP"
        elif norbs == 3:
            hyb = "SP2"
        elif norbs == 4:
            conj = any(mol.bonds[b].conjugated for b in a.bonds)
            hyb = "SP3" if (deg > 3 or not conj) else "SP2"
        elif norbs == 5:
            hyb = "SP3D"
        elif norbs == 6:
            hyb = "SP3D2"
        else:
            hyb = "UNSPECIFIED"
        a.hybridization = hyb


def mol_from_smiles(smiles: str) -> Mol:
    mol = remove_explicit_hydrogens(parse(smiles))
    if any(len(a.bonds) > 10 for a in mol.atoms):
        raise SmilesError("atom degree above 10")  # the one-hot degree block has no such column (:369)
    cleanup(mol)
    mark_ring_bonds(mol)
    kekulize(mol)
    assign_valence(mol)
    assign_radicals(mol)
    mol.rings = relevant_cycles(mol)
    perceive_aromaticity(mol)
    mark_conjugation(mol)
    set_hybridization(mol)
    return mol


# --------------------------------------------------------------------------------------------- features

def one_of_k(x, allowable, unk: bool) -> List[float]:
    if x not in allowable:
        if not unk:
            raise ValueError("input {0} not in allowable set{1}:".format(x, allowable))
        x = allowable[-1]
    return [1.0 if x == s else 0.0 for s in allowable]


def atom_features(mol: Mol, i: int) -> np.ndarray:
    """graph_features.py:282-391 with the defaults (no chirality, total-H block present)."""
    a = mol.atoms[i]
    out = one_of_k(a.symbol, ATOM_SYMBOL_LIST, True)
    out += one_of_k(mol.degree(i), list(range(11)), False)
    out += one_of_k(a.implicit_h, list(range(7)), True)
    out += [float(a.charge), float(a.radicals)]
    out += one_of_k(a.hybridization, HYBRIDIZATIONS, True)
    out += [1.0 if a.aromatic else 0.0]
    out += one_of_k(mol.total_h(i), list(range(5)), True)
    return np.asarray(out, dtype=np.float64)


def bond_features(b: Bond) -> np.ndarray:
    """graph_features.py:394-459 (no chirality)."""
    return np.asarray([not b.aromatic and b.order == 1, not b.aromatic and b.order == 2,
                       not b.aromatic and b.order == 3, b.aromatic, b.conjugated, b.in_ring], dtype=np.float64)


def conv_mol_arrays(smiles: str) -> Tuple[np.ndarray, List[List[int]]]:
    """(atom feature matrix (n,75), adjacency lists in bond order) -- the two arguments the reference hands to
    ``ConvMol`` (graph_features.py:881-911)."""
    mol = mol_from_smiles(smiles)
    nodes = np.vstack([atom_features(mol, i) for i in range(len(mol.atoms))])
    adj: List[List[int]] = [[] for _ in mol.atoms]
    for b in mol.bonds:
        adj[b.a].append(b.b)
        adj[b.b].append(b.a)
    return nodes, adj


def find_distance(a1: int, n: int, adj: List[List[int]], max_distance: int = 7) -> np.ndarray:
    """graph_features.py:655-695."""
    distance = np.zeros((n, max_distance))
    frontier, seen = set(adj[a1]), {a1}
    for radial in range(max_distance):
        distance[list(frontier), radial] = 1
        seen |= frontier
        nxt = set()
        for x in frontier:
            nxt.update(adj[x])
        frontier = nxt - seen
    return distance


def weave_mol_arrays(smiles: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(nodes (n,75), pair features (n*n,14), pair_edges (2,n*n)) for max_pair_distance=None
    (graph_features.py:532-652, :1037-1078).  All pairs, row-major (a1 major)."""
    mol = mol_from_smiles(smiles)
    n = len(mol.atoms)
    nodes = np.vstack([atom_features(mol, i) for i in range(n)])
    adj: List[List[int]] = [[] for _ in range(n)]
    feats = np.zeros((n * n, 14))
    for b in mol.bonds:
        adj[b.a].append(b.b)
        adj[b.b].append(b.a)
        bf = bond_features(b)
        feats[b.a * n + b.b, :6] = bf
        feats[b.b * n + b.a, :6] = bf
    for ring in mol.rings:
        for x in ring:
            for y in ring:
                if x != y:
                    feats[x * n + y, 6] = 1
    for a1 in range(n):
        feats[a1 * n:(a1 + 1) * n, 7:] = find_distance(a1, n, adj)
    edges = np.stack([np.repeat(np.arange(n), n), np.tile(np.arange(n), n)])
    return nodes, feats, edges

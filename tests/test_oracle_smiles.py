"""The SMILES featurization oracle against what can be pinned without rdkit:

* the reference's known answers for 'C', 'CCC', 'C[N+](C)(C)C' (feat/tests/test_graph_features.py:14-104,
  feat/tests/test_weave.py:47-124);
* the reference's WeaveLayer assets: A.npy / P.npy are the layer outputs on rdkit-featurized ['CCC', 'C']
  with the stored weights (models/tests/test_weavelayer_pytorch.py:14-79) -- they pin the full 75-column
  vectors of CH3 / CH2 / CH4 carbons and all 14 pair columns of those molecules;
* hand-derived chemistry for the rest.
"""
import os

import numpy as np
import pytest
import torch

from oracle import smiles_oracle as so
from oracle import weave_oracle as wo
from oracle.mol_graphs_oracle import conv_mol

HERE = os.path.dirname(os.path.abspath(__file__))


def _props(smiles):
    mol = so.mol_from_smiles(smiles)
    return mol, [(a.symbol, mol.degree(i), a.implicit_h, a.explicit_h, a.charge, a.radicals, a.hybridization,
                  a.aromatic) for i, a in enumerate(mol.atoms)]


def test_reference_known_answers_for_adjacency():
    nodes, adj = so.conv_mol_arrays("C[N+](C)(C)C")
    mol = conv_mol(nodes, adj)
    assert mol["n_atoms"] == 5
    lists = mol["deg_adj_lists"]
    assert np.array_equal(lists[1], np.array([[4], [4], [4], [4]], dtype=np.int32))
    assert np.array_equal(lists[4], np.array([[0, 1, 2, 3]], dtype=np.int32))
    for d in (2, 3, 5, 6):
        assert lists[d].shape == (0, d)

    nodes, adj = so.conv_mol_arrays("C")
    mol = conv_mol(nodes, adj)
    assert mol["n_atoms"] == 1
    assert mol["deg_adj_lists"][0].shape == (1, 0)

    nodes, adj = so.conv_mol_arrays("CCC")
    lists = conv_mol(nodes, adj)["deg_adj_lists"]
    assert np.array_equal(lists[1], np.array([[2], [2]], dtype=np.int32))
    assert np.array_equal(lists[2], np.array([[0, 1]], dtype=np.int32))


def test_reference_known_answers_for_weave_shapes():
    for smi, n in (("C", 1), ("CCC", 3), ("C[N+](C)(C)C", 5), ("CCCCC", 5)):
        nodes, pairs, edges = so.weave_mol_arrays(smi)
        assert nodes.shape == (n, 75)
        assert pairs.shape == (n * n, 14)
        assert edges.shape == (2, n * n)


def test_weave_layer_assets_pin_the_carbon_vectors_and_pair_features():
    d = os.path.join(HERE, "golden", "weave_assets")
    w = {k: torch.from_numpy(np.load(os.path.join(d, k + ".npy"))) for k in ("W_AA", "W_PA", "W_A", "W_AP", "W_PP",
                                                                             "W_P")}
    p = dict(w)
    for k in list(w):
        p["b" + k[1:]] = torch.zeros(w[k].shape[1])
    fresh_bn = lambda n: {"running_mean": torch.zeros(n), "running_var": torch.ones(n), "weight": torch.ones(n),
                          "bias": torch.zeros(n)}
    bns = {k: fresh_bn(50) for k in ("AA", "PA", "A", "AP", "PP", "P")}
    atom_feat, pair_feat, a2p, split = [], [], [], []
    start = 0
    for smi in ("CCC", "C"):
        nodes, pairs, edges = so.weave_mol_arrays(smi)
        atom_feat.append(nodes)
        pair_feat.append(pairs)
        a2p.append(edges.T + start)
        split.extend(edges[0] + start)
        start += len(nodes)
    A, P = wo.weave_layer(np.concatenate(atom_feat).astype(np.float32), np.concatenate(pair_feat).astype(np.float32),
                          np.asarray(split), np.concatenate(a2p), p, bns)
    np.testing.assert_allclose(A.numpy(), np.load(os.path.join(d, "A.npy")), atol=1e-4)
    np.testing.assert_allclose(P.numpy(), np.load(os.path.join(d, "P.npy")), atol=1e-4)
    # and they equal the hand-derived vectors the other fixtures use
    np.testing.assert_array_equal(atom_feat[0][0], wo.carbon_atom_features(1, 3))
    np.testing.assert_array_equal(atom_feat[0][1], wo.carbon_atom_features(2, 2))
    np.testing.assert_array_equal(atom_feat[1][0], wo.carbon_atom_features(0, 4))


def test_atom_properties_of_common_groups():
    _, p = _props("CC(=O)O")
    assert p == [("C", 1, 3, 0, 0, 0, "SP3", False), ("C", 3, 0, 0, 0, 0, "SP2", False),
                 ("O", 1, 0, 0, 0, 0, "SP2", False), ("O", 1, 1, 0, 0, 0, "SP2", False)]
    _, p = _props("CC#N")
    assert [x[6] for x in p] == ["SP3", "SP", "SP"]
    _, p = _props("CCO")
    assert p[2] == ("O", 1, 1, 0, 0, 0, "SP3", False)
    _, p = _props("[Na+].[Cl-]")
    assert p == [("Na", 0, 0, 0, 1, 0, "S", False), ("Cl", 0, 0, 0, -1, 0, "SP3", False)]
    _, p = _props("C[N+](C)(C)C")
    assert p[1] == ("N", 4, 0, 0, 1, 0, "SP3", False)
    _, p = _props("[O-][N+](=O)c1ccccc1")
    assert p[0][4] == -1 and p[1][4] == 1 and p[1][6] == "SP2"
    _, p = _props("OS(=O)(=O)O")
    assert p[1] == ("S", 4, 0, 0, 0, 0, "SP3", False)
    _, p = _props("[CH3]")
    assert p == [("C", 0, 0, 3, 0, 1, "SP3", False)]
    _, p = _props("Clc1ccccc1")
    assert p[0][6] == "SP3"  # halogens do not conjugate
    _, p = _props("Oc1ccccc1")
    assert p[0][6] == "SP2"  # phenol oxygen does
    _, p = _props("CC(=O)Nc1ccccc1")
    assert p[3] == ("N", 2, 1, 0, 0, 0, "SP2", False)
    _, p = _props("[H]C([H])([H])O")
    assert p == [("C", 1, 3, 0, 0, 0, "SP3", False), ("O", 1, 1, 0, 0, 0, "SP3", False)]


def test_aromatic_rings_written_either_way():
    for smi in ("c1ccccc1", "C1=CC=CC=C1", "C1:C:C:C:C:C1"):
        mol, p = _props(smi)
        assert all(x == ("C", 2, 1, 0, 0, 0, "SP2", True) for x in p), smi
        assert all(b.aromatic and b.conjugated and b.in_ring for b in mol.bonds)
    mol, p = _props("c1ccncc1")
    assert p[3] == ("N", 2, 0, 0, 0, 0, "SP2", True)
    mol, p = _props("c1cc[nH]c1")
    assert p[3] == ("N", 2, 0, 1, 0, 0, "SP2", True)
    mol, p = _props("C1=CNC=C1")  # pyrrole, Kekule
    assert p[2] == ("N", 2, 1, 0, 0, 0, "SP2", True)
    mol, p = _props("Cn1cccc1")
    assert p[1] == ("N", 3, 0, 0, 0, 0, "SP2", True)
    mol, p = _props("c1ccsc1")
    assert p[3] == ("S", 2, 0, 0, 0, 0, "SP2", True)
    mol, p = _props("c1ccoc1")
    assert p[3] == ("O", 2, 0, 0, 0, 0, "SP2", True)
    mol, p = _props("O=c1cccc[nH]1")  # 2-pyridone: the carbonyl carbon gives no electron, N gives two
    assert all(x[7] for x in p[1:]) and not p[0][7]
    mol, p = _props("C1=CCC=C1")  # cyclopentadiene: sp3 carbon in the ring
    assert not any(x[7] for x in p)
    mol, p = _props("C1=CC=CC=CC=C1")  # cyclooctatetraene: 8 electrons
    assert not any(x[7] for x in p)
    mol, p = _props("c1ccc2ccccc2c1")
    assert all(x[7] for x in p) and all(b.aromatic for b in mol.bonds)
    mol, p = _props("c1ccccc1c1ccccc1")  # biphenyl: the link is a single bond
    link = [b for b in mol.bonds if not b.in_ring]
    assert len(link) == 1 and link[0].order == 1 and not link[0].aromatic and link[0].conjugated
    mol, p = _props("c1ccc2cccc2cc1")  # azulene: aromatic along the outer 10 bonds only
    assert all(x[7] for x in p)
    assert sum(1 for b in mol.bonds if not b.aromatic) == 1
    mol, p = _props("[cH-]1cccc1")
    assert all(x[7] for x in p) and p[0][4] == -1
    mol, p = _props("c1cc[nH+]cc1")
    assert p[3] == ("N", 2, 0, 1, 1, 0, "SP2", True)


def test_rings_are_the_relevant_cycles():
    assert sorted(len(r) for r in so.mol_from_smiles("C1CC2CCC1C2").rings) == [5, 5]  # norbornane: the 6-ring is the sum of the two 5-rings
    assert sorted(len(r) for r in so.mol_from_smiles("C12C3C4C1C5C2C3C45").rings) == [4] * 6  # cubane
    assert sorted(len(r) for r in so.mol_from_smiles("C1CC2CCC1CC2").rings) == [6, 6, 6]  # bicyclo[2.2.2]octane
    assert sorted(len(r) for r in so.mol_from_smiles("C1CC2CC1C2").rings) == [4, 5, 5]
    assert sorted(len(r) for r in so.mol_from_smiles("C1C2CC3CC1CC(C2)C3").rings) == [6, 6, 6, 6]  # adamantane
    assert so.mol_from_smiles("CCCC").rings == []
    assert sorted(len(r) for r in so.mol_from_smiles("C1CC1C1CC1").rings) == [3, 3]
    assert sorted(len(r) for r in so.mol_from_smiles("C1CCC2(CC1)CCCC2").rings) == [5, 6]  # spiro


def test_molecules_the_reference_pipeline_would_drop():
    for smi in ("c1cccc1", "CN(=O)(=O)=O", "C(C)(C)(C)(C)C", "C1CC", "C(", "c1ccccc1)", "", "[Xx]", "cC", "C%1CC%1"):
        with pytest.raises(so.SmilesError):
            so.mol_from_smiles(smi)


def test_hypervalent_groups_are_charge_separated_first():
    """rdkit's cleanUp step; delaney-processed.csv writes its 58 nitro compounds as N(=O)=O."""
    _, p = _props("CN(=O)=O")
    assert p[1] == ("N", 3, 0, 0, 1, 0, "SP2", False)
    assert sorted(x[4] for x in p) == [-1, 0, 0, 1]
    _, p = _props("CN=N#N")
    assert [x[4] for x in p] == [0, 0, 1, -1]
    _, p = _props("OCl(=O)(=O)=O")
    assert p[1][4] == 3 and [x[4] for x in p].count(-1) == 3
    mol, p = _props("Nc1cc(nc(N)n1=O)N2CCCCC2")  # minoxidil as the datasets write it
    assert p[7][4] == 1 and p[8][4] == -1 and p[7][7]


def test_bond_and_pair_columns_of_a_ring_molecule():
    nodes, pairs, edges = so.weave_mol_arrays("OC1CC1")  # cyclopropanol
    n = 4
    P = pairs.reshape(n, n, 14)
    assert np.array_equal(P[0, 1, :6], [1, 0, 0, 0, 0, 0])  # O-C single, not in ring
    assert np.array_equal(P[1, 2, :6], [1, 0, 0, 0, 0, 1])
    assert P[1, 2, 6] == 1 and P[2, 3, 6] == 1 and P[1, 3, 6] == 1 and P[0, 1, 6] == 0 and P[1, 1, 6] == 0
    assert np.array_equal(P[0, 2, 7:], [0, 1, 0, 0, 0, 0, 0])  # graph distance 2
    assert np.array_equal(P[0, 0, 7:], np.zeros(7))
    assert np.array_equal(edges[:, 5], [1, 1])
    # a chain longer than the 7 distance bins: distance 8 has no bin
    nodes, pairs, edges = so.weave_mol_arrays("CCCCCCCCC")
    P = pairs.reshape(9, 9, 14)
    assert np.array_equal(P[0, 7, 7:], [0, 0, 0, 0, 0, 0, 1])
    assert np.array_equal(P[0, 8, 7:], np.zeros(7))


def test_atom_feature_vector_layout():
    nodes, _ = so.conv_mol_arrays("c1ccncc1")
    n_vec = nodes[3]
    assert n_vec.shape == (75,)
    assert n_vec[1] == 1 and n_vec[:44].sum() == 1  # N is symbol 1
    assert n_vec[44 + 2] == 1  # degree 2
    assert n_vec[55 + 0] == 1  # implicit valence 0
    assert n_vec[62] == 0 and n_vec[63] == 0
    assert np.array_equal(n_vec[64:69], [0, 1, 0, 0, 0])  # SP2
    assert n_vec[69] == 1
    assert np.array_equal(n_vec[70:], [1, 0, 0, 0, 0])
    nodes, _ = so.conv_mol_arrays("[U]")  # not in the symbol list -> 'Unknown'; hybridisation 'S' -> last slot
    assert nodes[0][43] == 1

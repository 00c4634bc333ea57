"""Eight-byte atom codes: the 75-column atom feature row of ``atom_features``
(deepchem/feat/graph_features.py:282-391) is five one-hot blocks, two small integers and a flag, i.e. 300 bytes
that say 8 bytes' worth.  A molecule set whose rows ARE such rows (anything the reference's or this package's
ConvMolFeaturizer produces) is kept, collated and sent over PCIe as codes and expanded to float32 rows on the
GPU (``gcmi_expand_atom_codes``): 38x less host memory traffic in collation and in the H2D copy.

code byte | meaning                                   | feature columns
   0      | symbol column 0..43                       | 0..43   one-hot
   1      | degree 0..10                              | 44..54  one-hot
   2      | implicit-valence column 0..6              | 55..61  one-hot
   3      | formal charge (int8)                      | 62      value
   4      | radical electrons                         | 63      value
   5      | hybridisation column 0..4                 | 64..68  one-hot
   6      | aromatic 0/1                              | 69      value
   7      | total-H column 0..4                       | 70..74  one-hot

``codes_from_features`` only returns codes when expanding them gives the input back bit for bit; rows with other
content (``master_atom`` means, user features, chirality columns) stay float.
"""
from typing import Optional

import numpy as np

N_FEAT = 75
_BLOCKS = ((0, 0, 44), (1, 44, 11), (2, 55, 7), (5, 64, 5), (7, 70, 5))  # (code byte, first column, width)

# column of each atomic number in the reference's symbol list (index 43 = 'Unknown')
_SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
            "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu "
            "Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr").split()
_LIST = ['C', 'N', 'O', 'S', 'F', 'Si', 'P', 'Cl', 'Br', 'Mg', 'Na', 'Ca', 'Fe', 'As', 'Al', 'I', 'B', 'V', 'K', 'Tl',
         'Yb', 'Sb', 'Sn', 'Ag', 'Pd', 'Co', 'Se', 'Ti', 'Zn', 'H', 'Li', 'Ge', 'Cu', 'Au', 'Ni', 'Cd', 'In', 'Mn', 'Zr',
         'Cr', 'Pt', 'Hg', 'Pb']
SYMBOL_COLUMN = np.full(len(_SYMBOLS) + 1, 43, np.uint8)
for _c, _s in enumerate(_LIST):
    SYMBOL_COLUMN[_SYMBOLS.index(_s) + 1] = _c
# hybridisation id of gcmi_smiles_featurize's property rows (0 unspecified, 1 S, 2 SP, 3 SP2, 4 SP3, 5 SP3D,
# 6 SP3D2) -> column of [SP, SP2, SP3, SP3D, SP3D2] with everything else in the last one
_HYB_COLUMN = np.array([4, 4, 0, 1, 2, 3, 4], np.uint8)


def features_from_codes(codes: np.ndarray) -> np.ndarray:
    """(A, 8) uint8 -> (A, 75) float32 (host expansion; the device does the same in gcmi_expand_atom_codes)."""
    codes = np.asarray(codes, np.uint8).reshape(-1, 8)
    a = codes.shape[0]
    out = np.zeros((a, N_FEAT), np.float32)
    rows = np.arange(a)
    for byte, first, width in _BLOCKS:
        out[rows, first + np.minimum(codes[:, byte], width - 1)] = 1.0
    out[:, 62] = codes[:, 3].view(np.int8)
    out[:, 63] = codes[:, 4]
    out[:, 69] = codes[:, 6]
    return out


def codes_from_features(features: np.ndarray) -> Optional[np.ndarray]:
    """(A, 75) feature rows -> (A, 8) uint8 codes, or None when the rows are not exactly code-shaped."""
    f = np.asarray(features)
    if f.ndim != 2 or f.shape[1] != N_FEAT:
        return None
    codes = np.zeros((f.shape[0], 8), np.uint8)
    for byte, first, width in _BLOCKS:
        codes[:, byte] = np.argmax(f[:, first:first + width], axis=1)
    charge, rad, arom = f[:, 62], f[:, 63], f[:, 69]
    if f.shape[0] and (np.abs(charge).max() > 127 or rad.min() < 0 or rad.max() > 255):
        return None
    codes[:, 3] = charge.astype(np.int8).view(np.uint8)
    codes[:, 4] = rad.astype(np.uint8)
    codes[:, 6] = arom.astype(np.uint8)
    if not np.array_equal(features_from_codes(codes), f.astype(np.float32)) or not np.array_equal(f.astype(np.float32), f):
        return None
    return codes


def codes_from_props(props: np.ndarray) -> np.ndarray:
    """(A, 8) int32 property rows of ``gcmi_smiles_featurize`` (atomic number, degree, implicit H, explicit H,
    charge, radicals, hybridisation id, aromatic) -> (A, 8) uint8 codes."""
    p = np.asarray(props, np.int64).reshape(-1, 8)
    codes = np.zeros((p.shape[0], 8), np.uint8)
    codes[:, 0] = SYMBOL_COLUMN[np.clip(p[:, 0], 0, len(SYMBOL_COLUMN) - 1)]
    codes[:, 1] = np.minimum(p[:, 1], 10)
    codes[:, 2] = np.where((p[:, 2] >= 0) & (p[:, 2] < 7), p[:, 2], 6)
    codes[:, 3] = p[:, 4].astype(np.int8).view(np.uint8)
    codes[:, 4] = np.clip(p[:, 5], 0, 255)
    codes[:, 5] = _HYB_COLUMN[np.clip(p[:, 6], 0, 6)]
    codes[:, 6] = p[:, 7] != 0
    toth = p[:, 2] + p[:, 3]
    codes[:, 7] = np.where((toth >= 0) & (toth < 5), toth, 4)
    return codes

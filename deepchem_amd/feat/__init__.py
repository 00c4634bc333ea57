from deepchem_amd.feat import mol_graphs
from deepchem_amd.feat.mol_graphs import ConvMol, MultiConvMol

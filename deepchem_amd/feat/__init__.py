from deepchem_amd.feat import mol_graphs
from deepchem_amd.feat.mol_graphs import ConvMol, MultiConvMol
from deepchem_amd.feat import graph_features
from deepchem_amd.feat.graph_features import ConvMolFeaturizer, WeaveFeaturizer

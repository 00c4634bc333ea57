from deepchem_amd.models.torch_models.torch_model import TorchModel
from deepchem_amd.models.torch_models.graphconvmodel import GraphConvModel, _GraphConvTorchModel
from deepchem_amd.models.torch_models import layers
from deepchem_amd.models.torch_models.weavemodel_pytorch import Weave, WeaveModel, WeaveMol
from deepchem_amd.models.torch_models.mpnn import MPNNModel

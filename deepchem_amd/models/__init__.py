from deepchem_amd.models.models import Model
from deepchem_amd.models import losses, optimizers, torch_models
from deepchem_amd.models.torch_models import GraphConvModel, TorchModel, WeaveModel

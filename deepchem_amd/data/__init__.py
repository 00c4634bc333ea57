from deepchem_amd.data.datasets import Dataset, NumpyDataset, pad_batch
from deepchem_amd.data.packed_dataset import PackedDataset, packed_from_convmols

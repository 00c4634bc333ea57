from deepchem_amd.data.datasets import Dataset, DiskDataset, NumpyDataset, pad_batch  # noqa: F401
from deepchem_amd.data.data_loader import CSVLoader, DataLoader  # noqa: F401
from deepchem_amd.data.packed_dataset import PackedDataset, packed_from_convmols  # noqa: F401

from deepchem_amd.data.datasets import Dataset, NumpyDataset, pad_batch

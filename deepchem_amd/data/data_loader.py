"""CSV -> sharded DiskDataset (deepchem/data/data_loader.py): label / weight extraction
(``_convert_df_to_numpy`` :35-69), the shard loop of ``DataLoader.create_dataset`` (:177-240) and
``CSVLoader`` (:281-437) over ``load_csv_files`` (utils/data_utils.py:320-350).

The featurizer is any callable ``featurizer(sequence of inputs) -> sequence of features`` (an empty
array marks a failed input, which is dropped together with its labels, as in the reference).  The
reference's molecular featurizers need rdkit, which is not part of this package: plug in
``deepchem.feat.ConvMolFeaturizer()`` where DeepChem is installed."""
from typing import Any, Iterator, List, Optional, Tuple

import numpy as np

from deepchem_amd.data.datasets import Dataset, DiskDataset


def convert_df_to_numpy(df, tasks: List[str]) -> Tuple[np.ndarray, np.ndarray]:
    """``(y, w)`` from the task columns of a dataframe: a missing label (empty string after
    ``load_csv_files`` replaced NaN) becomes y = 0 with w = 0; everything else has w = 1."""
    n_samples = df.shape[0]
    n_tasks = len(tasks)
    y = np.hstack([np.reshape(np.array(df[task].values), (n_samples, 1)) for task in tasks])
    w = np.ones((n_samples, n_tasks))
    if y.dtype.kind in ["O", "U"]:
        missing = (y == "")
        y[missing] = 0
        w[missing] = 0
    return y.astype(float), w.astype(float)


_convert_df_to_numpy = convert_df_to_numpy


def load_csv_files(input_files: List[str], shard_size: Optional[int] = None) -> Iterator[Any]:
    """Dataframes of at most ``shard_size`` rows; NaN cells become empty strings (only when
    sharding -- the reference leaves the unsharded frame untouched)."""
    import pandas as pd
    for input_file in input_files:
        if shard_size is None:
            yield pd.read_csv(input_file)
        else:
            for df in pd.read_csv(input_file, chunksize=shard_size):
                yield df.replace(np.nan, str(""), regex=True)


class DataLoader(object):
    """Template: ``_get_shards`` yields dataframes, ``_featurize_shard`` returns
    ``(features, valid_mask)`` (data_loader.py:72-278)."""

    def __init__(self, tasks: List[str], featurizer, id_field: Optional[str] = None, log_every_n: int = 1000):
        if self.__class__ is DataLoader:
            raise ValueError("DataLoader should never be instantiated directly. Use a subclass instead.")
        if not isinstance(tasks, list):
            raise ValueError("tasks must be a list.")
        self.tasks = tasks
        self.id_field = id_field
        self.featurizer = featurizer
        self.log_every_n = log_every_n

    def featurize(self, inputs, data_dir: Optional[str] = None, shard_size: Optional[int] = 8192) -> Dataset:
        return self.create_dataset(inputs, data_dir, shard_size)

    def create_dataset(self, inputs, data_dir: Optional[str] = None, shard_size: Optional[int] = 8192) -> Dataset:
        if not isinstance(inputs, list):
            inputs = [inputs]

        def shard_generator():
            for shard in self._get_shards(inputs, shard_size):
                X, valid_inds = self._featurize_shard(shard)
                ids = shard[self.id_field].values[valid_inds]
                if len(self.tasks) > 0:
                    y, w = convert_df_to_numpy(shard, self.tasks)
                    y, w = y[valid_inds], w[valid_inds]
                    assert len(X) == len(ids) == len(y) == len(w)
                else:
                    y, w = None, None
                    assert len(X) == len(ids)
                yield X, y, w, ids

        return DiskDataset.create_dataset(shard_generator(), data_dir, self.tasks)

    def _get_shards(self, inputs: List, shard_size: Optional[int]) -> Iterator:
        raise NotImplementedError

    def _featurize_shard(self, shard: Any):
        raise NotImplementedError


class CSVLoader(DataLoader):
    """One CSV column is featurized, the task columns become y / w (data_loader.py:281-437)."""

    def __init__(self, tasks: List[str], featurizer, feature_field: Optional[str] = None,
                 id_field: Optional[str] = None, smiles_field: Optional[str] = None, log_every_n: int = 1000):
        if not isinstance(tasks, list):
            raise ValueError("tasks must be a list.")
        if smiles_field is not None:
            if feature_field is not None and smiles_field != feature_field:
                raise ValueError("smiles_field and feature_field if both set must have the same value.")
            if feature_field is None:
                feature_field = smiles_field
        self.tasks = tasks
        self.feature_field = feature_field
        self.id_field = feature_field if id_field is None else id_field
        self.featurizer = featurizer
        self.log_every_n = log_every_n

    def _get_shards(self, input_files: List[str], shard_size: Optional[int]):
        return load_csv_files(input_files, shard_size)

    def _featurize_shard(self, shard) -> Tuple[np.ndarray, np.ndarray]:
        if self.featurizer is None:
            raise ValueError("featurizer must be specified in constructor to featurizer data/")
        features = [elt for elt in self.featurizer(shard[self.feature_field])]
        valid_inds = np.array([1 if np.array(elt).size > 0 else 0 for elt in features], dtype=bool)
        features = [elt for (is_valid, elt) in zip(valid_inds, features) if is_valid]
        return np.array(features), valid_inds

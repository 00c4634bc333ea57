"""CSV -> sharded DiskDataset, behind the reference's loader classes (deepchem/data/data_loader.py: label and
weight extraction ``_convert_df_to_numpy`` :35-69, ``DataLoader.create_dataset`` :177-240, ``CSVLoader`` :281-437;
the chunked CSV reader ``load_csv_files`` of utils/data_utils.py:320-350).

The featurizer is any callable ``featurizer(sequence of inputs) -> sequence of features``.  An input whose
features come back empty failed: the row is dropped with its labels, weights and id.  The native featurizers of
``deepchem_amd.feat`` fit this slot; so does ``deepchem.feat.ConvMolFeaturizer()`` where DeepChem is installed."""
from typing import Any, Iterator, List, Optional, Tuple

import numpy as np

from deepchem_amd.data.datasets import Dataset, DiskDataset


def convert_df_to_numpy(df, tasks: List[str]) -> Tuple[np.ndarray, np.ndarray]:
    """Labels and weights ``(rows, len(tasks))`` as float64 from the task columns of a dataframe.  A cell that
    holds the empty string (what ``load_csv_files`` leaves where the file had no value) is a missing label:
    y = 0 under w = 0.  Every other cell keeps its value under w = 1."""
    rows = df.shape[0]
    columns = [np.asarray(df[name].values).reshape(rows, 1) for name in tasks]
    y = np.hstack(columns)
    w = np.ones((rows, len(tasks)))
    if y.dtype.kind in "OU":
        blank = y == ""
        w[blank] = 0
        y[blank] = 0
    return y.astype(float), w.astype(float)


_convert_df_to_numpy = convert_df_to_numpy


def load_csv_files(input_files: List[str], shard_size: Optional[int] = None) -> Iterator[Any]:
    """One dataframe per file, or per ``shard_size`` rows of a file.  Chunked frames have their NaN cells
    replaced by empty strings; a whole-file frame is handed over untouched (the reference's asymmetry)."""
    import pandas as pd
    for path in input_files:
        if shard_size is None:
            yield pd.read_csv(path)
            continue
        for chunk in pd.read_csv(path, chunksize=shard_size):
            yield chunk.replace(np.nan, str(""), regex=True)


def _as_task_list(tasks):
    if not isinstance(tasks, list):
        raise ValueError("tasks must be a list.")
    return tasks


class DataLoader(object):
    """Shard loop shared by the loaders.  A subclass says where shards come from (``_get_shards``: an iterator
    of dataframes) and how one is featurized (``_featurize_shard``: features of the rows that succeeded, plus
    a boolean mask over all rows)."""

    def __init__(self, tasks: List[str], featurizer, id_field: Optional[str] = None, log_every_n: int = 1000):
        if type(self) is DataLoader:
            raise ValueError("DataLoader should never be instantiated directly. Use a subclass instead.")
        self._configure(_as_task_list(tasks), featurizer, id_field, log_every_n)

    def _configure(self, tasks, featurizer, id_field, log_every_n):
        self.tasks, self.featurizer = tasks, featurizer
        self.id_field, self.log_every_n = id_field, log_every_n

    def _rows_of(self, shard):
        """``(X, y, w, ids)`` of the rows of one shard that featurized."""
        X, kept = self._featurize_shard(shard)
        ids = shard[self.id_field].values[kept]
        y = w = None
        if self.tasks:
            y, w = (a[kept] for a in convert_df_to_numpy(shard, self.tasks))
        for column in (ids, y, w):
            assert column is None or len(column) == len(X)
        return X, y, w, ids

    def create_dataset(self, inputs, data_dir: Optional[str] = None, shard_size: Optional[int] = 8192) -> Dataset:
        sources = inputs if isinstance(inputs, list) else [inputs]
        shards = (self._rows_of(shard) for shard in self._get_shards(sources, shard_size))
        return DiskDataset.create_dataset(shards, data_dir, self.tasks)

    def featurize(self, inputs, data_dir: Optional[str] = None, shard_size: Optional[int] = 8192) -> Dataset:
        """The reference's older name for ``create_dataset``."""
        return self.create_dataset(inputs, data_dir, shard_size)

    def _get_shards(self, inputs: List, shard_size: Optional[int]) -> Iterator:
        raise NotImplementedError

    def _featurize_shard(self, shard: Any):
        raise NotImplementedError


class CSVLoader(DataLoader):
    """Featurizes one column of a CSV file; the task columns become y and w.  ``smiles_field`` is the
    deprecated spelling of ``feature_field``; ids default to the featurized column."""

    def __init__(self, tasks: List[str], featurizer, feature_field: Optional[str] = None,
                 id_field: Optional[str] = None, smiles_field: Optional[str] = None, log_every_n: int = 1000):
        tasks = _as_task_list(tasks)
        if smiles_field is not None:
            if feature_field not in (None, smiles_field):
                raise ValueError("smiles_field and feature_field if both set must have the same value.")
            feature_field = smiles_field
        self.feature_field = feature_field
        self._configure(tasks, featurizer, feature_field if id_field is None else id_field, log_every_n)

    def _get_shards(self, input_files: List[str], shard_size: Optional[int]):
        return load_csv_files(input_files, shard_size)

    def _featurize_shard(self, shard) -> Tuple[np.ndarray, np.ndarray]:
        if self.featurizer is None:
            raise ValueError("featurizer must be specified in constructor to featurizer data/")
        produced = list(self.featurizer(shard[self.feature_field]))
        kept = np.fromiter((np.asarray(f).size > 0 for f in produced), dtype=bool, count=len(produced))
        return np.array([f for f, ok in zip(produced, kept) if ok]), kept

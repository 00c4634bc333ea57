"""In-memory dataset with the iteration contract the GraphConv generator relies on
(deepchem/data/datasets.py: ``pad_batch`` :142-218, ``NumpyDataset`` :671,
``iterbatches`` :843-898)."""
import math
from typing import Iterator, Optional, Tuple

import numpy as np

Batch = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]


def pad_batch(batch_size: int, X_b, y_b, w_b, ids_b) -> Batch:
    """Tile X, y and ids up to ``batch_size`` rows; the weights of the padding
    rows are zero, so they do not count in the loss (they DO enter BatchNorm
    statistics, as in the reference)."""
    n = len(X_b)
    if n == batch_size:
        return (X_b, y_b, w_b, ids_b)
    rep = np.arange(batch_size) % n
    X_out = X_b[rep]
    y_out = None if y_b is None else y_b[rep]
    ids_out = ids_b[rep]
    if w_b is None:
        w_out = None
    else:
        w_out = np.zeros((batch_size,) + tuple(w_b.shape[1:]), dtype=w_b.dtype)
        w_out[:n] = w_b
    return (X_out, y_out, w_out, ids_out)


class Dataset(object):
    pass


class NumpyDataset(Dataset):
    """X (any array, e.g. an object array of ConvMol), y (n, tasks), w, ids."""

    def __init__(self, X, y=None, w=None, ids=None, n_tasks: int = 1):
        n_samples = np.shape(X)[0]
        if n_samples > 0:
            if y is None:
                y = np.zeros((n_samples, n_tasks), np.float32)
                if w is None:
                    w = np.zeros((n_samples, 1), np.float32)
            if ids is None:
                ids = np.arange(n_samples)
            if not isinstance(X, np.ndarray):
                X = np.array(X)
            if not isinstance(y, np.ndarray):
                y = np.array(y)
            if w is None:
                if len(y.shape) == 1:
                    w = np.ones(y.shape[0], np.float32)
                else:
                    w = np.ones((y.shape[0], 1), np.float32)
            if not isinstance(w, np.ndarray):
                w = np.array(w)
        self._X = X
        self._y = y
        self._w = w
        self._ids = np.array(ids, dtype=object)

    def __len__(self) -> int:
        return len(self._y)

    @property
    def X(self):
        return self._X

    @property
    def y(self):
        return self._y

    @property
    def w(self):
        return self._w

    @property
    def ids(self):
        return self._ids

    def get_task_names(self):
        if len(self._y.shape) < 2:
            return np.array([0])
        return np.arange(self._y.shape[1])

    def iterbatches(self, batch_size: Optional[int] = None, epochs: int = 1,
                    deterministic: bool = False, pad_batches: bool = False) -> Iterator[Batch]:
        n_samples = self._X.shape[0]
        if batch_size is None:
            batch_size = n_samples
        sample_perm = np.arange(n_samples)
        for _ in range(epochs):
            if not deterministic:
                sample_perm = np.random.permutation(n_samples)
            for b in range(math.ceil(n_samples / batch_size)):
                idx = sample_perm[b * batch_size:min(n_samples, (b + 1) * batch_size)]
                batch = (self._X[idx], self._y[idx], self._w[idx], self._ids[idx])
                if pad_batches:
                    batch = pad_batch(batch_size, *batch)
                yield batch

    def itersamples(self):
        for i in range(self._X.shape[0]):
            yield (self._X[i], self._y[i], self._w[i], self._ids[i])

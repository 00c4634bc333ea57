"""The transformers the MolNet loaders put in front of GraphConvModel
(deepchem/trans/transformers.py): ``Transformer`` (:56), ``undo_transforms`` (:238-268),
``MinMaxTransformer`` (:272), ``NormalizationTransformer`` (:426), ``ClippingTransformer`` (:613),
``LogTransformer`` (:711), ``BalancingTransformer`` (:870).  Host-side NumPy: they run once per
dataset, not per step."""
from typing import List, Optional, Tuple

import numpy as np

from deepchem_amd.data.datasets import Dataset, DiskDataset

Arrays = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]


def _unit_scale(scale):
    """A scale vector with its zeros (constant columns) replaced by one."""
    scale = np.asarray(scale)
    return np.where(scale > 0, scale, np.ones_like(scale))


def _against_trailing_units(vec, like: np.ndarray):
    """``vec`` (one entry per task) shaped to broadcast against predictions ``like`` whose trailing axes may be
    unit axes that are not the task axis (e.g. (N, T, 1)): one axis is appended to ``vec`` per such axis."""
    vec = np.asarray(vec)
    n_tasks = 1 if vec.ndim == 0 else vec.shape[0]
    appended = sum(1 for extent in like.shape if extent == 1 and extent != n_tasks)
    return vec.reshape(vec.shape + (1,) * appended)


class Transformer(object):
    """Abstract base (transformers.py:56-236).  The flags say which of X / y / w / ids a transformer touches;
    a subclass provides ``transform_array`` and, where the map can be inverted, ``untransform``."""

    def __init__(self, transform_X: bool = False, transform_y: bool = False, transform_w: bool = False,
                 transform_ids: bool = False, dataset: Optional[Dataset] = None):
        if type(self).__name__ == "Transformer":
            raise ValueError("Transformer is an abstract superclass and cannot be directly instantiated. "
                             "You probably want to instantiate a concrete subclass instead.")
        flags = dict(transform_X=transform_X, transform_y=transform_y, transform_w=transform_w,
                     transform_ids=transform_ids)
        assert any(flags.values())
        self.__dict__.update(flags)

    def transform_array(self, X, y, w, ids) -> Arrays:
        raise NotImplementedError("Each Transformer is responsible for its own transform_array method.")

    def untransform(self, transformed: np.ndarray) -> np.ndarray:
        raise NotImplementedError("Each Transformer is responsible for its own untransform method.")

    def transform_on_array(self, X, y, w, ids) -> Arrays:
        return self.transform_array(X, y, w, ids)

    def transform(self, dataset: Dataset, parallel: bool = False, out_dir: Optional[str] = None, **kwargs) -> Dataset:
        """A new dataset with this transformer applied shard by shard (on disk under ``out_dir`` if given)."""
        wants_disk = out_dir is not None and not isinstance(dataset, DiskDataset)
        if wants_disk:
            dataset = DiskDataset.from_numpy(dataset.X, dataset.y, dataset.w, dataset.ids)
        shapes = dict(zip("Xywi", dataset.get_shape()))
        for name in "yw":
            if getattr(self, "transform_" + name) and shapes[name] == ():
                raise ValueError("Cannot transform %s when %s_values are not present" % (name, name))
        return dataset.transform(self, out_dir=out_dir, parallel=parallel)


def undo_transforms(y, transformers: List[Transformer]) -> np.ndarray:
    """Predictions (or labels) taken back through the y-transformers, last applied first
    (transformers.py:238-268).  Transformers that do not act on y are skipped."""
    out = np.asarray(y)
    for t in transformers[::-1]:
        out = t.untransform(out) if t.transform_y else out
    return out


def _one_of_X_y(transform_X, transform_y):
    if transform_X and transform_y:
        raise ValueError("Can only transform only one of X and y")


class MinMaxTransformer(Transformer):
    """X or y scaled into [0, 1] by the per-column minimum and maximum of the dataset it was built from
    (transformers.py:272-424); a constant column is shifted only.  Attributes ``X_min`` / ``X_max`` or
    ``y_min`` / ``y_max``."""

    def __init__(self, transform_X: bool = False, transform_y: bool = False, dataset: Optional[Dataset] = None):
        _one_of_X_y(transform_X, transform_y)
        which = "X" if transform_X else ("y" if transform_y else None)
        if dataset is not None and which is not None:
            values = getattr(dataset, which)
            lo, hi = values.min(axis=0), values.max(axis=0)
            if which == "y" and values.ndim > 1:
                assert len(lo) == values.shape[1]
            setattr(self, which + "_min", lo)
            setattr(self, which + "_max", hi)
        super().__init__(transform_X=transform_X, transform_y=transform_y, dataset=dataset)

    def _range(self):
        which = "X" if self.transform_X else "y"
        return getattr(self, which + "_min"), getattr(self, which + "_max")

    def transform_array(self, X, y, w, ids) -> Arrays:
        lo, hi = self._range()
        scaled = np.nan_to_num(((X if self.transform_X else y) - lo) / _unit_scale(hi - lo))
        return (scaled, y, w, ids) if self.transform_X else (X, scaled, w, ids)

    def untransform(self, z: np.ndarray) -> np.ndarray:
        lo, hi = self._range()
        if self.transform_y:
            lo, hi = _against_trailing_units(lo, z), _against_trailing_units(hi, z)
        return z * (hi - lo) + lo


class NormalizationTransformer(Transformer):
    """X or y shifted to zero mean (unless ``move_mean=False``) and scaled to unit standard deviation, both
    from ``dataset.get_statistics`` (transformers.py:426-610).  A constant label column keeps std 1.
    Attributes ``X_means`` / ``X_stds`` or ``y_means`` / ``y_stds``."""

    def __init__(self, transform_X: bool = False, transform_y: bool = False, transform_w: bool = False,
                 dataset: Optional[Dataset] = None, transform_gradients: bool = False, move_mean: bool = True):
        _one_of_X_y(transform_X, transform_y)
        if transform_w:
            raise ValueError("MinMaxTransformer doesn't support w transformation.")  # (sic) the reference's text
        if transform_gradients:
            raise NotImplementedError("transform_gradients is deprecated in the reference and not provided")
        if dataset is not None and transform_X:
            self.X_means, self.X_stds = dataset.get_statistics(X_stats=True, y_stats=False)
        elif dataset is not None and transform_y:
            self.y_means, spread = dataset.get_statistics(X_stats=False, y_stats=True)
            spread = np.array(spread)
            self.y_stds = np.where(spread == 0, 1., spread)
        self.transform_gradients, self.move_mean = transform_gradients, move_mean
        super().__init__(transform_X=transform_X, transform_y=transform_y, transform_w=transform_w, dataset=dataset)

    def _standardize(self, a, mean, spread):
        centred = a - mean if self.move_mean else a
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.nan_to_num(centred / spread)

    def transform_array(self, X, y, w, ids) -> Arrays:
        if self.transform_X:
            X = self._standardize(X, self.X_means, self.X_stds)
        if self.transform_y:
            y = self._standardize(y, self.y_means, self.y_stds)
        return (X, y, w, ids)

    def untransform(self, z: np.ndarray) -> np.ndarray:
        if self.transform_X:
            mean, spread = self.X_means, self.X_stds
        elif self.transform_y:
            mean, spread = _against_trailing_units(self.y_means, z), _against_trailing_units(self.y_stds, z)
        else:
            return z
        return z * spread + mean if self.move_mean else z * spread


class ClippingTransformer(Transformer):
    """X clipped to [-x_max, x_max] and / or y to [-y_max, y_max] (transformers.py:613-708).  Not invertible."""

    def __init__(self, transform_X: bool = False, transform_y: bool = False, dataset: Optional[Dataset] = None,
                 x_max: float = 5., y_max: float = 500.):
        super().__init__(transform_X=transform_X, transform_y=transform_y, dataset=dataset)
        self.x_max, self.y_max = x_max, y_max

    def transform_array(self, X, y, w, ids) -> Arrays:
        if self.transform_X:
            X = np.clip(X, -self.x_max, self.x_max)
        if self.transform_y:
            y = np.clip(y, -self.y_max, self.y_max)
        return (X, y, w, ids)

    def untransform(self, z):
        raise NotImplementedError("Cannot untransform datasets with ClippingTransformer.")


class LogTransformer(Transformer):
    """log(1 + v) on every column of X (or y), or only on the columns listed in ``features`` (``tasks``)
    (transformers.py:711-868); the inverse is exp(v) - 1 on the same columns."""

    def __init__(self, transform_X: bool = False, transform_y: bool = False, features: Optional[List[int]] = None,
                 tasks: Optional[List[str]] = None, dataset: Optional[Dataset] = None):
        _one_of_X_y(transform_X, transform_y)
        self.features, self.tasks = features, tasks
        super().__init__(transform_X=transform_X, transform_y=transform_y, dataset=dataset)

    @staticmethod
    def _on_columns(a, columns, fn):
        if columns is None:
            return fn(a)
        out = np.array(a, dtype=np.float64, copy=True)
        picked = [j for j in range(out.shape[1]) if j in columns]
        out[:, picked] = fn(out[:, picked])
        return out

    def _map(self, X_or_y, fn):
        return self._on_columns(X_or_y, self.features if self.transform_X else self.tasks, fn)

    def transform_array(self, X, y, w, ids) -> Arrays:
        if self.transform_X:
            X = self._map(X, _log_of_one_plus)
        if self.transform_y:
            y = self._map(y, _log_of_one_plus)
        return (X, y, w, ids)

    def untransform(self, z: np.ndarray) -> np.ndarray:
        if not (self.transform_X or self.transform_y):
            return z
        return self._map(z, _exp_minus_one)


def _log_of_one_plus(v):
    return np.log(v + 1)  # the reference's form: not log1p, whose last bits differ


def _exp_minus_one(v):
    return np.exp(v) - 1


def _two_dims(a, name):
    a = a.reshape(len(a), 1) if a.ndim == 1 else a
    if a.ndim != 2:
        raise ValueError("%s must be of shape (N,) or (N, n_tasks)" % name)
    return a


class BalancingTransformer(Transformer):
    """Sample weights rescaled so that within each task every class carries the same total weight
    (transformers.py:870-1018): over the samples of task t with non-zero weight (N_t of them) class c gets
    N_t / count_t(c); a sample with weight zero keeps zero.  ``classes``: sorted label values of the dataset;
    ``weights[t][i]``: weight of ``classes[i]`` in task t."""

    def __init__(self, dataset: Dataset):
        super().__init__(transform_w=True, dataset=dataset)
        y, w = _two_dims(dataset.y, "y"), _two_dims(dataset.w, "w")
        self.classes = sorted(np.unique(y))
        self.weights = []
        for t in range(len(dataset.get_task_names())):
            seen = y[w[:, t] != 0, t]
            per_class = [int((seen == c).sum()) for c in self.classes]
            self.weights.append([len(seen) / float(n) if n > 0 else 0 for n in per_class])

    def transform_array(self, X, y, w, ids) -> Arrays:
        if y.ndim == 1 and w.ndim == 2 and w.shape[1] == 1:
            y = y[:, None]
        if y.ndim not in (1, 2):
            raise ValueError("y must be of shape (N,) or (N, n_tasks)")
        balanced = np.zeros_like(w)
        y_cols, w_cols, out_cols = ((a.reshape(len(a), -1)) for a in (y, w, balanced))
        for t in range(y_cols.shape[1]):
            counted = w_cols[:, t] != 0
            for c, weight in zip(self.classes, self.weights[t]):
                out_cols[counted & (y_cols[:, t] == c), t] = weight
        return (X, y, balanced, ids)

"""Datasets with the iteration contract ``GraphConvModel.default_generator`` relies on
(deepchem/data/datasets.py): ``pad_batch`` (:142-218), ``Dataset`` (:221), ``NumpyDataset``
(:746), ``DiskDataset`` (:1110) with the reference's on-disk format (``metadata.csv.gzip``,
``tasks.json``, ``shard-<i>-{X,y,w,ids}.npy``), so a directory written by DeepChem opens here and
the other way round.

What is different from the reference is WHERE the work happens: iteration is split into a *batch
plan* -- which rows of which shard form each batch, reproducing the reference's shard order,
per-shard shuffles, carry-over between shards and its exact consumption of ``np.random`` -- and the
data movement.  ``iterbatches`` materialises the plan into NumPy batches (the reference's
contract); the GraphConv fast path hands the same plan to the native collation, which writes
batches straight into pinned staging memory (``deepchem_amd.data.packed_dataset``).
"""
import concurrent.futures
import json
import math
import os
import random
import shutil
import sys
import tempfile
from typing import Any, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

Batch = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]


def pad_features(batch_size: int, X_b: np.ndarray) -> np.ndarray:
    """``X_b`` tiled up to exactly ``batch_size`` entries (data/datasets.py:83-139): what the generators of the
    batch-size-bound models (MPNN) do to the features of a short last batch."""
    n = len(X_b)
    if n > batch_size:
        raise ValueError("Cannot pad an array longer than `batch_size`")
    if n == batch_size:
        return X_b
    return X_b[np.arange(batch_size) % n]


def pad_batch(batch_size: int, X_b, y_b, w_b, ids_b) -> Batch:
    """Tile X, y and ids up to ``batch_size`` rows (datasets.py:142-218); the weights of the padding
    rows are zero, so they do not count in the loss (they DO enter BatchNorm statistics, as in the
    reference)."""
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


def _merge_moments(n_a, mean_a, m2_a, x):
    """Chan et al. update of (count, mean, sum of squared deviations) with the rows of ``x``."""
    n_b = x.shape[0]
    if n_b == 0:
        return n_a, mean_a, m2_a
    x = x.astype(np.float64, copy=False)
    mean_b = x.mean(axis=0)
    m2_b = ((x - mean_b) ** 2).sum(axis=0)
    n = n_a + n_b
    delta = mean_b - mean_a
    mean = mean_a + delta * (n_b / n)
    m2 = m2_a + m2_b + delta * delta * (n_a * n_b / n)
    return n, mean, m2


class Dataset(object):
    """Abstract dataset: ``X`` (samples), ``y`` (labels), ``w`` (weights), ``ids`` (datasets.py:221)."""

    def __len__(self) -> int:
        raise NotImplementedError()

    def get_shape(self):
        raise NotImplementedError()

    def get_task_names(self) -> np.ndarray:
        raise NotImplementedError()

    def iterbatches(self, batch_size=None, epochs=1, deterministic=False, pad_batches=False):
        raise NotImplementedError()

    def itersamples(self):
        raise NotImplementedError()

    def _iter_chunks(self) -> Iterator[Batch]:
        """Whole chunks of the data in order (one for in-memory sets, one per shard on disk)."""
        raise NotImplementedError()

    def get_statistics(self, X_stats: bool = True, y_stats: bool = True):
        """Feature / label means and (population) standard deviations (datasets.py:440-492).
        The reference walks sample by sample; here every chunk is reduced as an array and the
        chunks are merged, which gives the same numbers to rounding and scales to sharded data."""
        x_shape, y_shape, _, _ = self.get_shape()
        n_x, X_means, X_m2 = 0, np.zeros(x_shape[1:]), np.zeros(x_shape[1:])
        n_y, y_means, y_m2 = 0, np.zeros(y_shape[1:]), np.zeros(y_shape[1:])
        for X, y, _, _ in self._iter_chunks():
            if X_stats:
                n_x, X_means, X_m2 = _merge_moments(n_x, X_means, X_m2, np.asarray(X))
            if y_stats:
                n_y, y_means, y_m2 = _merge_moments(n_y, y_means, y_m2, np.asarray(y))
        X_stds = np.sqrt(X_m2 / n_x) if n_x >= 2 else np.zeros(x_shape[1:])
        y_stds = np.sqrt(y_m2 / n_y) if n_y >= 2 else np.zeros(y_shape[1:])
        if X_stats and not y_stats:
            return X_means, X_stds
        if y_stats and not X_stats:
            return y_means, y_stds
        if X_stats and y_stats:
            return X_means, X_stds, y_means, y_stds
        return tuple()

    def __repr__(self) -> str:
        x_shape, y_shape, w_shape, _ = self.get_shape()
        return "<%s X.shape: %s, y.shape: %s, w.shape: %s, task_names: %s>" % (
            self.__class__.__name__, x_shape, y_shape, w_shape, self.get_task_names())


class NumpyDataset(Dataset):
    """X (any array, e.g. an object array of ConvMol), y (n, tasks), w, ids (datasets.py:746-1099)."""

    def __init__(self, X, y=None, w=None, ids=None, n_tasks: int = 1):
        rows = np.shape(X)[0]
        if rows > 0:  # an empty set keeps whatever it was given
            X = X if isinstance(X, np.ndarray) else np.array(X)
            unlabelled = y is None
            if unlabelled:
                # no labels: zero labels, and (unless weights came along) zero weights, so nothing trains on them
                y = np.zeros((rows, n_tasks), np.float32)
                w = np.zeros((rows, 1), np.float32) if w is None else w
            y = y if isinstance(y, np.ndarray) else np.array(y)
            if w is None:  # labels without weights: every sample counts once
                w = np.ones(y.shape[:1] if y.ndim == 1 else (y.shape[0], 1), np.float32)
            w = w if isinstance(w, np.ndarray) else np.array(w)
            ids = np.arange(rows) if ids is None else ids
        self._X, self._y, self._w = X, y, w
        self._ids = np.array(ids, dtype=object)

    def __len__(self) -> int:
        return len(self._y)

    def get_shape(self):
        return self._X.shape, self._y.shape, self._w.shape, self._ids.shape

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
        """datasets.py:843-898: one permutation of the whole set per epoch."""
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

    def _iter_chunks(self):
        yield (self._X, self._y, self._w, self._ids)

    def transform(self, transformer, **args) -> "NumpyDataset":
        newx, newy, neww, newids = transformer.transform_array(self._X, self._y, self._w, self._ids)
        return NumpyDataset(newx, newy, neww, newids)

    def select(self, indices, select_dir: Optional[str] = None) -> "NumpyDataset":
        return NumpyDataset(self.X[indices], self.y[indices], self.w[indices], self.ids[indices])

    @staticmethod
    def from_DiskDataset(ds: "DiskDataset") -> "NumpyDataset":
        return NumpyDataset(ds.X, ds.y, ds.w, ds.ids)

    @staticmethod
    def merge(datasets: Sequence[Dataset]) -> "NumpyDataset":
        X, y, w, ids = (np.concatenate([getattr(d, part) for d in datasets], axis=0) if len(datasets) > 1
                        else getattr(datasets[0], part) for part in ("X", "y", "w", "ids"))
        return NumpyDataset(X, y, w, ids, n_tasks=y.shape[1] if y.ndim > 1 else 1)


# ------------------------------------------------------------------------------------------ disk
_COLUMNS = ("ids", "X", "y", "w", "ids_shape", "X_shape", "y_shape", "w_shape")


def _alias_reference_modules():
    """Object arrays written by DeepChem pickle ``deepchem.feat.mol_graphs.ConvMol``; when DeepChem
    itself is not installed, resolve that path to this package's class for the duration of a load."""
    if "deepchem" in sys.modules:
        return []
    try:
        import importlib.util
        if importlib.util.find_spec("deepchem") is not None:
            return []
    except (ImportError, ValueError):
        pass
    import types
    import deepchem_amd.feat.mol_graphs as mg
    pkg, feat = types.ModuleType("deepchem"), types.ModuleType("deepchem.feat")
    pkg.feat, feat.mol_graphs = feat, mg
    added = {"deepchem": pkg, "deepchem.feat": feat, "deepchem.feat.mol_graphs": mg}
    sys.modules.update(added)
    return list(added)


def _load_npy(path: str) -> np.ndarray:
    try:
        return np.load(path, allow_pickle=False)
    except ValueError:
        pass
    added = _alias_reference_modules()
    try:
        return np.load(path, allow_pickle=True)
    finally:
        for k in added:
            sys.modules.pop(k, None)


def _shape_to_str(shape) -> Optional[str]:
    return None if shape is None else str(tuple(int(s) for s in shape))


def _str_to_shape(s) -> Tuple[int, ...]:
    if s is None or (isinstance(s, float) and math.isnan(s)):
        return tuple()
    if isinstance(s, (tuple, list)):
        return tuple(int(v) for v in s)
    body = str(s).strip().strip("()[]")
    return tuple(int(v) for v in body.split(",") if v.strip())


class _Shard(object):

    def __init__(self, X, y, w, ids):
        self.X, self.y, self.w, self.ids = X, y, w, ids


class DiskDataset(Dataset):
    """A dataset stored as shards on disk (datasets.py:1110-2721).  Same files, same metadata,
    same batch sequences for the same ``np.random`` state as the reference."""

    def __init__(self, data_dir: str) -> None:
        self.data_dir = data_dir
        tasks, self.metadata_df = self.load_metadata()
        self.tasks = np.array(tasks)
        cols = list(self.metadata_df.columns)
        if cols == ["ids", "X", "y", "w"]:
            self.legacy_metadata = True
        elif cols == list(_COLUMNS):
            self.legacy_metadata = False
        else:
            raise ValueError(
                "Malformed metadata on disk. Metadata must have columns 'ids', 'X', 'y', 'w', "
                "'ids_shape', 'X_shape', 'y_shape', 'w_shape' (or if in legacy metadata format,"
                "columns 'ids', 'X', 'y', 'w')")
        self._cached_shards: Optional[List] = None
        self._memory_cache_size = 20 * (1 << 20)
        self._cache_used = 0

    # ---------------------------------------------------------------- metadata
    def _get_metadata_filename(self) -> Tuple[str, str]:
        return (os.path.join(self.data_dir, "tasks.json"), os.path.join(self.data_dir, "metadata.csv.gzip"))

    def load_metadata(self):
        import pandas as pd
        tasks_filename, metadata_filename = self._get_metadata_filename()
        try:
            with open(tasks_filename) as fin:
                tasks = json.load(fin)
            metadata_df = pd.read_csv(metadata_filename, compression="gzip", dtype=object)
            metadata_df = metadata_df.astype(object).where(pd.notnull(metadata_df), None)
            return tasks, metadata_df
        except Exception:
            pass
        legacy = os.path.join(self.data_dir, "metadata.joblib")
        if os.path.exists(legacy):
            import joblib
            tasks, metadata_df = joblib.load(legacy)
            del metadata_df["task_names"]
            del metadata_df["basename"]
            DiskDataset._save_metadata(metadata_df, self.data_dir, tasks)
            return tasks, metadata_df
        raise ValueError("No Metadata found in the path %s" % self.data_dir)

    @staticmethod
    def _save_metadata(metadata_df, data_dir: str, tasks) -> None:
        if tasks is None:
            tasks = []
        elif isinstance(tasks, np.ndarray):
            tasks = tasks.tolist()
        tasks = [t.item() if isinstance(t, np.generic) else t for t in tasks]
        with open(os.path.join(data_dir, "tasks.json"), "w") as fout:
            json.dump(tasks, fout)
        metadata_df.to_csv(os.path.join(data_dir, "metadata.csv.gzip"), index=False, compression="gzip")

    @staticmethod
    def _construct_metadata(metadata_entries: List):
        import pandas as pd
        return pd.DataFrame(metadata_entries, columns=_COLUMNS)

    @staticmethod
    def write_data_to_disk(data_dir: str, basename: str, X=None, y=None, w=None, ids=None) -> List[Any]:
        """One shard -> ``<basename>-{X,y,w,ids}.npy`` (datasets.py:1359-1427); returns its
        metadata row."""
        out = {}
        for name, arr in (("ids", ids), ("X", X), ("y", y), ("w", w)):
            if arr is not None:
                fname = "%s-%s.npy" % (basename, name)
                np.save(os.path.join(data_dir, fname), arr)
                out[name] = (fname, tuple(np.shape(arr)))
            else:
                out[name] = (None, None)
        return [out["ids"][0], out["X"][0], out["y"][0], out["w"][0],
                out["ids"][1], out["X"][1], out["y"][1], out["w"][1]]

    @staticmethod
    def create_dataset(shard_generator: Iterable[Batch], data_dir: Optional[str] = None,
                       tasks=None) -> "DiskDataset":
        if data_dir is None:
            data_dir = tempfile.mkdtemp()
        elif not os.path.exists(data_dir):
            os.makedirs(data_dir)
        rows = []
        for shard_num, (X, y, w, ids) in enumerate(shard_generator):
            if shard_num == 0 and tasks is None and y is not None:
                tasks = np.array([0]) if y.ndim < 2 else np.arange(y.shape[1])
            rows.append(DiskDataset.write_data_to_disk(data_dir, "shard-%d" % shard_num, X, y, w, ids))
        DiskDataset._save_metadata(DiskDataset._construct_metadata(rows), data_dir, tasks)
        return DiskDataset(data_dir)

    @staticmethod
    def from_numpy(X, y=None, w=None, ids=None, tasks=None, data_dir: Optional[str] = None) -> "DiskDataset":
        dataset = NumpyDataset(X, y, w, ids)
        if tasks is None:
            tasks = dataset.get_task_names()
        return DiskDataset.create_dataset([(dataset.X, dataset.y, dataset.w, dataset.ids)],
                                          data_dir=data_dir, tasks=tasks)

    def _content_changed(self) -> None:
        """Called by every in-place mutator: derived copies (the packed molecule set of the native batch
        pipeline, with the resident set and labels in HBM behind it) describe the old rows."""
        self.__dict__.pop("_gcmi_packed", None)

    def save_to_disk(self) -> None:
        self._content_changed()
        DiskDataset._save_metadata(self.metadata_df, self.data_dir, self.tasks)
        self._cached_shards = None

    def move(self, new_data_dir: str, delete_if_exists: bool = True) -> None:
        if delete_if_exists and os.path.isdir(new_data_dir):
            shutil.rmtree(new_data_dir)
        shutil.move(self.data_dir, new_data_dir)
        self.data_dir = new_data_dir if delete_if_exists else os.path.join(
            new_data_dir, os.path.basename(self.data_dir))

    def copy(self, new_data_dir: str) -> "DiskDataset":
        if os.path.isdir(new_data_dir):
            shutil.rmtree(new_data_dir)
        shutil.copytree(self.data_dir, new_data_dir)
        return DiskDataset(new_data_dir)

    def get_task_names(self) -> np.ndarray:
        return self.tasks

    def get_number_shards(self) -> int:
        return self.metadata_df.shape[0]

    # ---------------------------------------------------------------- shards
    @property
    def memory_cache_size(self) -> int:
        return self._memory_cache_size

    @memory_cache_size.setter
    def memory_cache_size(self, size: int) -> None:
        self._memory_cache_size = size
        if self._cache_used > size:
            self._cached_shards = None

    def get_shard(self, i: int) -> Batch:
        """Shard ``i`` from the cache or from disk (datasets.py:2204-2270); shards are cached in
        arrival order until ``memory_cache_size`` is used up, nothing is ever evicted."""
        if self._cached_shards is None:
            self._cached_shards = [None] * self.get_number_shards()
            self._cache_used = 0
        if self._cached_shards[i] is not None:
            s = self._cached_shards[i]
            return (s.X, s.y, s.w, s.ids)
        row = self.metadata_df.iloc[i]
        X = np.array(_load_npy(os.path.join(self.data_dir, row["X"])))
        y = np.array(_load_npy(os.path.join(self.data_dir, row["y"]))) if row["y"] is not None else None
        if row["w"] is not None:
            w_filename = os.path.join(self.data_dir, row["w"])
            if os.path.exists(w_filename):
                w = np.array(_load_npy(w_filename))
            elif y is not None:
                w = np.ones(y.shape[0], np.float32) if len(y.shape) == 1 else np.ones((y.shape[0], 1), np.float32)
            else:
                w = None
        else:
            w = None
        ids = np.array(_load_npy(os.path.join(self.data_dir, row["ids"])), dtype=object)
        shard = _Shard(X, y, w, ids)
        size = X.nbytes + ids.nbytes + (y.nbytes if y is not None else 0) + (w.nbytes if w is not None else 0)
        if self._cache_used + size < self._memory_cache_size:
            self._cached_shards[i] = shard
            self._cache_used += size
        return (X, y, w, ids)

    def _get_shard_part(self, i: int, col: str, dtype=None):
        if self._cached_shards is not None and self._cached_shards[i] is not None:
            return getattr(self._cached_shards[i], col)
        row = self.metadata_df.iloc[i]
        if row[col] is None:
            return None
        return np.array(_load_npy(os.path.join(self.data_dir, row[col])), dtype=dtype)

    def get_shard_ids(self, i: int) -> np.ndarray:
        return self._get_shard_part(i, "ids", object)

    def get_shard_y(self, i: int) -> np.ndarray:
        return self._get_shard_part(i, "y")

    def get_shard_w(self, i: int) -> np.ndarray:
        return self._get_shard_part(i, "w")

    def add_shard(self, X, y=None, w=None, ids=None) -> None:
        rows = self.metadata_df.values.tolist()
        rows.append(DiskDataset.write_data_to_disk(self.data_dir, "shard-%d" % self.get_number_shards(),
                                                   X, y, w, ids))
        self.metadata_df = DiskDataset._construct_metadata(rows)
        self.save_to_disk()

    def set_shard(self, shard_num: int, X, y=None, w=None, ids=None) -> None:
        DiskDataset.write_data_to_disk(self.data_dir, "shard-%d" % shard_num, X, y, w, ids)
        self._cached_shards = None
        self._content_changed()
        self.legacy_metadata = True

    def itershards(self) -> Iterator[Batch]:
        return (self.get_shard(i) for i in range(self.get_number_shards()))

    def _iter_chunks(self):
        return self.itershards()

    def _get_shard_shape(self, shard_num: int):
        if self.legacy_metadata:
            raise ValueError("This function requires the new metadata format to be called. Please reshard "
                             "this dataset by calling the reshard() method.")
        row = self.metadata_df.iloc[shard_num]
        has_tasks = len(self.get_task_names()) > 0
        return (_str_to_shape(row["X_shape"]), _str_to_shape(row["y_shape"]) if has_tasks else tuple(),
                _str_to_shape(row["w_shape"]) if has_tasks else tuple(), _str_to_shape(row["ids_shape"]))

    def get_shape(self):
        """Total shapes of X, y, w, ids (datasets.py:2667-2712): leading dimensions add up over
        the shards, trailing dimensions come from the first non-empty shard."""
        n_rows = self.get_number_shards()
        if n_rows == 0:
            raise ValueError("No data in dataset.")
        totals = [None, None, None, None]
        for shard_num in range(n_rows):
            if self.legacy_metadata:
                X, y, w, ids = self.get_shard(shard_num)
                shapes = [np.shape(X), np.shape(y) if y is not None else tuple(),
                          np.shape(w) if w is not None else tuple(), np.shape(ids)]
            else:
                shapes = list(self._get_shard_shape(shard_num))
            for k, shp in enumerate(shapes):
                if totals[k] is None:
                    totals[k] = list(shp) if len(shp) else None
                elif len(shp):
                    totals[k][0] += shp[0]
        out = [tuple(t) if t is not None else tuple() for t in totals]
        return out[0], out[1], out[2], out[3]

    def get_data_shape(self):
        if not len(self.metadata_df):
            raise ValueError("No data in dataset.")
        if self.legacy_metadata:
            X, _, _, _ = next(self.itershards())
            return X.shape[1:]
        X_shape, _, _, _ = self._get_shard_shape(0)
        return X_shape[1:]

    def get_shard_size(self) -> int:
        if not len(self.metadata_df):
            raise ValueError("No data in dataset.")
        return len(self.get_shard_ids(0))

    def __len__(self) -> int:
        total = 0
        for i in range(self.get_number_shards()):
            if self.legacy_metadata:
                total += len(self.get_shard_ids(i))
            else:
                shp = self._get_shard_shape(i)[3]
                total += shp[0] if len(shp) else 0
        return total

    def _concat(self, which: int, empty_dtype=None):
        parts = []
        for shard in self.itershards():
            a = shard[which]
            if a is not None and len(a):
                parts.append(a)
        if not parts:
            return np.array([], dtype=empty_dtype)
        return np.concatenate(parts, axis=0) if which != 3 else np.concatenate(parts)

    @property
    def X(self) -> np.ndarray:
        return self._concat(0)

    @property
    def y(self) -> np.ndarray:
        return self._concat(1)

    @property
    def w(self) -> np.ndarray:
        return self._concat(2)

    @property
    def ids(self) -> np.ndarray:
        return np.array(self._concat(3, object), dtype=object)

    # ---------------------------------------------------------------- iteration
    def batch_plan(self, shard_indices: Optional[Sequence[int]] = None, batch_size: Optional[int] = None,
                   epochs: int = 1, deterministic: bool = False):
        """The batches of ``_iterbatches_from_shards`` (datasets.py:1651-1766) as index lists:
        yields ``(shard_of_row, row_in_shard, wants_padding)`` per batch, WITHOUT touching sample data.

        Reference behaviour kept: one permutation of the shard order per epoch; each visited shard
        is prefixed by the incomplete batch carried over from the previous one and the rows of
        that concatenation are permuted together; an incomplete batch is only emitted from the
        last shard; the loop runs until ceil(len(dataset) / batch_size) batches were produced, the
        count being taken over the WHOLE dataset even when only some shards are walked (the last
        shard is then revisited, as in the reference); ``np.random`` is consumed in the same order.
        """
        if shard_indices is None:
            shard_indices = list(range(self.get_number_shards()))
        shard_indices = list(shard_indices)
        num_shards = len(shard_indices)
        lens = {}

        def shard_len(s):
            if s not in lens:
                if self.legacy_metadata:
                    lens[s] = len(self.get_shard_ids(s))
                else:
                    shp = self._get_shard_shape(s)[0]
                    lens[s] = shp[0] if len(shp) else 0
            return lens[s]

        if batch_size is None:
            num_global_batches = num_shards
        else:
            num_global_batches = math.ceil(self.get_shape()[0][0] / batch_size)
        shard_perm = np.arange(num_shards)
        for epoch in range(epochs):
            if not deterministic:
                shard_perm = np.random.permutation(num_shards)
            cur_global_batch, cur_shard = 0, 0
            carry = None
            fetched = None
            while cur_global_batch < num_global_batches:
                if cur_shard < num_shards:
                    fetched = shard_indices[shard_perm[cur_shard]]
                elif fetched is None:
                    return
                s = fetched  # past the end the reference's prefetch handle returns the last shard again
                n_s = shard_len(s)
                src_shard = np.full(n_s, s, dtype=np.int64)
                src_row = np.arange(n_s, dtype=np.int64)
                if carry is not None:
                    src_shard = np.concatenate([carry[0], src_shard])
                    src_row = np.concatenate([carry[1], src_row])
                    carry = None
                n = src_shard.shape[0]
                if n == 0:
                    cur_shard += 1
                    if batch_size is None:
                        cur_global_batch += 1
                    continue
                bs = n if batch_size is None else batch_size
                num_local = math.ceil(n / bs)
                sample_perm = np.arange(n) if deterministic else np.random.permutation(n)
                for b in range(num_local):
                    sel = sample_perm[b * bs:min(n, (b + 1) * bs)]
                    if len(sel) < bs and cur_shard != num_shards - 1:
                        carry = (src_shard[sel], src_row[sel])
                    else:
                        yield src_shard[sel], src_row[sel], bs
                        cur_global_batch += 1
                cur_shard += 1

    def _materialize(self, shard_of_row, row_in_shard, next_hint=None) -> Batch:
        parts = []
        # runs of rows from the same shard, in batch order
        bounds = np.flatnonzero(np.diff(shard_of_row)) + 1
        starts = np.concatenate([[0], bounds])
        ends = np.concatenate([bounds, [len(shard_of_row)]])
        for a, b in zip(starts, ends):
            X, y, w, ids = self.get_shard(int(shard_of_row[a]))
            r = row_in_shard[a:b]
            parts.append((X[r], None if y is None else y[r], None if w is None else w[r], ids[r]))
        if len(parts) == 1:
            return parts[0]

        def cat(k):
            if parts[0][k] is None:
                return None
            return np.concatenate([p[k] for p in parts], axis=0)
        return cat(0), cat(1), cat(2), cat(3)

    def _iterbatches_from_shards(self, shard_indices: Sequence[int], batch_size: Optional[int] = None,
                                 epochs: int = 1, deterministic: bool = False,
                                 pad_batches: bool = False) -> Iterator[Batch]:
        """datasets.py:1651-1766 (see ``batch_plan``).  The next shard is read ahead on one worker
        thread, as the reference does with its one-thread pool."""
        pool = concurrent.futures.ThreadPoolExecutor(1)
        try:
            plan = self.batch_plan(shard_indices, batch_size, epochs, deterministic)
            pending = next(plan, None)
            while pending is not None:
                shard_of_row, row_in_shard, bs = pending
                pending = next(plan, None)
                if pending is not None and len(pending[0]):  # read ahead: warm the shard cache
                    nxt = int(pending[0][-1])
                    if self._cached_shards is None or self._cached_shards[nxt] is None:
                        pool.submit(self.get_shard, nxt)
                X_b, y_b, w_b, ids_b = self._materialize(shard_of_row, row_in_shard)
                if pad_batches:
                    X_b, y_b, w_b, ids_b = pad_batch(bs, X_b, y_b, w_b, ids_b)
                yield X_b, y_b, w_b, ids_b
        finally:
            pool.shutdown(wait=False)

    def iterbatches(self, batch_size: Optional[int] = None, epochs: int = 1, deterministic: bool = False,
                    pad_batches: bool = False) -> Iterator[Batch]:
        return self._iterbatches_from_shards(list(range(self.get_number_shards())), batch_size, epochs,
                                             deterministic, pad_batches)

    def iterbatches_for_rank(self, rank: int, world_size: int, batch_size: Optional[int] = None,
                             epochs: int = 1, deterministic: bool = False) -> Iterator[Batch]:
        """The shards rank ``rank`` of ``world_size`` walks: ``_TorchDiskDataset.__iter__``
        (data/pytorch_datasets.py:95-121) -- contiguous shard ranges per process."""
        n_shards = self.get_number_shards()
        first = (rank * n_shards) // world_size
        last = ((rank + 1) * n_shards) // world_size
        if first == last:
            return iter(())
        return self._iterbatches_from_shards(list(range(first, last)), batch_size, epochs, deterministic)

    def itersamples(self):
        for X, y, w, ids in self.itershards():
            for i in range(X.shape[0]):
                yield (X[i], None if y is None else y[i], None if w is None else w[i], ids[i])

    # ---------------------------------------------------------------- derived datasets
    def transform(self, transformer, parallel: bool = False, out_dir: Optional[str] = None, **args) -> "DiskDataset":
        """Apply ``transformer.transform_array`` shard by shard into a new dataset
        (datasets.py:1800-1878)."""
        if out_dir is None:
            out_dir = tempfile.mkdtemp()
        tasks = self.get_task_names()

        def generator():
            for shard_num in range(self.get_number_shards()):
                X, y, w, ids = self.get_shard(shard_num)
                yield transformer.transform_array(X, y, w, ids)
        return DiskDataset.create_dataset(generator(), data_dir=out_dir, tasks=tasks)

    def reshard(self, shard_size: int) -> None:
        """Rewrite the data in shards of ``shard_size`` rows, in place (datasets.py:1491-1568)."""
        reshard_dir = tempfile.mkdtemp()
        n_tasks = len(self.get_task_names())
        _, y_shape, w_shape, _ = self.get_shape()
        # per-row shapes of X, y, w (a 1-d label or weight array is given the reference's (1, n_tasks) reading)
        row_shapes = [tuple(self.get_data_shape())] + [
            tuple(((len(shape), n_tasks) if len(shape) == 1 else shape)[1:]) for shape in (y_shape, w_shape)]

        def generator():
            # rows waiting to fill a shard, [X, y, w, ids]; they start as empty float64 arrays, so the
            # rewritten shards are float64 whatever the old ones were (as in the reference)
            pending = [np.zeros((0,) + tail) for tail in row_shapes] + [np.zeros((0,), dtype=object)]
            for shard in self.itershards():
                arrived = [None if a is None else np.reshape(a, (len(a),) + tail)
                           for a, tail in zip(shard[:3], row_shapes)] + [shard[3]]
                if arrived[1] is None:  # an unlabelled shard re-appends the waiting labels (reference quirk)
                    arrived[1], arrived[2] = pending[1], pending[2]
                pending = [np.concatenate([have, new], axis=0) for have, new in zip(pending, arrived)]
                while len(pending[0]) > shard_size:
                    yield tuple(a[:shard_size] for a in pending)
                    pending = [a[shard_size:] for a in pending]
            yield tuple(pending)

        resharded = DiskDataset.create_dataset(generator(), data_dir=reshard_dir, tasks=self.tasks)
        shutil.rmtree(self.data_dir)
        shutil.move(reshard_dir, self.data_dir)
        self.legacy_metadata = False
        self.metadata_df = resharded.metadata_df
        self._content_changed()
        self.save_to_disk()

    def subset(self, shard_nums: Sequence[int], subset_dir: Optional[str] = None) -> "DiskDataset":
        if subset_dir is not None:
            os.makedirs(subset_dir, exist_ok=True)
        else:
            subset_dir = tempfile.mkdtemp()
        wanted = set(int(s) for s in shard_nums)
        return DiskDataset.create_dataset((self.get_shard(i) for i in range(self.get_number_shards()) if i in wanted),
                                          data_dir=subset_dir, tasks=self.get_task_names())

    @staticmethod
    def merge(datasets: Iterable[Dataset], merge_dir: Optional[str] = None) -> "DiskDataset":
        if merge_dir is not None:
            os.makedirs(merge_dir, exist_ok=True)
        else:
            merge_dir = tempfile.mkdtemp()
        datasets = list(datasets)
        tasks = []
        for d in datasets:
            try:
                tasks.append(list(d.get_task_names()))
            except Exception:
                pass
        if tasks and any(t != tasks[0] for t in tasks):
            raise ValueError("Cannot merge datasets with different task specifications")
        task_names = tasks[0] if tasks else []

        def generator():
            for d in datasets:
                for chunk in d._iter_chunks():
                    yield chunk
        return DiskDataset.create_dataset(generator(), data_dir=merge_dir, tasks=task_names)

    def select(self, indices, select_dir: Optional[str] = None, select_shard_size: Optional[int] = None,
               output_numpy_dataset: bool = False) -> Dataset:
        """Rows ``indices`` (in that order) as a new dataset (datasets.py:2386-2555), written in
        shards of ``select_shard_size`` rows; every source shard is read at most once per output
        shard."""
        if output_numpy_dataset and (select_dir is not None or select_shard_size is not None):
            raise ValueError("If output_numpy_dataset is set, then select_dir and select_shard_size must both be None")
        indices = np.asarray(indices, dtype=np.int64)
        N = len(indices)
        tasks = self.get_task_names()
        if output_numpy_dataset:
            select_shard_size = N
        else:
            if select_dir is not None:
                os.makedirs(select_dir, exist_ok=True)
            else:
                select_dir = tempfile.mkdtemp()
            if select_shard_size is None:
                select_shard_size = self.get_shard_size()
        if not N:
            if output_numpy_dataset:
                return NumpyDataset(np.array([]), np.array([]), np.array([]), np.array([]))
            return DiskDataset.create_dataset([], data_dir=select_dir, tasks=tasks)
        n_shards = self.get_number_shards()
        lens = [len(self.get_shard_ids(s)) if self.legacy_metadata else
                (self._get_shard_shape(s)[0][0] if len(self._get_shard_shape(s)[0]) else 0) for s in range(n_shards)]
        offsets = np.concatenate([[0], np.cumsum(lens)])

        def generator():
            for start in range(0, N, select_shard_size):
                want = indices[start:start + select_shard_size]
                shard_of = np.searchsorted(offsets, want, side="right") - 1
                Xo = yo = wo = io = None
                for s in np.unique(shard_of):
                    X, y, w, ids = self.get_shard(int(s))
                    pos = np.flatnonzero(shard_of == s)
                    rows = want[pos] - offsets[s]
                    if Xo is None:
                        Xo = np.empty((len(want),) + X.shape[1:], X.dtype)
                        yo = None if y is None else np.empty((len(want),) + y.shape[1:], y.dtype)
                        wo = None if w is None else np.empty((len(want),) + w.shape[1:], w.dtype)
                        io = np.empty((len(want),), object)
                    Xo[pos] = X[rows]
                    if yo is not None:
                        yo[pos] = y[rows]
                    if wo is not None:
                        wo[pos] = w[rows]
                    io[pos] = ids[rows]
                yield (Xo, np.array([]) if yo is None else yo, np.array([]) if wo is None else wo, io)

        if output_numpy_dataset:
            X, y, w, ids = next(generator())
            return NumpyDataset(X, y, w, ids)
        return DiskDataset.create_dataset(generator(), data_dir=select_dir, tasks=tasks)

    # ---------------------------------------------------------------- shuffles
    def complete_shuffle(self, data_dir: Optional[str] = None) -> Dataset:
        perm = np.random.permutation(len(self)).tolist()
        return self.select(perm, data_dir, self.get_shard_size())

    def shuffle_each_shard(self, shard_basenames: Optional[List[str]] = None) -> None:
        n_rows = len(self.metadata_df.index)
        if shard_basenames is not None:
            if len(shard_basenames) != n_rows:
                raise ValueError("shard_basenames must provide a basename for each shard in this DiskDataset.")
        else:
            shard_basenames = ["shard-%d" % i for i in range(n_rows)]
        for i, basename in enumerate(shard_basenames):
            X, y, w, ids = self.get_shard(i)
            p = np.random.permutation(X.shape[0])
            DiskDataset.write_data_to_disk(self.data_dir, basename, X[p], y[p], w[p], ids[p])
        self._cached_shards = None
        self._content_changed()

    def shuffle_shards(self) -> None:
        rows = self.metadata_df.values.tolist()
        random.shuffle(rows)
        self.metadata_df = DiskDataset._construct_metadata(rows)
        self.save_to_disk()

    def sparse_shuffle(self) -> None:
        """Shuffle all rows across shards in memory and write them back, in place
        (datasets.py:2082-2133; the reference compresses sparse rows first -- the row permutation
        and the resulting files are the same)."""
        shard_size = self.get_shard_size()
        num_shards = self.get_number_shards()
        X, y, w, ids = self.X, self.y, self.w, self.ids
        p = np.random.permutation(len(X))
        X, y, w, ids = X[p], y[p], w[p], ids[p]
        for i in range(num_shards):
            a, b = i * shard_size, (i + 1) * shard_size
            self.set_shard(i, X[a:b], y[a:b], w[a:b], ids[a:b])

    def get_label_means(self):
        import pandas as pd
        return pd.Series(np.asarray(self.y, np.float64).mean(axis=0))

    def get_label_stds(self):
        import pandas as pd
        return pd.Series(np.asarray(self.y, np.float64).std(axis=0))

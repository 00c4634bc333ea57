"""Molecule sets as flat arrays, and the prefetching batch pipeline on top of them.

``PackedDataset`` is the dataset form of ``PackedMols`` (+ labels, weights, ids): what a
featurized MolNet dataset becomes once its ``ConvMol`` objects are concatenated.  It follows
``NumpyDataset.iterbatches`` (deepchem/data/datasets.py:843-898) for order, shuffling and padding
but hands out molecule INDICES instead of object arrays, so that a batch can be collated natively
(``gcmi_collate``) straight into a pinned arena.

``DeviceBatchPipeline`` runs collation + host-to-device copy on a worker thread and a side HIP
stream, ``depth`` batches ahead of the training loop (the reference collates each batch in
Python on the training thread, ~2.5 ms per 100 molecules)."""
import collections
import logging
import math
import os
import queue
import threading
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from deepchem_amd.data.collate import DeviceBatch, PinnedRing, collate_to_device
from deepchem_amd.data.datasets import Dataset
from deepchem_amd.utils.synthetic import PackedMols


logger = logging.getLogger(__name__)


class PackedDataset(Dataset):

    def __init__(self, packed: PackedMols, y=None, w=None, ids=None, n_tasks: int = 1):
        n = packed.n_mols
        if y is None:
            y = np.zeros((n, n_tasks), np.float32)
            if w is None:
                w = np.zeros((n, 1), np.float32)
        y = np.asarray(y)
        if w is None:
            w = np.ones((n,) if y.ndim == 1 else (n, 1), np.float32)
        if len(y) != n or len(w) != n:
            raise ValueError("labels / weights do not match the %d molecules" % n)
        self.packed = packed
        self._y = y
        self._w = np.asarray(w)
        self._ids = np.arange(n) if ids is None else np.asarray(ids, dtype=object)

    def __len__(self) -> int:
        return self.packed.n_mols

    @property
    def X(self):
        from deepchem_amd.feat.mol_graphs import convmols_from_packed
        return convmols_from_packed(self.packed)

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
        """Task indices, as NumpyDataset names them (data/datasets.py:793-797)."""
        return np.array([0]) if self._y.ndim < 2 else np.arange(self._y.shape[1])

    def get_shape(self):
        return (len(self),), self._y.shape, self._w.shape, self._ids.shape

    def iter_index_batches(self, batch_size: int, epochs: int = 1, deterministic: bool = False,
                           pad_batches: bool = False) -> Iterator[Tuple[np.ndarray, int]]:
        """(molecule indices, number of real molecules) per batch; with ``pad_batches`` the
        indices are tiled up to batch_size (pad_batch, deepchem/data/datasets.py:142-218)."""
        n = len(self)
        perm = np.arange(n)
        for _ in range(epochs):
            if not deterministic:
                perm = np.random.permutation(n)
            for b in range(math.ceil(n / batch_size)):
                idx = perm[b * batch_size:min(n, (b + 1) * batch_size)]
                n_real = idx.shape[0]
                if pad_batches and n_real < batch_size:
                    idx = idx[np.arange(batch_size) % n_real]
                yield idx, n_real

    def iterbatches(self, batch_size: Optional[int] = None, epochs: int = 1,
                    deterministic: bool = False, pad_batches: bool = False):
        """Reference-shaped batches (X as ConvMol objects) for callers that want them."""
        from deepchem_amd.feat.mol_graphs import convmols_from_packed
        for idx, n_real in self.iter_index_batches(batch_size or len(self), epochs, deterministic,
                                                   pad_batches):
            w_b = self._w[idx].copy()
            w_b[n_real:] = 0
            yield (convmols_from_packed(self.packed.select(idx)), self._y[idx], w_b, self._ids[idx])


def packed_from_convmols(X) -> PackedMols:
    """Concatenate ConvMol objects (deepchem_amd's or DeepChem's own) into a PackedMols, keeping
    each molecule's internal (degree-sorted) atom order, so that collating the result gives exactly
    what ``ConvMol.agglomerate_mols`` gives."""
    feats, sizes, degs, idx = [], [], [], []
    for m in X:
        f = np.asarray(m.get_atom_features(), dtype=np.float32)
        feats.append(f)
        sizes.append(f.shape[0])
        if hasattr(m, "adj_ptr"):
            degs.append(np.diff(m.adj_ptr))
            idx.append(np.asarray(m.adj_idx, np.int32))
        else:  # a DeepChem ConvMol: list-of-lists adjacency in the new order
            adj = m.get_adjacency_list()
            degs.append(np.fromiter((len(a) for a in adj), np.int64, len(adj)))
            idx.append(np.fromiter((j for a in adj for j in a), np.int32))
    atom_ptr = np.zeros(len(sizes) + 1, np.int64)
    np.cumsum(sizes, out=atom_ptr[1:])
    deg = np.concatenate(degs) if degs else np.zeros(0, np.int64)
    adj_ptr = np.zeros(deg.shape[0] + 1, np.int64)
    np.cumsum(deg, out=adj_ptr[1:])
    allf = np.concatenate(feats) if feats else np.zeros((0, 1), np.float32)
    # rows that are the reference featurizer's one-hot rows are kept as 8-byte codes (feat/atom_codes.py)
    from deepchem_amd.feat.atom_codes import codes_from_features
    codes = codes_from_features(allf) if allf.shape[0] else None
    return PackedMols(None if codes is not None else allf, atom_ptr, adj_ptr,
                      np.concatenate(idx) if idx else np.zeros(0, np.int32), codes)


def _disk_fingerprint(dataset):
    """What the packed copy of a DiskDataset depends on: the metadata rows in their current order and, per shard
    file, its size and modification time.  ``set_shard`` / ``shuffle_each_shard`` / ``sparse_shuffle`` rewrite
    files in place, ``shuffle_shards`` / ``add_shard`` / ``reshard`` change the rows; all of them change this."""
    marks = []
    for row in dataset.metadata_df.values.tolist():
        files = []
        for cell in row:
            path = os.path.join(dataset.data_dir, cell) if isinstance(cell, str) and cell.endswith(".npy") else None
            if path is not None:
                try:
                    st = os.stat(path)
                    files.append((cell, st.st_size, st.st_mtime_ns))
                except OSError:
                    files.append((cell, -1, -1))
        marks.append((tuple(str(c) for c in row), tuple(files)))
    return tuple(marks)


def packed_from_disk(dataset):
    """A DiskDataset of ConvMol objects as ONE PackedMols (+ labels, weights and the first molecule
    index of every shard), converted shard by shard once and cached on the dataset object under a
    fingerprint of the shard files (an in-place mutation of the dataset invalidates the copy, and with it
    the resident set and labels in HBM that hang off it).  Returns None when the samples are not
    ConvMol-like."""
    from deepchem_amd.utils.synthetic import concat_packed
    mark = _disk_fingerprint(dataset)
    cached = dataset.__dict__.get("_gcmi_packed")
    if cached is not None and cached[4] == mark:
        return cached[:4]
    parts, ys, ws, lens = [], [], [], []
    for X, y, w, _ in dataset.itershards():
        if len(X) and not (getattr(X, "dtype", None) == object and hasattr(X[0], "get_atom_features")):
            return None
        lens.append(len(X))
        if len(X):
            parts.append(packed_from_convmols(X))
            ys.append(y)
            ws.append(w)
    if not parts:
        return None
    packed = concat_packed(parts)
    y = None if ys[0] is None else np.concatenate(ys, axis=0)
    w = None if ws[0] is None else np.concatenate(ws, axis=0)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dataset.__dict__["_gcmi_packed"] = (packed, y, w, offsets, mark)
    return packed, y, w, offsets


def disk_index_batches(dataset, shard_offsets, batch_size, epochs, deterministic, pad_batches):
    """``DiskDataset.batch_plan`` as (molecule indices into the packed set, real molecules) per
    batch, one ``iterbatches`` pass per epoch like ``default_generator`` (graphconvmodel.py:395-398);
    padding tiles the indices (pad_batch, data/datasets.py:142-218)."""
    for _ in range(epochs):
        for shard_of_row, row_in_shard, bs in dataset.batch_plan(None, batch_size, 1, deterministic):
            idx = shard_offsets[shard_of_row] + row_in_shard
            n_real = idx.shape[0]
            if pad_batches and n_real < bs:
                idx = idx[np.arange(bs) % n_real]
            yield idx, n_real


def resident_labels(packed, device, arr, fn, tag):
    """float32 copy of the whole label (or weight) array on the device, after ``fn`` (the one-hot transform).
    Successive fit() calls over the same set find it again: the copy is kept on the molecule set under a
    fingerprint of the host array's bytes (a changed label array is converted and uploaded afresh)."""
    a = np.ascontiguousarray(np.asarray(arr))
    cache = key = mark = None
    if tag is not None and a.dtype != object:
        try:
            import xxhash
            digest = xxhash.xxh3_128_hexdigest(a.data)
        except ImportError:
            import hashlib
            digest = hashlib.blake2b(a.data, digest_size=16).hexdigest()
        cache = packed.__dict__.setdefault("_label_cache", {})
        key = (str(torch.device(device)), tag)
        mark = (digest, a.shape, a.dtype.str)
        hit = cache.get(key)
        if hit is not None and hit[0] == mark:
            return hit[1]
    conv = a if fn is None else fn(a)
    t = torch.as_tensor(np.ascontiguousarray(conv, np.float32)).to(device)
    if cache is not None:
        cache[key] = (mark, t)
    return t


class DeviceBatchPipeline:
    """Iterate ``(DeviceBatch, labels, weights)`` with collation + H2D running ``depth`` batches
    ahead on a worker thread and its own stream."""

    # labels and weights of the whole set live in HBM when they fit in this many bytes (float32, after the label
    # transform): a batch then costs one index copy and two device gathers instead of host gathers, dtype
    # conversions and two pageable copies (15 ms per 65 536 molecules x 12 tasks -- three times the collation)
    RESIDENT_LABEL_BYTES = 8 << 30
    # the molecule set itself lives in HBM below this size and batches are collated by the GPU (data/resident.py);
    # above it, or for a set that lists bonds from one end only, batches are collated on the host as before
    RESIDENT_SET_BYTES = 64 << 30
    RESIDENT_MIN_BATCH = 512

    def __init__(self, packed: PackedMols, y, w, index_batches, device: torch.device, label_fn=None,
                 depth: int = 2, workers: int = 2, resident: Optional[bool] = None, label_key=None):
        self.workers = int(os.environ.get("GCMI_PIPELINE_WORKERS", workers))
        # None = decide per batch (device collation from RESIDENT_MIN_BATCH molecules up), True = always, False = never
        self._resident_mode = resident
        if resident is None and (torch.device(device).type != "cuda" or packed.n_mols == 0
                                 or os.environ.get("GCMI_RESIDENT_SET", "1") == "0"):
            self._resident_mode = False
        self._resident_lock = threading.Lock()
        self._resident_set = None
        self.packed, self.y, self.w = packed, y, w
        self.index_batches = index_batches
        self.device = device
        if resident:
            self._resident_for(1 << 30)
        self.label_fn = label_fn
        self.depth = max(1, depth)
        self.y_dev = self.w_dev = None
        n = packed.n_mols
        if y is not None and len(y) == n and n > 0:
            per_row = int(np.prod(np.shape(y)[1:])) * (2 if label_fn is not None else 1) * 4
            if per_row * n <= self.RESIDENT_LABEL_BYTES:
                self.y_dev = self._labels_in_hbm(y, label_fn, ("y", label_key) if (label_fn is None or label_key) else None)
        if w is not None and len(w) == n and n > 0 and int(np.prod(np.shape(w)[1:])) * 4 * n <= self.RESIDENT_LABEL_BYTES:
            self.w_dev = self._labels_in_hbm(w, None, ("w", None))

    def _labels_in_hbm(self, arr, fn, tag):
        return resident_labels(self.packed, self.device, arr, fn, tag)

    @property
    def resident(self):
        """The resident set batches are collated from, once one has been built (None before / without)."""
        return self._resident_set

    def _resident_for(self, n_sel: int):
        """The set in HBM when this batch should be collated by the GPU, else None.  Small batches stay on the host:
        below a few hundred molecules one H2D copy of a tiny arena beats a plan upload plus three launches."""
        if self._resident_mode is False or (self._resident_mode is None and n_sel < self.RESIDENT_MIN_BATCH):
            return None
        if self._resident_set is not None:
            return self._resident_set
        with self._resident_lock:
            if self._resident_set is None and self._resident_mode is not False:
                from deepchem_amd.data.resident import ResidentMolSet
                cache = self.packed.__dict__.setdefault("_resident_sets", {})
                key = str(torch.device(self.device))
                forced = self._resident_mode is True
                if key not in cache or (forced and cache[key] is None):
                    cache[key] = None
                    if forced or ResidentMolSet.bytes_needed(self.packed) <= self.RESIDENT_SET_BYTES:
                        try:
                            cache[key] = ResidentMolSet(self.packed, self.device)
                        except ValueError:  # bonds listed from one end only
                            if forced:
                                raise
                if cache[key] is None:
                    self._resident_mode = False
                self._resident_set = cache[key]
        return self._resident_set

    def _make(self, idx, n_real, stream, ring):
        with torch.cuda.stream(stream):
            rset = self._resident_for(int(idx.shape[0]))
            if rset is not None:
                batch = rset.collate(idx, n_samples=idx.shape[0], ring=ring)
            else:
                batch = collate_to_device(self.packed, idx, self.device, n_samples=idx.shape[0], ring=ring)
            idx_t = None
            if self.y_dev is not None or self.w_dev is not None:
                idx_t = torch.from_numpy(np.ascontiguousarray(idx, np.int64)).to(self.device)
            if self.y_dev is not None:
                y_t = self.y_dev[idx_t]
            else:
                y_b = None if self.y is None else self.y[idx]
                if y_b is not None and self.label_fn is not None:
                    y_b = self.label_fn(y_b)
                y_t = None if y_b is None else torch.as_tensor(np.ascontiguousarray(y_b, np.float32)).to(self.device)
            if self.w_dev is not None:
                w_t = self.w_dev[idx_t]
                if n_real < idx.shape[0]:
                    w_t[n_real:] = 0
            else:
                w_b = None
                if self.w is not None:
                    w_b = self.w[idx].copy()
                    w_b[n_real:] = 0
                w_t = None if w_b is None else torch.as_tensor(np.ascontiguousarray(w_b, np.float32)).to(self.device)
            ev = torch.cuda.Event()
            ev.record(stream)
        return batch, y_t, w_t, ev

    def __iter__(self):
        """Batches in the order of ``index_batches``.  ``workers`` threads (own stream, own pinned ring) collate
        ahead; a thread takes the next index batch under a lock, so results are handed out by sequence number."""
        dev = self.device
        n_workers = max(1, self.workers)
        window = self.depth + n_workers - 1  # batches taken but not yet consumed
        cond = threading.Condition()
        src_lock = threading.Lock()
        source = iter(self.index_batches)
        state = {"taken": 0, "consumed": 0, "exhausted": False, "alive": n_workers, "error": None}
        results = {}
        stop = threading.Event()

        def take():
            with src_lock:
                if state["exhausted"]:
                    return None
                try:
                    idx, n_real = next(source)
                except StopIteration:
                    state["exhausted"] = True
                    return None
                except BaseException as e:  # the batch source failed: every batch before this one is still delivered
                    state["exhausted"] = True
                    with cond:
                        if state["error"] is None:
                            state["error"] = (state["taken"], e)
                        cond.notify_all()
                    return None
                seq = state["taken"]
                state["taken"] += 1
                return seq, idx, n_real

        def work():
            seq = None
            try:
                torch.cuda.set_device(dev)
                stream = torch.cuda.Stream(device=dev)
                ring = PinnedRing(max(2, self.depth + 3 - n_workers))  # a worker holds at most this many arenas
                while not stop.is_set():
                    with cond:
                        while not stop.is_set() and state["taken"] - state["consumed"] >= window:
                            cond.wait(0.05)
                    if stop.is_set():
                        break
                    seq = None
                    job = take()
                    if job is None:
                        break
                    seq, idx, n_real = job
                    item = self._make(idx, n_real, stream, ring)
                    with cond:
                        results[seq] = item
                        cond.notify_all()
            except BaseException as e:  # surface worker errors in the consumer, at the batch they belong to
                with cond:
                    at = seq if seq is not None else state["taken"]
                    if state["error"] is None or at < state["error"][0]:
                        state["error"] = (at, e)
                    cond.notify_all()
            finally:
                with cond:
                    state["alive"] -= 1
                    cond.notify_all()

        threads = [threading.Thread(target=work, daemon=True) for _ in range(n_workers)]
        for t in threads:
            t.start()
        held = collections.deque()  # (event on the consumer's stream, tensors of a batch it has been given)
        try:
            seq = 0
            while True:
                with cond:
                    def failed_here():
                        return state["error"] is not None and state["error"][0] <= seq
                    while seq not in results and not failed_here() and state["alive"] > 0:
                        cond.wait(0.05)
                    if seq not in results:
                        if state["error"] is not None:
                            raise state["error"][1]  # batches before the failing one have all been handed out
                        break  # every worker has finished and this batch was never produced: end of the stream
                    batch, y_t, w_t, ev = results.pop(seq)
                    state["consumed"] += 1
                    cond.notify_all()
                seq += 1
                torch.cuda.current_stream(dev).wait_event(ev)
                # Everything the worker's stream allocated and the consumer's stream reads -- the arena, the plan words,
                # the feature rows expanded from atom codes (their own allocation), the gathered labels / weights -- must
                # not go back to the allocator (which would hand it to the worker's next batch) while the step still
                # reads it.  Tensor.record_stream does that at 25 us per tensor (100 us per batch: a quarter of a
                # 100-molecule step); instead the tensors are held here until an event recorded on the consumer's
                # stream AFTER it has enqueued its work on the batch has completed.
                keep = (batch.graph._arena, getattr(batch.graph, "_plan", None), batch.atom_features, y_t, w_t)
                yield batch, y_t, w_t
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(dev))  # the consumer is back: its kernels on the batch are queued
                held.append((done, keep))
                batch = y_t = w_t = keep = None
                while held and held[0][0].query():
                    held.popleft()
        finally:
            stop.set()
            with cond:
                cond.notify_all()
            for t in threads:
                t.join(timeout=5.0)
            # the batch handed out last (and any still held) may be in use by queued kernels: wait for them before the
            # references go -- once per iteration of a whole epoch set, where fit()/predict() synchronise anyway
            try:
                torch.cuda.current_stream(dev).synchronize()
            except Exception:  # a failed synchronize means a kernel on a held batch faulted: say so, then let go
                logger.exception("DeviceBatchPipeline: synchronizing the consumer stream failed while releasing batches")
            held.clear()


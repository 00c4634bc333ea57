"""Generate tests/golden/data_*.npz and tests/golden/disk_ref/ by running THE REFERENCE's dataset
code in the build container (same arrangement as oracle/gen_golden.py: run here once, commit the
outputs; /root/reference does not travel).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_data.py

Covers SURVEY.md 8f-1: ``NumpyDataset.iterbatches`` (data/datasets.py:843-898),
``DiskDataset._iterbatches_from_shards`` (:1651-1766) incl. carry-over, padding, empty shards and the
rank-sharded walk of ``_TorchDiskDataset`` (data/pytorch_datasets.py:95-121), ``pad_batch``
(:142-218), ``select`` / ``reshard`` / the shuffles, ``get_statistics`` (:440-492),
``_convert_df_to_numpy`` and ``CSVLoader`` (data/data_loader.py:35-69, :281-437),
``NormalizationTransformer`` / ``BalancingTransformer`` / ``MinMaxTransformer`` / ``LogTransformer``
(trans/transformers.py).  Inputs are seeded NumPy arrays; outputs are plain numeric arrays.
It also checks, here, that the reference opens and iterates a directory written by
``deepchem_amd.data.DiskDataset`` (the formats are interchangeable).
"""
import os
import random
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import OUT, import_reference  # noqa: E402

SHARD_SETS = {"even": [10, 10, 10, 10, 10, 7], "ragged": [7, 3, 12, 1, 9, 25], "one": [57]}


def make_arrays(n=57, seed=7):
    rng = np.random.RandomState(seed)
    X = rng.standard_normal((n, 3))
    y = (rng.rand(n, 2) < 0.3).astype(np.float64)
    w = (rng.rand(n, 2) < 0.85).astype(np.float64)
    ids = np.arange(n)
    return X, y, w, ids


def shard_tuples(sizes, X, y, w, ids):
    out, a = [], 0
    for s in sizes:
        out.append((X[a:a + s], y[a:a + s], w[a:a + s], ids[a:a + s]))
        a += s
    return out


def trace(it):
    ids, lens, wsum, ysum = [], [], [], []
    for X_b, y_b, w_b, ids_b in it:
        ids.extend(int(i) for i in ids_b)
        lens.append(len(ids_b))
        wsum.append(float(np.sum(w_b)))
        ysum.append(float(np.sum(y_b)))
    return np.array(ids, np.int64), np.array(lens, np.int64), np.array(wsum), np.array(ysum)


def main():
    dc = import_reference()
    import pandas as pd
    from deepchem.data.data_loader import _convert_df_to_numpy
    out = {}
    X, y, w, ids = make_arrays()
    out["X"], out["y"], out["w"] = X.copy(), y.copy(), w.copy()

    # ------------------------------------------------ iteration traces
    cases = []
    for sname, sizes in SHARD_SETS.items():
        for bs in (8, 4, None):
            for det in (True, False):
                for pad in (True, False):
                    cases.append((sname, bs, det, pad))
    tmp = tempfile.mkdtemp()
    for k, (sname, bs, det, pad) in enumerate(cases):
        d = os.path.join(tmp, "ds%d" % k)
        ds = dc.data.DiskDataset.create_dataset(shard_tuples(SHARD_SETS[sname], X, y, w, ids), data_dir=d)
        np.random.seed(100 + k)
        t = trace(ds.iterbatches(batch_size=bs, epochs=2, deterministic=det, pad_batches=pad))
        for name, arr in zip(("ids", "lens", "wsum", "ysum"), t):
            out["iter_%s_%s_%d_%d_%s" % (sname, bs, det, pad, name)] = arr
        out["iter_%s_%s_%d_%d_seed" % (sname, bs, det, pad)] = np.array(100 + k)
    # NumpyDataset
    nds = dc.data.NumpyDataset(X, y, w, ids)
    for bs in (8, None):
        for det in (True, False):
            for pad in (True, False):
                np.random.seed(7)
                t = trace(nds.iterbatches(batch_size=bs, epochs=2, deterministic=det, pad_batches=pad))
                for name, arr in zip(("ids", "lens", "wsum", "ysum"), t):
                    out["np_%s_%d_%d_%s" % (bs, det, pad, name)] = arr
    # rank-sharded walk (a subset of the shards; the batch count is still taken over the whole set)
    ds = dc.data.DiskDataset.create_dataset(shard_tuples(SHARD_SETS["ragged"], X, y, w, ids),
                                            data_dir=os.path.join(tmp, "ranked"))
    for rank, world in ((0, 2), (1, 2), (2, 3)):
        n_shards = ds.get_number_shards()
        first, last = (rank * n_shards) // world, ((rank + 1) * n_shards) // world
        np.random.seed(55)
        t = trace(ds._iterbatches_from_shards(list(range(first, last)), batch_size=4, epochs=1, deterministic=False))
        for name, arr in zip(("ids", "lens", "wsum", "ysum"), t):
            out["rank_%d_%d_%s" % (rank, world, name)] = arr

    # ------------------------------------------------ a reference-written directory, committed as data
    ref_dir = os.path.join(OUT, "disk_ref")
    if os.path.isdir(ref_dir):
        shutil.rmtree(ref_dir)
    dc.data.DiskDataset.create_dataset(shard_tuples(SHARD_SETS["ragged"], X.astype(np.float32), y, w, ids),
                                       data_dir=ref_dir, tasks=["t0", "t1"])

    # ------------------------------------------------ derived datasets
    ds = dc.data.DiskDataset.create_dataset(shard_tuples(SHARD_SETS["ragged"], X, y, w, ids),
                                            data_dir=os.path.join(tmp, "derived"))
    out["shape_X"], out["shape_y"], out["shape_w"], out["shape_ids"] = [np.array(s) for s in ds.get_shape()]
    sel = [5, 50, 3, 22, 41, 0, 56, 13, 14]
    s = ds.select(sel, select_shard_size=4)
    out["select_idx"] = np.array(sel)
    out["select_X"], out["select_ids"] = s.X, s.ids.astype(np.int64)
    out["select_n_shards"] = np.array(s.get_number_shards())
    np.random.seed(9)
    cs = ds.complete_shuffle()
    out["complete_shuffle_ids"] = cs.ids.astype(np.int64)
    out["complete_shuffle_n_shards"] = np.array(cs.get_number_shards())
    ds2 = ds.copy(os.path.join(tmp, "copy1"))
    ds2.reshard(10)
    out["reshard_lens"] = np.array([len(ds2.get_shard_ids(i)) for i in range(ds2.get_number_shards())])
    out["reshard_ids"] = ds2.ids.astype(np.int64)
    np.random.seed(10)
    ds2.shuffle_each_shard()
    out["shuffle_each_shard_ids"] = ds2.ids.astype(np.int64)
    random.seed(11)
    ds2.shuffle_shards()
    out["shuffle_shards_ids"] = ds2.ids.astype(np.int64)
    np.random.seed(12)
    ds2.sparse_shuffle()
    out["sparse_shuffle_ids"] = ds2.ids.astype(np.int64)
    out["sparse_shuffle_X"] = ds2.X
    sub = ds.subset([1, 2, 4])
    out["subset_ids"] = sub.ids.astype(np.int64)
    m = dc.data.DiskDataset.merge([sub, ds.subset([0])])
    out["merge_ids"] = m.ids.astype(np.int64)
    stats = ds.get_statistics()
    for name, arr in zip(("X_means", "X_stds", "y_means", "y_stds"), stats):
        out["stats_" + name] = arr

    # ------------------------------------------------ transformers
    yr = np.random.RandomState(3).standard_normal((57, 2)) * np.array([3.0, 0.5]) + np.array([1.0, -2.0])
    yr[:, 1] = np.round(yr[:, 1], 1)
    rds = dc.data.NumpyDataset(X, yr, w, ids)
    out["reg_y"] = yr
    t = dc.trans.NormalizationTransformer(transform_y=True, dataset=rds)
    out["norm_y"] = t.transform(rds).y
    out["norm_y_means"], out["norm_y_stds"] = t.y_means, t.y_stds
    out["norm_y_undo"] = dc.trans.undo_transforms(out["norm_y"][:, :, None], [t])
    t = dc.trans.NormalizationTransformer(transform_X=True, dataset=rds)
    out["norm_X"] = t.transform(rds).X
    t = dc.trans.NormalizationTransformer(transform_y=True, dataset=rds, move_mean=False)
    out["norm_y_nomove"] = t.transform(rds).y
    dsr = dc.data.DiskDataset.create_dataset(shard_tuples(SHARD_SETS["ragged"], X, yr, w, ids),
                                             data_dir=os.path.join(tmp, "reg"))
    t = dc.trans.NormalizationTransformer(transform_y=True, dataset=dsr)
    td = t.transform(dsr)
    out["norm_y_disk"] = td.y
    out["norm_y_disk_n_shards"] = np.array(td.get_number_shards())
    cds = dc.data.NumpyDataset(X, y, w, ids)
    t = dc.trans.BalancingTransformer(dataset=cds)
    out["bal_w"] = t.transform(cds).w
    out["bal_weights"] = np.array(t.weights, np.float64)
    y1 = y[:, 0].copy()
    w1 = w[:, 0].copy()
    c1 = dc.data.NumpyDataset(X, y1, w1, ids)
    t = dc.trans.BalancingTransformer(dataset=c1)
    out["bal1_w"] = t.transform(c1).w
    t = dc.trans.MinMaxTransformer(transform_y=True, dataset=rds)
    out["minmax_y"] = t.transform(rds).y
    out["minmax_y_undo"] = t.untransform(out["minmax_y"])
    pos = np.abs(X)
    pds = dc.data.NumpyDataset(pos, yr, w, ids)
    t = dc.trans.LogTransformer(transform_X=True, dataset=pds)
    out["log_X"] = t.transform(pds).X
    t = dc.trans.LogTransformer(transform_X=True, features=[0, 2], dataset=pds)
    out["log_X_cols"] = t.transform(pds).X
    t = dc.trans.ClippingTransformer(transform_X=True, x_max=0.5)
    out["clip_X"] = t.transform(dc.data.NumpyDataset(X.copy(), yr, w, ids)).X  # the reference clips in place

    # ------------------------------------------------ CSV -> labels / weights, CSVLoader
    csv = os.path.join(OUT, "data_toy.csv")
    with open(csv, "w") as f:
        f.write("smiles,t0,t1,name\n")
        rows = [("CCO", "1", "0", "a"), ("CXC", "0", "1", "b"), ("CCCC", "", "1", "c"), ("C", "0", "", "d"),
                ("CCN", "1", "1", "e"), ("OXO", "", "", "f"), ("CCCCCC", "0", "0", "g"), ("N", "1", "", "h"),
                ("CC", "0", "1", "i")]
        for r in rows:
            f.write(",".join(r) + "\n")
    df = next(iter(dc.utils.data_utils.load_csv_files([csv], shard_size=100)))
    yy, ww = _convert_df_to_numpy(df, ["t0", "t1"])
    out["csv_y"], out["csv_w"] = yy, ww

    class Toy(dc.feat.Featurizer):
        """[length, carbons]; an input containing 'X' fails (empty feature array).  ``featurize``
        is overridden only to build the ragged result as an object array: the reference's own
        ``np.asarray(features)`` (feat/base_classes.py:58) needs its pinned numpy<2 for that."""

        def _featurize(self, s, **kwargs):
            if "X" in s:
                return np.array([])
            return np.array([len(s), s.count("C")], np.float64)

        def featurize(self, datapoints, **kwargs):
            feats = [self._featurize(d) for d in datapoints]
            res = np.empty(len(feats), dtype=object)
            for i, f in enumerate(feats):
                res[i] = f
            return res

    loader = dc.data.CSVLoader(["t0", "t1"], featurizer=Toy(), feature_field="smiles", id_field="name")
    lds = loader.create_dataset(csv, shard_size=4)
    out["loader_X"], out["loader_y"], out["loader_w"] = lds.X, lds.y, lds.w
    out["loader_ids"] = np.array([str(s) for s in lds.ids])
    out["loader_shard_lens"] = np.array([len(lds.get_shard_ids(i)) for i in range(lds.get_number_shards())])

    # ------------------------------------------------ format interchange: ours -> reference
    import deepchem_amd as dca
    ours = dca.data.DiskDataset.create_dataset(shard_tuples(SHARD_SETS["ragged"], X, y, w, ids),
                                               data_dir=os.path.join(tmp, "ours"), tasks=["t0", "t1"])
    ref_open = dc.data.DiskDataset(ours.data_dir)
    assert ref_open.get_shape() == ds.get_shape(), (ref_open.get_shape(), ds.get_shape())
    assert np.array_equal(ref_open.X, X) and np.array_equal(ref_open.ids.astype(np.int64), ids)
    np.random.seed(1)
    a = trace(ref_open.iterbatches(batch_size=8, epochs=1, pad_batches=True))
    np.random.seed(1)
    b = trace(ds.iterbatches(batch_size=8, epochs=1, pad_batches=True))
    assert all(np.array_equal(p, q) for p, q in zip(a, b))
    out["ours_opened_by_reference"] = np.array(1)

    np.savez_compressed(os.path.join(OUT, "data_iter.npz"), **out)
    shutil.rmtree(tmp)
    print("wrote data_iter.npz with", len(out), "arrays;", ref_dir)


if __name__ == "__main__":
    main()

"""The two metric helpers the GraphConv path touches."""
import numpy as np


def to_one_hot(y: np.ndarray, n_classes: int = 2) -> np.ndarray:
    """(N,) or (N,1) labels -> (N, n_classes) one-hot float64
    (deepchem/metrics/metric.py:371-400, same errors)."""
    if len(y.shape) > 2:
        raise ValueError("y must be a vector of shape (N,) or (N, 1)")
    if len(y.shape) == 2 and y.shape[1] != 1:
        raise ValueError("y must be a vector of shape (N,) or (N, 1)")
    if len(np.unique(y)) > n_classes:
        raise ValueError("y has more than n_class unique elements.")
    n = np.shape(y)[0]
    y_hot = np.zeros((n, n_classes))
    y_hot[np.arange(n), np.asarray(y).reshape(-1).astype(np.int64)] = 1
    return y_hot


def roc_auc_per_task(y_true: np.ndarray, y_prob: np.ndarray, w: np.ndarray = None):
    """Per-task ROC-AUC as deepchem.metrics.Metric(roc_auc_score) computes it
    (metrics/metric.py:568-665: samples with zero weight are dropped; sklearn's
    roc_auc_score on the class-1 probability).  y_prob: (N, T, 2) or (N, T)."""
    from sklearn.metrics import roc_auc_score
    y_true = np.asarray(y_true)
    if y_prob.ndim == 3:
        y_prob = y_prob[:, :, 1]
    out = []
    for t in range(y_true.shape[1]):
        keep = np.ones(y_true.shape[0], bool) if w is None else (w[:, t] != 0)
        yt, yp = y_true[keep, t], y_prob[keep, t]
        out.append(float("nan") if len(np.unique(yt)) < 2 else roc_auc_score(yt, yp))
    return np.array(out)

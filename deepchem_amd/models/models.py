"""Abstract ``Model`` (deepchem/models/models.py:22-235): it owns the model directory, and ``evaluate`` scores
predictions with metric callables."""
import os
import shutil
import tempfile
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


class Model(object):

    def __init__(self, model=None, model_dir: Optional[str] = None, **kwargs) -> None:
        if type(self).__name__ == "Model":
            raise ValueError("This constructor is for an abstract class and should never be called directly.")
        # a directory of our own making is removed again with the object; a caller's directory is kept
        self.model_dir_is_temp = model_dir is None
        if model_dir is None:
            model_dir = tempfile.mkdtemp()
        os.makedirs(model_dir, exist_ok=True)
        self.model_dir = model_dir
        self.model = model
        self.model_class = type(model)

    def __del__(self):
        if getattr(self, "model_dir_is_temp", False):
            shutil.rmtree(self.model_dir, ignore_errors=True)

    def get_task_type(self) -> str:
        raise NotImplementedError

    def get_num_tasks(self) -> int:
        raise NotImplementedError

    def evaluate(self, dataset, metrics: Sequence[Callable], transformers: List = [],
                 per_task_metrics: bool = False) -> Dict[str, float]:
        """Score ``predict(dataset)`` against ``dataset.y``.  ``metrics``: callables
        ``f(y_true, y_pred, w) -> float or per-task array`` (``dc.metrics.Metric`` objects are such callables
        where DeepChem is installed; models/models.py:162-223).  Labels and predictions both go back through
        the y-transformers before scoring, as in the reference's Evaluator (utils/evaluate.py:197-307).
        Returns ``{metric name: score}``; the mean over tasks unless ``per_task_metrics``."""
        from deepchem_amd.trans.transformers import undo_transforms
        on_labels = [t for t in transformers if t.transform_y]
        truth = undo_transforms(dataset.y, on_labels)
        predicted = self.predict(dataset, on_labels)
        scores = {}
        for metric in metrics:
            value = metric(truth, predicted, dataset.w)
            label = getattr(metric, "name", None) or getattr(metric, "__name__", "metric")
            scores[label] = value if per_task_metrics else float(np.nanmean(value))
        return scores

"""Abstract ``Model`` (deepchem/models/models.py:22-235): owns the model
directory; ``evaluate`` scores predictions with plain metric callables."""
import os
import shutil
import tempfile
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


class Model(object):

    def __init__(self, model=None, model_dir: Optional[str] = None, **kwargs) -> None:
        if self.__class__.__name__ == "Model":
            raise ValueError("This constructor is for an abstract class and should never be called directly.")
        self.model_dir_is_temp = False
        if model_dir is not None:
            if not os.path.exists(model_dir):
                os.makedirs(model_dir)
        else:
            model_dir = tempfile.mkdtemp()
            self.model_dir_is_temp = True
        self.model_dir = model_dir
        self.model = model
        self.model_class = model.__class__

    def __del__(self):
        if getattr(self, "model_dir_is_temp", False):
            shutil.rmtree(self.model_dir, ignore_errors=True)

    def get_task_type(self) -> str:
        raise NotImplementedError

    def get_num_tasks(self) -> int:
        raise NotImplementedError

    def evaluate(self, dataset, metrics: Sequence[Callable], transformers: List = [],
                 per_task_metrics: bool = False) -> Dict[str, float]:
        """``metrics``: callables ``f(y_true, y_pred, w) -> float or per-task array``
        (a stand-in for dc.metrics.Metric, which is used as-is when DeepChem itself is
        installed; models/models.py:162-223).  As the reference's Evaluator does
        (utils/evaluate.py:197-307), labels and predictions are both taken back through the
        y-transformers before scoring.  Returns {name: score}."""
        from deepchem_amd.trans.transformers import undo_transforms
        output_transformers = [t for t in transformers if t.transform_y]
        y_true = undo_transforms(dataset.y, output_transformers)
        y_pred = self.predict(dataset, output_transformers)
        out = {}
        for m in metrics:
            name = getattr(m, "name", getattr(m, "__name__", "metric"))
            score = m(y_true, y_pred, dataset.w)
            out[name] = score if per_task_metrics else float(np.nanmean(score))
        return out

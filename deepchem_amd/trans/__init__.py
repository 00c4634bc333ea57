"""Dataset transformers (deepchem/trans/transformers.py)."""
from deepchem_amd.trans.transformers import (BalancingTransformer, ClippingTransformer, LogTransformer,  # noqa: F401
                                             MinMaxTransformer, NormalizationTransformer, Transformer,
                                             undo_transforms)

"""deepchem_amd: the DeepChem GraphConv training path on MI355X (gfx950).

One hot path -- GraphConv -> GraphPool -> GraphGather -> dense heads, forward and
backward -- as hand-written HIP kernels behind a C ABI (include/gcmi.h), presented
through the reference's own class contract:

    import deepchem_amd as dc
    model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64])
    model.fit(dataset, nb_epoch=10); model.predict(dataset)

Everything else of DeepChem is out of scope.
"""
__version__ = "0.1.0"

from deepchem_amd import data, feat, metrics, models, trans, utils  # noqa: E402,F401


def set_gemm_mode(mode: str) -> None:
    """``"exact"``: every matrix product on the exact-fp32 matrix-core chain (the reference's
    summation order; training trajectories track the CPU reference to ~1e-5).  ``"fast"`` (default):
    split-bf16 products, fp32-accurate per product and ~1.25x faster end to end."""
    from deepchem_amd import _lib
    if mode not in ("exact", "fast"):
        raise ValueError("mode must be 'exact' or 'fast'")
    _lib.call("gcmi_set_option", _lib.GCMI_OPT_GEMM_EXACT, 1 if mode == "exact" else 0)

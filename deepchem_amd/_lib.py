"""ctypes binding of libgcmi.so (include/gcmi.h).

There is no fallback: if the library cannot be loaded (and cannot be built
because hipcc is absent) importing a compute entry point raises.  Nothing in
``deepchem_amd`` computes the hot path with torch ops or on the CPU.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_void_p)

from deepchem_amd import _build

GCMI_MAX_DEG = 10
BN_ACC_REPLICAS = 32


def bn_acc_doubles(n_feat: int) -> int:
    return 66 * n_feat
K_GATHER_SUM, K_GATHER_MAX, K_READOUT, K_SEG_GEMM, K_WGRAD, K_GATHER_MAX_BWD, K_BATCHNORM, K_FUSED_BWD = 0, 1, 2, 3, 4, 5, 6, 7


GCMI_OPT_GEMM_EXACT = 1
GCMI_OPT_FUSED_BN_STATS = 2
GCMI_OPT_FUSED_BWD = 3
GCMI_OPT_FUSED_BWD_LAUNCHES = 4
GCMI_OPT_READOUT_PIPELINED = 5
GCMI_WIN_META_INTS = 24
GCMI_COLLATE_WIN_DESC_INTS = 36
GCMI_WIN_MAX_SLOTS = 4095


class GcmiGraph(Structure):
    """struct gcmi_graph (include/gcmi.h)."""
    _fields_ = [
        ("n_atoms", c_int32),
        ("n_edges", c_int32),
        ("n_mols", c_int32),
        ("max_deg", c_int32),
        ("deg_start", c_int32 * (GCMI_MAX_DEG + 2)),
        ("edge_start", c_int32 * (GCMI_MAX_DEG + 2)),
        ("d_col_idx", c_void_p),
        ("d_membership", c_void_p),
        ("d_mol_runs", c_void_p),
        ("d_rev_pos", c_void_p),
        ("n_win", c_int32),
        ("n_win_big", c_int32),
        ("win_alloc", c_int32),
        ("win_ecap", c_int32),
        ("win_alloc_big", c_int32),
        ("win_ecap_big", c_int32),
        ("win_reserved", c_int32 * 2),
        ("d_win_meta", c_void_p),
        ("d_win_edges", c_void_p),
    ]


MAX_CONV_LAYERS = 4


class GcmiModelDesc(Structure):
    """struct gcmi_model_desc (include/gcmi.h)."""
    _fields_ = [
        ("n_layers", c_int32),
        ("max_deg", c_int32),
        ("n_feat_in", c_int32),
        ("conv_width", c_int32 * MAX_CONV_LAYERS),
        ("dense_width", c_int32),
        ("n_tasks", c_int32),
        ("n_classes", c_int32),
        ("mode", c_int32),
        ("batch_norm", c_int32),
        ("grad_mode", c_int32),
        ("bn_eps", c_float),
        ("bn_momentum", c_float),
        ("off_conv_w", c_int64 * MAX_CONV_LAYERS),
        ("off_conv_b", c_int64 * MAX_CONV_LAYERS),
        ("off_bn_gamma", c_int64 * (MAX_CONV_LAYERS + 1)),
        ("off_bn_beta", c_int64 * (MAX_CONV_LAYERS + 1)),
        ("off_dense_w", c_int64),
        ("off_dense_b", c_int64),
        ("off_head_w", c_int64),
        ("off_head_b", c_int64),
        ("n_params", c_int64),
        ("storage", c_int32),
        ("reserved_", c_int32),
    ]


class GcmiModelIO(Structure):
    """struct gcmi_model_io (include/gcmi.h)."""
    _fields_ = [
        ("d_atom_features", c_void_p),
        ("ld_features", c_int64),
        ("d_workspace", c_void_p),
        ("d_bn_running_mean", c_void_p * (MAX_CONV_LAYERS + 1)),
        ("d_bn_running_var", c_void_p * (MAX_CONV_LAYERS + 1)),
        ("d_bn_batches_tracked", c_void_p * (MAX_CONV_LAYERS + 1)),
        ("d_logits", c_void_p),
        ("d_probs", c_void_p),
        ("d_fingerprint", c_void_p),
        ("d_loss", c_void_p),
    ]


class GcmiSmallBatch(Structure):
    """struct gcmi_small_batch (include/gcmi.h)."""
    _fields_ = [
        ("graph", GcmiGraph),
        ("d_atom_features", c_void_p),
        ("ld_features", c_int64),
        ("d_labels", c_void_p),
        ("d_weights", c_void_p),
        ("n_rows", c_int64),
        ("d_logits", c_void_p),
        ("d_probs", c_void_p),
        ("d_fingerprint", c_void_p),
    ]


_P = c_void_p
_G = POINTER(GcmiGraph)
_MD = POINTER(GcmiModelDesc)
_MIO = POINTER(GcmiModelIO)
_I32P = POINTER(c_int32)
_I64P = POINTER(c_int64)

# name -> argtypes; every function returns int (status) unless noted
_SIGNATURES = {
    "gcmi_collate_sizes": [_P, _P, _P, c_int64, _I64P, _I64P],
    "gcmi_collate": [_P, c_int64, _P, _P, _P, _P, c_int64, c_int32, _P, c_int64, c_int64, _P, _P,
                     c_int64, _P, _G],
    "gcmi_collate_plans": [_P, c_int64, _P, _P, _P, _P, c_int64, c_int32, _P, c_int64, c_int64, _P, _P,
                           c_int64, _P, _P, _I32P, c_int32, _P, _P, _G],
    "gcmi_molset_tables": [_P, _P, _P, c_int64, c_int32, _P, _P, _P, _I32P, c_int32],
    "gcmi_collate_plan": [_P, _P, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, _G],
    "gcmi_collate_rows": [_P, c_int64, _P, _P, _P, _P, _P, _P, _G, _P, c_int64, _P, _P, _P, _P, _P, _P, _P],
    "gcmi_collate_rows_host": [_P, c_int64, _P, _P, _P, _P, _P, _P, _G, _P, c_int64, _P, _P, _P, _P, _P],
    "gcmi_build_mol_runs": [_G, _P, _P, _P],
    "gcmi_build_rev_pos": [_G, _P, _P, _P],
    "gcmi_gather_sum_fwd": [_G, _P, c_int64, c_int32, _P, c_int64, c_int32, _P],
    "gcmi_scatter_add": [_G, _P, c_int64, c_int32, _P, c_int64, _P],
    "gcmi_gather_max_fwd": [_G, _P, c_int64, c_int32, _P, _P, _P, c_int64, _P, _P],
    "gcmi_gather_max_bwd": [_G, _P, c_int64, c_int32, _P, _P, c_int64, _P],
    "gcmi_readout_fwd": [_G, _P, c_int64, c_int32, _P, _P, c_int32, _P, c_int64, _P, _P],
    "gcmi_readout_bwd": [_G, _P, c_int64, _P, c_int64, c_int32, c_int32, _P, _P, c_int64, _P],
    "gcmi_bn_stats": [_P, c_int64, c_int64, c_int32, _P, _P, c_float, c_float, _P, _P, _P, _P, _P,
                      _P, _P, _P],
    "gcmi_bn_fold_eval": [_P, _P, _P, _P, c_float, c_int32, _P, _P, _P],
    "gcmi_bn_apply": [_P, c_int64, c_int64, c_int32, _P, _P, _P, c_int64, _P],
    "gcmi_bn_bwd": [_P, c_int64, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, _P, c_int64,
                    c_int32, _P, _P],
    "gcmi_seg_gemm": [c_int32, _I32P, _I32P, _P, c_int64, c_int32, _P, _I64P, _P, c_int64, c_int32,
                      _P, _I64P, _P, _I64P, c_int32, c_int32, c_int32, _P, c_int64, _P],
    "gcmi_seg_gemm_wgrad": [c_int32, _I32P, _I32P, _P, c_int64, c_int32, _P, c_int64, c_int32, _P,
                            _I64P, _P, _I64P, c_int32, _P],
    "gcmi_task_head_forward": [_P, c_int64, c_int64, c_int32, _P, _P, c_int32, _P, _P, c_int64, _P],
    "gcmi_relu_bwd": [_P, c_int64, _P, c_int64, c_int64, c_int32, _P],
    "gcmi_loss_fwd_bwd": [c_int32, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P],
    "gcmi_softmax": [_P, c_int64, c_int32, _P, _P],
    "gcmi_adam_step": [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int64, _P],
    "gcmi_fold_affine": [_P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P],
    "gcmi_weave_pair_to_atom": [_P, c_int64, c_int32, _P, c_int64, c_int32, _P, _P, c_int32, _P, c_int64, _P],
    "gcmi_weave_pair_features": [_P, _P, c_int64, c_int32, _P, _P, c_int64, c_int32, _P, _P, c_int32, _P,
                                 c_int64, _P, c_int64, _P],
    "gcmi_weave_gather": [_P, c_int64, c_int32, _P, c_int32, c_int32, _P, c_int64, _P],
    "gcmi_tanh_": [_P, c_int64, c_int64, c_int32, _P],
    "gcmi_edge_network_sum": [_P, c_int64, c_int32, c_int32, _P, c_int64, _P, _P, c_int32, _P, c_int64, _P],
    "gcmi_edge_network_moments": [_P, c_int64, c_int32, c_int32, _P, c_int64, _P, _P, c_int32, _P, c_int64, _P],
    "gcmi_edge_network_moments_mol": [_P, c_int64, c_int32, c_int32, _P, c_int64, _P, _P, c_int32, _P, c_int32, c_int32, _P,
                                      c_int64, _P],
    "gcmi_gru_gates": [_P, _P, _P, _P, c_int64, _P],
    "gcmi_gru_out": [_P, _P, _P, _P, c_int64, _P],
    "gcmi_gru_gates_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P],
    "gcmi_gru_out_bwd": [_P, _P, _P, _P, _P, _P, _P, c_int64, _P],
    "gcmi_lstm_cell_bwd": [_P, c_int64, c_int32, c_int64, _P, _P, _P, _P, _P, _P],
    "gcmi_set2set_attend_bwd": [_P, c_int64, c_int32, _P, c_int32, _P, c_int64, _P, c_int64, _P, c_int64, c_int32, _P,
                                c_int64, _P],
    "gcmi_set2set_attend": [_P, c_int64, c_int32, _P, c_int32, _P, c_int64, _P, c_int64, _P],
    "gcmi_lstm_cell": [_P, c_int64, c_int32, c_int64, _P, c_int64, _P, c_int64, _P],
    "gcmi_model_forward": [_MD, _G, _P, _MIO, c_int32, _P],
    "gcmi_model_loss_backward": [_MD, _G, _P, _P, _MIO, _P, _P, c_int64, _I64P, _I64P, _P],
    "gcmi_collate_batches": [_P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int32, c_int64, c_int64, _P, _P, _P, _P, _P,
                             c_int32],
    "gcmi_small_bind": [_P, _P, _P, _P, c_int64, _P, _P, c_int64, c_int64, _P, _P, c_int64, _P, c_int64, _P, _P,
                        c_int64, _P, c_int64],
    "gcmi_small_fit": [_MD, _P, _P, _P, _P, _MIO, _P, c_int64, c_int64, c_int64, c_float, c_float, c_float, c_float,
                       c_int64, _P, _I64P, _I64P, _P],
    "gcmi_small_fit_dp": [_MD, _P, _P, _P, _P, _MIO, _P, c_int64, c_int64, c_int64, c_float, c_float, c_float, c_float,
                          c_int64, _P, _I64P, _I64P, _P, _P, _P],
    "gcmi_small_predict": [_MD, _P, _MIO, _P, c_int64, c_int64, c_int64, _P],
    "gcmi_diag_mfma_peak": [c_int32, c_int32, _P, _P],
    "gcmi_set_option": [c_int32, c_int32],
    "gcmi_get_option": [c_int32, _I32P],
    "gcmi_expand_atom_codes": [_P, c_int64, c_int64, _P, c_int64, _P],
    "gcmi_smiles_sizes": [_P, c_int64, _I32P, _I32P, ctypes.c_int],
    "gcmi_smiles_featurize": [_P, c_int64, _I64P, _I64P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int],
    "gcmi_timing_enable": [c_int32, c_int32],
    "gcmi_timing_read": [c_int32, _I64P, POINTER(c_double), c_int32],
}

EXPORTS = ["gcmi_version", "gcmi_last_error", "gcmi_model_workspace_floats", "gcmi_small_workspace_floats",
           "gcmi_task_head_scratch_floats",
           "gcmi_smiles_check", "gcmi_collate_plan_words", "gcmi_collate_batches_layout"] + sorted(_SIGNATURES)

_lib = None


class GcmiError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB


def load():
    """dlopen libgcmi.so (building it first when it is missing or stale and
    hipcc is available).  Raises GcmiError otherwise -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    # GCMI_HOST_ONLY_LIB: a sanitizer build of the host-side sources only (tools/asan_host.sh); the kernels'
    # entry points are absent from it and anything that needs them fails with AttributeError
    host_only = os.environ.get("GCMI_HOST_ONLY_LIB")
    path = host_only or _build.LIB
    if not os.path.exists(path):
        try:
            _build.build_lib(verbose=False)
        except Exception as e:  # no hipcc on this machine
            raise GcmiError(
                "libgcmi.so is missing and could not be built (%s). Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` where hipcc exists." % e)
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise GcmiError("cannot load %s: %s" % (path, e))
    special = {
        "gcmi_version": (ctypes.c_int, None),
        "gcmi_last_error": (c_char_p, None),
        "gcmi_model_workspace_floats": (c_int64, [_MD, c_int64, c_int64]),
        "gcmi_collate_batches_layout": (c_int64, [_P, _P, _P, _P, c_int64, c_int64, c_int32, c_int64, _P, _P]),
        "gcmi_small_workspace_floats": (c_int64, [_MD, c_int64, c_int64]),
        "gcmi_smiles_check": (c_char_p, [c_char_p]),
        "gcmi_collate_plan_words": (c_int64, [c_int64]),
        "gcmi_task_head_scratch_floats": (c_int64, []),
    }
    for name, (restype, argtypes) in special.items():
        if host_only and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = restype
        if argtypes is not None:
            fn.argtypes = argtypes
    for name, argtypes in _SIGNATURES.items():
        if host_only and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gcmi_last_error()
        raise GcmiError("%s failed (status %d): %s" % (what or "gcmi call", rc,
                                                      msg.decode() if msg else "?"))


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)

"""Typed wrappers over the C ABI (include/gcmi.h) and the autograd glue.

Each wrapper checks on the HOST that shapes, dtypes, strides and devices are
what the kernel and its grid assume (a faulting kernel can take the GPU down)
and then enqueues the kernel on torch's current stream.  The
``torch.autograd.Function``s below only sequence those calls; every FLOP and
every byte of the hot path moves in libgcmi.so.
"""
import ctypes
from typing import Optional

import torch

from deepchem_amd import _lib
from deepchem_amd.graph import BatchGraph, _stream

_vp = ctypes.c_void_p


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else _vp(t.data_ptr())


def _mat(t: torch.Tensor, name: str, rows: Optional[int] = None, cols: Optional[int] = None,
         dtype=torch.float32):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise _lib.GcmiError("%s must be a CUDA tensor: the hot path has no CPU implementation" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("%s must be 2-D with unit column stride (shape %s, strides %s)" %
                         (name, tuple(t.shape), t.stride()))
    if rows is not None and t.shape[0] != rows:
        raise ValueError("%s has %d rows, expected %d" % (name, t.shape[0], rows))
    if cols is not None and t.shape[1] != cols:
        raise ValueError("%s has %d columns, expected %d" % (name, t.shape[1], cols))
    return t


def _ld(t: torch.Tensor) -> int:
    return max(int(t.stride(0)), int(t.shape[1])) if t.shape[0] > 1 else int(t.shape[1])


def _vec(t: Optional[torch.Tensor], name: str, n: int, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or t.numel() != n or not t.is_contiguous():
        raise ValueError("%s must be a contiguous %s CUDA vector of %d" % (name, dtype, n))
    return t


def rowmajor(t: torch.Tensor) -> torch.Tensor:
    """A view usable by the kernels (unit column stride), copying only if needed."""
    if t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1) and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1]):
        return t
    return t.contiguous()


# ------------------------------------------------------------------ raw kernels
def gather_sum(g: BatchGraph, x: torch.Tensor, out: Optional[torch.Tensor] = None,
               accumulate: bool = False) -> torch.Tensor:
    _mat(x, "x", rows=g.n_atoms)
    F_ = x.shape[1]
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an output")
        out = torch.empty((g.n_atoms, F_), dtype=torch.float32, device=x.device)
    _mat(out, "out", rows=g.n_atoms, cols=F_)
    _lib.call("gcmi_gather_sum_fwd", g.ref, _ptr(x), _ld(x), F_, _ptr(out), _ld(out),
              1 if accumulate else 0, _stream())
    return out


def scatter_add(g: BatchGraph, ds: torch.Tensor, dx: torch.Tensor) -> torch.Tensor:
    _mat(ds, "ds", rows=g.n_atoms)
    _mat(dx, "dx", rows=g.n_atoms, cols=ds.shape[1])
    _lib.call("gcmi_scatter_add", g.ref, _ptr(ds), _ld(ds), ds.shape[1], _ptr(dx), _ld(dx), _stream())
    return dx


def gather_max(g: BatchGraph, x: torch.Tensor, scale=None, shift=None, want_arg: bool = True):
    _mat(x, "x", rows=g.n_atoms)
    F_ = x.shape[1]
    _vec(scale, "scale", F_)
    _vec(shift, "shift", F_)
    out = torch.empty((g.n_atoms, F_), dtype=torch.float32, device=x.device)
    arg = torch.empty((g.n_atoms, F_), dtype=torch.uint8, device=x.device) if want_arg else None
    _lib.call("gcmi_gather_max_fwd", g.ref, _ptr(x), _ld(x), F_, _ptr(scale), _ptr(shift), _ptr(out),
              _ld(out), _ptr(arg), _stream())
    return out, arg


def gather_max_bwd(g: BatchGraph, dout: torch.Tensor, arg: torch.Tensor) -> torch.Tensor:
    _mat(dout, "dout", rows=g.n_atoms)
    F_ = dout.shape[1]
    if arg.dtype != torch.uint8 or tuple(arg.shape) != (g.n_atoms, F_) or not arg.is_contiguous():
        raise ValueError("bad arg tensor")
    if g.ensure_rev_pos():  # gather form: every row written once, no atomics
        dx = torch.empty((g.n_atoms, F_), dtype=torch.float32, device=dout.device)
    else:
        dx = torch.zeros((g.n_atoms, F_), dtype=torch.float32, device=dout.device)
    _lib.call("gcmi_gather_max_bwd", g.ref, _ptr(dout), _ld(dout), F_, _ptr(arg), _ptr(dx), _ld(dx),
              _stream())
    return dx


def readout(g: BatchGraph, x: torch.Tensor, n_mols: int, scale=None, shift=None, tanh: bool = False):
    _mat(x, "x", rows=g.n_atoms)
    F_ = x.shape[1]
    _vec(scale, "scale", F_)
    _vec(shift, "shift", F_)
    g.set_mols(n_mols)
    out = torch.empty((n_mols, 2 * F_), dtype=torch.float32, device=x.device)
    arg = torch.empty((n_mols, F_), dtype=torch.int32, device=x.device)
    _lib.call("gcmi_readout_fwd", g.ref, _ptr(x), _ld(x), F_, _ptr(scale), _ptr(shift),
              1 if tanh else 0, _ptr(out), _ld(out), _ptr(arg), _stream())
    return out, arg


def readout_bwd(g: BatchGraph, dout: torch.Tensor, out: torch.Tensor, arg: torch.Tensor,
                tanh: bool) -> torch.Tensor:
    F_ = arg.shape[1]
    _mat(dout, "dout", rows=g.n_mols, cols=2 * F_)
    _mat(out, "out", rows=g.n_mols, cols=2 * F_)
    if arg.dtype != torch.int32 or arg.shape[0] != g.n_mols or not arg.is_contiguous():
        raise ValueError("bad arg tensor")
    dx = torch.empty((g.n_atoms, F_), dtype=torch.float32, device=dout.device)
    _lib.call("gcmi_readout_bwd", g.ref, _ptr(dout), _ld(dout), _ptr(out), _ld(out), F_,
              1 if tanh else 0, _ptr(arg), _ptr(dx), _ld(dx), _stream())
    return dx


def bn_stats(x, gamma, beta, running_mean, running_var, eps: float, momentum: float):
    """Training statistics + folded scale/shift; updates running stats in place."""
    _mat(x, "x")
    n, F_ = x.shape
    if n < 1:
        raise ValueError("batch norm over zero rows")
    dev = x.device
    mean = torch.empty(F_, dtype=torch.float32, device=dev)
    invstd = torch.empty(F_, dtype=torch.float32, device=dev)
    scale = torch.empty(F_, dtype=torch.float32, device=dev)
    shift = torch.empty(F_, dtype=torch.float32, device=dev)
    acc = torch.empty(_lib.bn_acc_doubles(F_), dtype=torch.float64, device=dev)
    _lib.call("gcmi_bn_stats", _ptr(x), _ld(x), n, F_, _ptr(_vec(gamma, "gamma", F_)),
              _ptr(_vec(beta, "beta", F_)), float(eps), float(momentum),
              _ptr(_vec(running_mean, "running_mean", F_)), _ptr(_vec(running_var, "running_var", F_)),
              _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), _ptr(acc), _stream())
    return mean, invstd, scale, shift


def bn_fold_eval(gamma, beta, running_mean, running_var, eps: float):
    F_ = running_mean.numel()
    scale = torch.empty(F_, dtype=torch.float32, device=running_mean.device)
    shift = torch.empty_like(scale)
    _lib.call("gcmi_bn_fold_eval", _ptr(_vec(gamma, "gamma", F_)), _ptr(_vec(beta, "beta", F_)),
              _ptr(_vec(running_mean, "running_mean", F_)), _ptr(_vec(running_var, "running_var", F_)),
              float(eps), F_, _ptr(scale), _ptr(shift), _stream())
    return scale, shift


def bn_apply(x, scale, shift):
    _mat(x, "x")
    n, F_ = x.shape
    y = torch.empty((n, F_), dtype=torch.float32, device=x.device)
    _lib.call("gcmi_bn_apply", _ptr(x), _ld(x), n, F_, _ptr(_vec(scale, "scale", F_)),
              _ptr(_vec(shift, "shift", F_)), _ptr(y), _ld(y), _stream())
    return y


def bn_bwd(dy, x, gamma, mean, invstd, need_dx: bool, relu_mask: bool = False):
    _mat(dy, "dy")
    _mat(x, "x", rows=dy.shape[0], cols=dy.shape[1])
    n, F_ = x.shape
    dev = x.device
    dgamma = torch.empty(F_, dtype=torch.float32, device=dev)
    dbeta = torch.empty(F_, dtype=torch.float32, device=dev)
    dx = torch.empty((n, F_), dtype=torch.float32, device=dev) if need_dx else None
    acc = torch.empty(_lib.bn_acc_doubles(F_), dtype=torch.float64, device=dev)
    _lib.call("gcmi_bn_bwd", _ptr(dy), _ld(dy), _ptr(x), _ld(x), n, F_, _ptr(_vec(gamma, "gamma", F_)),
              _ptr(_vec(mean, "mean", F_)), _ptr(_vec(invstd, "invstd", F_)), _ptr(dgamma), _ptr(dbeta),
              _ptr(dx), _ld(dx) if need_dx else 0, 1 if relu_mask else 0, _ptr(acc), _stream())
    return dgamma, dbeta, dx


def _i32arr(vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def _i64arr(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


def seg_gemm(seg_begin, seg_end, a1, w1, w1_off, a2, w2, w2_off, bias, bias_off, n_out: int,
             trans_w: bool, relu: bool, n_rows: int, k1: int = 0, k2: int = 0,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[n_rows, n_out]; rows not covered by any segment are left undefined.
    w1 / w2 / bias are flat float32 CUDA tensors holding the blocks at the
    given offsets (in floats).  ``out``: write into this (n_rows, n_out) matrix (a column block of a
    wider one is fine) instead of a fresh one."""
    n_seg = len(seg_begin)
    dev = (a1 if a1 is not None else a2).device
    for s in range(n_seg):
        if not (0 <= seg_begin[s] <= seg_end[s] <= n_rows):
            raise ValueError("segment %d = [%d,%d) outside [0,%d]" % (s, seg_begin[s], seg_end[s], n_rows))
    for a, w, off, k, nm in ((a1, w1, w1_off, k1, "a1"), (a2, w2, w2_off, k2, "a2")):
        if a is None:
            continue
        _mat(a, nm, rows=n_rows, cols=k)
        if w is None or w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous():
            raise ValueError("weights of %s must be a contiguous float32 CUDA tensor" % nm)
        for s in range(n_seg):
            if off[s] >= 0 and off[s] + k * n_out > w.numel():
                raise ValueError("weight block %d of %s runs past the buffer" % (s, nm))
    if bias is not None:
        if bias.dtype != torch.float32 or not bias.is_cuda or not bias.is_contiguous():
            raise ValueError("bias must be a contiguous float32 CUDA tensor")
        for s in range(n_seg):
            if bias_off[s] >= 0 and bias_off[s] + n_out > bias.numel():
                raise ValueError("bias block %d runs past the buffer" % s)
    if out is None:
        out = torch.empty((n_rows, n_out), dtype=torch.float32, device=dev)
    else:
        _mat(out, "out", rows=n_rows, cols=n_out)
    _lib.call("gcmi_seg_gemm", n_seg, _i32arr(seg_begin), _i32arr(seg_end),
              _ptr(a1), _ld(a1) if a1 is not None else 0, k1, _ptr(w1),
              _i64arr(w1_off) if a1 is not None else None,
              _ptr(a2), _ld(a2) if a2 is not None else 0, k2, _ptr(w2),
              _i64arr(w2_off) if a2 is not None else None,
              _ptr(bias), _i64arr(bias_off) if bias is not None else None, n_out,
              1 if trans_w else 0, 1 if relu else 0, _ptr(out), _ld(out), _stream())
    return out


def task_head_forward(fingerprint: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                      scratch: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """logits = fingerprint . weight^T + bias with nn.Linear's (n_out, k) weight, as the whole-model path runs the task
    head (gcmi_task_head_forward): with 33..256 outputs and a ``scratch`` of ``task_head_scratch_floats()`` floats, from
    fragment images of the weight prepared in it."""
    _mat(fingerprint, "fingerprint")
    n, k = fingerprint.shape
    n_out = weight.shape[0]
    if weight.shape != (n_out, k) or weight.dtype != torch.float32 or not weight.is_cuda or not weight.is_contiguous():
        raise ValueError("weight must be a contiguous float32 CUDA tensor of shape (n_out, %d)" % k)
    if out is None:
        out = torch.empty((n, n_out), dtype=torch.float32, device=fingerprint.device)
    _lib.call("gcmi_task_head_forward", _ptr(fingerprint), _ld(fingerprint), n, k, _ptr(weight),
              _ptr(bias) if bias is not None else None, n_out, _ptr(scratch) if scratch is not None else None, _ptr(out),
              _ld(out), _stream())
    return out


def task_head_scratch_floats() -> int:
    return int(_lib.load().gcmi_task_head_scratch_floats())


def seg_gemm_wgrad(seg_begin, seg_end, a, g, dw, dw_off, dbias, dbias_off, trans_w: bool):
    """dw (+)= a^T g per segment; dw / dbias are flat, pre-zeroed, accumulated in place."""
    n_seg = len(seg_begin)
    _mat(a, "a")
    _mat(g, "g", rows=a.shape[0])
    k, n = a.shape[1], g.shape[1]
    for s in range(n_seg):
        if not (0 <= seg_begin[s] <= seg_end[s] <= a.shape[0]):
            raise ValueError("bad segment %d" % s)
        if dw_off[s] >= 0 and dw_off[s] + k * n > dw.numel():
            raise ValueError("dw block %d runs past the buffer" % s)
        if dbias is not None and dbias_off[s] >= 0 and dbias_off[s] + n > dbias.numel():
            raise ValueError("dbias block %d runs past the buffer" % s)
    if dw.dtype != torch.float32 or not dw.is_cuda or not dw.is_contiguous():
        raise ValueError("dw must be a contiguous float32 CUDA tensor")
    _lib.call("gcmi_seg_gemm_wgrad", n_seg, _i32arr(seg_begin), _i32arr(seg_end), _ptr(a), _ld(a), k,
              _ptr(g), _ld(g), n, _ptr(dw), _i64arr(dw_off), _ptr(dbias),
              _i64arr(dbias_off) if dbias is not None else None, 1 if trans_w else 0, _stream())


def relu_bwd_(g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    _mat(g, "g")
    _mat(y, "y", rows=g.shape[0], cols=g.shape[1])
    _lib.call("gcmi_relu_bwd", _ptr(g), _ld(g), _ptr(y), _ld(y), g.shape[0], g.shape[1], _stream())
    return g


def loss_fwd_bwd(kind: int, logits, labels, weights, want_probs: bool = False):
    """(loss scalar tensor, dlogits, probs|None).  kind 0: softmax CE over the last
    dim of (rows, tasks, classes); kind 1: L2 over (rows, tasks)."""
    if not logits.is_cuda:
        raise _lib.GcmiError("loss needs CUDA tensors")
    logits = logits.contiguous()
    labels = labels.contiguous().to(torch.float32)
    if kind == 0:
        if logits.dim() == 2:
            logits = logits.unsqueeze(1)
            labels = labels.unsqueeze(1)
        n_rows, n_tasks, n_classes = logits.shape
    else:
        if logits.dim() == 1:
            logits = logits.unsqueeze(1)
        n_rows, n_tasks = logits.shape[0], logits[0].numel()
        n_classes = 1
    if labels.numel() != logits.numel():
        raise ValueError("labels %s do not match outputs %s" % (tuple(labels.shape), tuple(logits.shape)))
    if weights is not None:
        weights = weights.contiguous().to(torch.float32)
        if weights.numel() != n_rows * n_tasks:
            raise ValueError("weights %s do not match (%d, %d)" % (tuple(weights.shape), n_rows, n_tasks))
    dev = logits.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dlogits = torch.empty_like(logits)
    probs = torch.empty_like(logits) if (want_probs and kind == 0) else None
    acc = torch.empty(1, dtype=torch.float64, device=dev)
    _lib.call("gcmi_loss_fwd_bwd", kind, _ptr(logits), _ptr(labels), _ptr(weights), n_rows, n_tasks,
              n_classes, _ptr(loss), _ptr(dlogits), _ptr(probs), _ptr(acc), _stream())
    return loss, dlogits, probs


def softmax_lastdim(logits: torch.Tensor) -> torch.Tensor:
    x = logits.contiguous()
    out = torch.empty_like(x)
    c = x.shape[-1]
    _lib.call("gcmi_softmax", _ptr(x), x.numel() // max(c, 1), c, _ptr(out), _stream())
    return out


def adam_step_(p, grad, m, v, lr, beta1, beta2, eps, step: int):
    n = p.numel()
    for t, nm in ((p, "param"), (grad, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError("adam: %s must be a contiguous float32 CUDA tensor of %d" % (nm, n))
    _lib.call("gcmi_adam_step", _ptr(p), _ptr(grad), _ptr(m), _ptr(v), n, float(lr), float(beta1),
              float(beta2), float(eps), int(step), _stream())


# ------------------------------------------------------------------ Weave
def fold_affine(w: torch.Tensor, b: Optional[torch.Tensor], scale: Optional[torch.Tensor],
                shift: Optional[torch.Tensor], trans_w: bool = False):
    """(W * diag(scale), b*scale + shift): an eval-mode BatchNorm folded into the preceding product."""
    w = _mat(w, "w")
    k, n = (w.shape[1], w.shape[0]) if trans_w else (w.shape[0], w.shape[1])
    w_out = torch.empty_like(w)
    b_out = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.call("gcmi_fold_affine", _ptr(w), _ptr(_vec(b, "b", n)), _ptr(_vec(scale, "scale", n)),
              _ptr(_vec(shift, "shift", n)), k, n, 1 if trans_w else 0, _ptr(w_out), _ptr(b_out), _stream())
    return w_out, b_out


def _i32vec(t: torch.Tensor, name: str, n: Optional[int] = None) -> torch.Tensor:
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise _lib.GcmiError("%s must be a contiguous int32 CUDA tensor" % name)
    if n is not None and t.numel() != n:
        raise _lib.GcmiError("%s has %d entries, expected %d" % (name, t.numel(), n))
    return t


def weave_pair_to_atom(pair_feat: torch.Tensor, pair_src: torch.Tensor, n_atoms: int, w: torch.Tensor, b: torch.Tensor):
    """sum over the pairs of every source atom (pair_src ascending) of relu(pair_feat . w + b)."""
    pf = _mat(pair_feat, "pair_feat")
    w = _mat(w, "w", rows=pf.shape[1])
    H = w.shape[1]
    out = torch.empty((n_atoms, H), dtype=torch.float32, device=pf.device)
    _lib.call("gcmi_weave_pair_to_atom", _ptr(pf), _ld(pf), pf.shape[1], _ptr(_i32vec(pair_src, "pair_src", pf.shape[0])),
              pf.shape[0], n_atoms, _ptr(w), _ptr(_vec(b, "b", H)), H, _ptr(out), _ld(out), _stream())
    return out


def weave_pair_features(u: torch.Tensor, v: torch.Tensor, b_ap, pair_feat: torch.Tensor, w_pp, b_pp,
                        atom_to_pair: torch.Tensor, out: Optional[torch.Tensor] = None):
    """[relu(U[i]+V[j]+b) + relu(U[j]+V[i]+b) | relu(pair_feat . w_pp + b_pp)] per ordered pair.
    ``w_pp=None``: only the atom-pair block.  ``out``: write into this (P, H + H2) matrix (may be a column
    block of a wider one)."""
    u, v = _mat(u, "u"), _mat(v, "v", rows=u.shape[0], cols=u.shape[1])
    if _ld(u) != _ld(v):
        raise _lib.GcmiError("u and v must share their leading dimension")
    pf = _mat(pair_feat, "pair_feat")
    if w_pp is not None:
        w_pp = _mat(w_pp, "w_pp", rows=pf.shape[1])
    P, H, H2 = pf.shape[0], u.shape[1], 0 if w_pp is None else w_pp.shape[1]
    if out is None:
        z = torch.empty((P, H + H2), dtype=torch.float32, device=pf.device)
    else:
        z = _mat(out, "out", rows=P, cols=H + H2)
    _lib.call("gcmi_weave_pair_features", _ptr(u), _ptr(v), _ld(u), H, _ptr(_vec(b_ap, "b_ap", H)), _ptr(pf), _ld(pf),
              pf.shape[1], _ptr(w_pp), _ptr(_vec(b_pp, "b_pp", H2)) if H2 else None, H2,
              _ptr(_i32vec(atom_to_pair, "atom_to_pair", 2 * P)), P, _ptr(z), _ld(z), _stream())
    return z


def weave_gather(x: torch.Tensor, mol_ptr: torch.Tensor, gaussian_expand: bool):
    x = _mat(x, "x")
    n_mols = mol_ptr.numel() - 1
    out = torch.empty((n_mols, x.shape[1] * (11 if gaussian_expand else 1)), dtype=torch.float32, device=x.device)
    _lib.call("gcmi_weave_gather", _ptr(x), _ld(x), x.shape[1], _ptr(_i32vec(mol_ptr, "mol_ptr")), n_mols,
              1 if gaussian_expand else 0, _ptr(out), _ld(out), _stream())
    return out


def tanh_(x: torch.Tensor) -> torch.Tensor:
    x = _mat(x, "x")
    _lib.call("gcmi_tanh_", _ptr(x), _ld(x), x.shape[0], x.shape[1], _stream())
    return x


# ------------------------------------------------------------------ message passing
def edge_network_sum(g: torch.Tensor, n_hidden: int, pair_feat: torch.Tensor, dst_ptr: torch.Tensor,
                     src: torch.Tensor) -> torch.Tensor:
    g = _mat(g, "g")
    pf = _mat(pair_feat, "pair_feat")
    K = pf.shape[1]
    if g.shape[1] != (K + 1) * n_hidden:
        raise ValueError("g must have (n_pair_feat + 1) * n_hidden columns")
    n_dst = dst_ptr.numel() - 1
    out = torch.empty((n_dst, n_hidden), dtype=torch.float32, device=g.device)
    _lib.call("gcmi_edge_network_sum", _ptr(g), _ld(g), n_hidden, K, _ptr(pf), _ld(pf),
              _ptr(_i32vec(dst_ptr, "dst_ptr")), _ptr(_i32vec(src, "src", pf.shape[0])), n_dst, _ptr(out), _ld(out),
              _stream())
    return out


def expand_atom_codes(codes: torch.Tensor, ld_out: int = 76) -> torch.Tensor:
    """(N, >= 8) uint8 code rows (row stride a multiple of 8 bytes) -> (N, ld_out) float32 feature rows."""
    if not torch.is_tensor(codes) or not codes.is_cuda or codes.dtype != torch.uint8 or codes.dim() != 2:
        raise _lib.GcmiError("codes must be a 2-D uint8 CUDA tensor")
    if codes.shape[1] < 8 or (codes.shape[1] > 1 and codes.stride(1) != 1):
        raise ValueError("code rows are 8 contiguous bytes")
    n = codes.shape[0]
    out = torch.empty((n, ld_out), dtype=torch.float32, device=codes.device)
    _lib.call("gcmi_expand_atom_codes", _ptr(codes), int(codes.stride(0)) if n > 1 else int(codes.shape[1]), n,
              _ptr(out), ld_out, _stream())
    return out


def edge_network_moments(h: torch.Tensor, pair_feat: torch.Tensor, dst_ptr: torch.Tensor, src: torch.Tensor,
                         mol_ptr: Optional[torch.Tensor] = None, max_mol_atoms: int = 0):
    """T[i] = [sum_p pf[p,0] h[src_p] | ... | sum_p pf[p,K-1] h[src_p] | sum_p h[src_p]] per destination atom.
    ``mol_ptr`` (int32 CSR of the atoms per molecule): the molecule-staged kernel (same T)."""
    h = _mat(h, "h")
    pf = _mat(pair_feat, "pair_feat")
    d, K = h.shape[1], pf.shape[1]
    n_dst = dst_ptr.numel() - 1
    t = torch.empty((n_dst, (K + 1) * d), dtype=torch.float32, device=h.device)
    if mol_ptr is not None:
        mp = _i32vec(mol_ptr, "mol_ptr")
        _lib.call("gcmi_edge_network_moments_mol", _ptr(h), _ld(h), d, K, _ptr(pf), _ld(pf),
                  _ptr(_i32vec(dst_ptr, "dst_ptr")), _ptr(_i32vec(src, "src", pf.shape[0])), n_dst, _ptr(mp),
                  mp.numel() - 1, int(max_mol_atoms), _ptr(t), _ld(t), _stream())
        return t
    _lib.call("gcmi_edge_network_moments", _ptr(h), _ld(h), d, K, _ptr(pf), _ld(pf), _ptr(_i32vec(dst_ptr, "dst_ptr")),
              _ptr(_i32vec(src, "src", pf.shape[0])), n_dst, _ptr(t), _ld(t), _stream())
    return t


def gru_gates_(z: torch.Tensor, r: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    for t, nm in ((z, "z"), (r, "r"), (h, "h")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == z.shape):
            raise ValueError("%s must be a contiguous float32 CUDA tensor of the gate shape" % nm)
    hr = torch.empty_like(h)
    _lib.call("gcmi_gru_gates", _ptr(z), _ptr(r), _ptr(h), _ptr(hr), z.numel(), _stream())
    return hr


def gru_out(z: torch.Tensor, hpre: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    for t, nm in ((z, "z"), (hpre, "hpre"), (x, "x")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == z.shape):
            raise ValueError("%s must be a contiguous float32 CUDA tensor of the gate shape" % nm)
    out = torch.empty_like(x)
    _lib.call("gcmi_gru_out", _ptr(z), _ptr(hpre), _ptr(x), _ptr(out), z.numel(), _stream())
    return out


def set2set_attend(x: torch.Tensor, mol_ptr: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    x = _mat(x, "x")
    n_mols = mol_ptr.numel() - 1
    h = _mat(h, "h", rows=n_mols, cols=x.shape[1])
    q = torch.empty((n_mols, 2 * x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.call("gcmi_set2set_attend", _ptr(x), _ld(x), x.shape[1], _ptr(_i32vec(mol_ptr, "mol_ptr")), n_mols, _ptr(h),
              _ld(h), _ptr(q), _ld(q), _stream())
    return q


def lstm_cell_(z: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    z = _mat(z, "z")
    H = z.shape[1] // 4
    c = _mat(c, "c", rows=z.shape[0], cols=H)
    h = torch.empty_like(c)
    _lib.call("gcmi_lstm_cell", _ptr(z), _ld(z), H, z.shape[0], _ptr(c), _ld(c), _ptr(h), _ld(h), _stream())
    return h


def timing_enable(kernel_id: int, on: bool = True):
    _lib.call("gcmi_timing_enable", kernel_id, 1 if on else 0)


def timing_read(kernel_id: int, reset: bool = True):
    n = ctypes.c_int64(0)
    ms = ctypes.c_double(0.0)
    _lib.call("gcmi_timing_read", kernel_id, ctypes.byref(n), ctypes.byref(ms), 1 if reset else 0)
    return int(n.value), float(ms.value)


# ------------------------------------------------------------------ autograd glue
def _graphconv_offsets(max_deg: int, k: int, n_out: int):
    """Offsets (floats) of the per-degree blocks inside the packed parameter
    order of the reference: rel_1, self_1, ..., rel_max, self_max, self_0
    (models/torch_models/layers.py:6189-6224).  Segment d = degree d."""
    blk = k * n_out
    w_rel = [-1] + [(2 * (d - 1)) * blk for d in range(1, max_deg + 1)]
    w_self = [2 * max_deg * blk] + [(2 * (d - 1) + 1) * blk for d in range(1, max_deg + 1)]
    b_off = [d * n_out for d in range(max_deg + 1)]
    return w_rel, w_self, b_off


class GraphConvFn(torch.autograd.Function):
    """out = act(S . W_rel[deg] + X . W_self[deg] + bsum[deg]),  S = gather-sum(X).
    wpack: (2*max_deg+1, K, n_out) in reference order; bsum: (max_deg+1, n_out)."""

    @staticmethod
    def forward(ctx, x, wpack, bsum, graph: BatchGraph, relu: bool, grad_masked: bool = False):
        """grad_masked: the consumer's backward already multiplies by (out > 0) (a folded
        BatchNorm does), so the ReLU derivative is not applied again."""
        x = rowmajor(x)
        wpack = wpack.contiguous()
        bsum = bsum.contiguous()
        k, n_out = wpack.shape[1], wpack.shape[2]
        if wpack.shape[0] != 2 * graph.max_deg + 1 or tuple(bsum.shape) != (graph.max_deg + 1, n_out):
            raise ValueError("GraphConv parameters do not match max_deg=%d" % graph.max_deg)
        _mat(x, "atom_features", rows=graph.n_atoms)
        if x.shape[1] < k or x.shape[1] >= k + 4:
            raise ValueError("atom_features has %d columns, the layer expects %d" % (x.shape[1], k))
        # x may carry zero padding columns (DeviceBatch pads 75 -> 76 so that the gather moves
        # 16 bytes per lane); the GEMMs read the first k columns only
        w_rel, w_self, b_off = _graphconv_offsets(graph.max_deg, k, n_out)
        s = gather_sum(graph, x)
        out = seg_gemm(list(graph.seg_begin), list(graph.seg_end), s[:, :k], wpack, w_rel, x[:, :k],
                       wpack, w_self, bsum, b_off, n_out, False, relu, graph.n_atoms, k, k)
        ctx.graph = graph
        ctx.relu = relu and not grad_masked
        ctx.save_for_backward(x, s, wpack, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, s, wpack, out = ctx.saved_tensors
        graph = ctx.graph
        k, n_out = wpack.shape[1], wpack.shape[2]
        w_rel, w_self, b_off = _graphconv_offsets(graph.max_deg, k, n_out)
        sb, se = list(graph.seg_begin), list(graph.seg_end)
        g = rowmajor(dout)
        if ctx.relu:
            g = relu_bwd_(g.clone(), out)
        dw = dbs = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.zeros_like(wpack)
            dbs = torch.zeros((graph.max_deg + 1, n_out), dtype=torch.float32, device=g.device)
            seg_gemm_wgrad(sb, se, s[:, :k], g, dw, w_rel, None, None, False)
            seg_gemm_wgrad(sb, se, x[:, :k], g, dw, w_self, dbs, b_off, False)
        if ctx.needs_input_grad[0]:
            # dS = g . W_rel^T, dX = g . W_self^T  (same blocks read transposed)
            ds = seg_gemm(sb, se, g, wpack, w_rel, None, None, None, None, None, k, True, False,
                          graph.n_atoms, n_out, 0)
            dx = seg_gemm(sb, se, g, wpack, w_self, None, None, None, None, None, k, True, False,
                          graph.n_atoms, n_out, 0)
            if graph.symmetric:
                gather_sum(graph, ds, dx, accumulate=True)  # no atomics: bonds are listed from both ends
            else:
                scatter_add(graph, ds, dx)
            if x.shape[1] > k:
                dx = torch.nn.functional.pad(dx, (0, x.shape[1] - k))
        return dx, dw, dbs, None, None, None


class PoolFn(torch.autograd.Function):
    """GraphPool, optionally with the preceding BatchNorm1d folded into it."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, graph: BatchGraph, bn: bool,
                training: bool, eps: float, momentum: float, relu_in: bool = False):
        """relu_in: x is the output of a fused ReLU whose producer was told
        ``grad_masked``; the returned gradient is then w.r.t. the ReLU input."""
        x = rowmajor(x)
        _mat(x, "atom_features", rows=graph.n_atoms)
        mean = invstd = scale = shift = None
        if bn:
            if training:
                mean, invstd, scale, shift = bn_stats(x, gamma, beta, running_mean, running_var, eps,
                                                      momentum)
            else:
                scale, shift = bn_fold_eval(gamma, beta, running_mean, running_var, eps)
        out, arg = gather_max(graph, x, scale, shift)
        if relu_in and not (bn and training):
            raise ValueError("relu_in needs a training-mode BatchNorm to fold the mask into")
        ctx.graph, ctx.bn, ctx.training, ctx.relu_in = graph, bn, training, relu_in
        ctx.save_for_backward(x, gamma, mean, invstd, scale, arg)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, mean, invstd, scale, arg = ctx.saved_tensors
        dy = gather_max_bwd(ctx.graph, rowmajor(dout), arg)
        if not ctx.bn:
            return (dy,) + (None,) * 10
        if not ctx.training:
            raise NotImplementedError("gradients through an eval-mode BatchNorm are not implemented")
        dgamma, dbeta, dx = bn_bwd(dy, x, gamma, mean, invstd, ctx.needs_input_grad[0], ctx.relu_in)
        return (dx, dgamma, dbeta) + (None,) * 8


class ReadoutFn(torch.autograd.Function):
    """GraphGather ([sum | max] per molecule, optional tanh), optionally with the
    preceding BatchNorm1d folded into it."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, graph: BatchGraph, n_mols: int,
                bn: bool, training: bool, eps: float, momentum: float, tanh: bool,
                relu_in: bool = False):
        x = rowmajor(x)
        _mat(x, "atom_features", rows=graph.n_atoms)
        mean = invstd = scale = shift = None
        if bn:
            if training:
                mean, invstd, scale, shift = bn_stats(x, gamma, beta, running_mean, running_var, eps,
                                                      momentum)
            else:
                scale, shift = bn_fold_eval(gamma, beta, running_mean, running_var, eps)
        out, arg = readout(graph, x, n_mols, scale, shift, tanh)
        if relu_in and not (bn and training):
            raise ValueError("relu_in needs a training-mode BatchNorm to fold the mask into")
        ctx.graph, ctx.bn, ctx.training, ctx.tanh, ctx.relu_in = graph, bn, training, tanh, relu_in
        ctx.save_for_backward(x, gamma, mean, invstd, out, arg)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, mean, invstd, out, arg = ctx.saved_tensors
        dy = readout_bwd(ctx.graph, rowmajor(dout), out, arg, ctx.tanh)
        if not ctx.bn:
            return (dy,) + (None,) * 12
        if not ctx.training:
            raise NotImplementedError("gradients through an eval-mode BatchNorm are not implemented")
        dgamma, dbeta, dx = bn_bwd(dy, x, gamma, mean, invstd, ctx.needs_input_grad[0], ctx.relu_in)
        return (dx, dgamma, dbeta) + (None,) * 10


# ---- parameter gradients straight into ``p.grad``
# The weight-gradient kernels ADD into their output.  When a training step has put a zeroed view of one flat arena
# behind every ``p.grad`` (deepchem_amd.dist.FlatGradArena.attach) and switches this on around ``backward()``, the
# autograd functions below and in models/torch_models hand the kernels ``p.grad`` itself and return no gradient for
# the parameter: a weight that is used T times (the message rounds of MPNN share theirs) then costs neither T
# temporaries with their zero fills nor T - 1 additions by the autograd engine.
_DIRECT_GRAD = [False]


class direct_param_grads:
    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _DIRECT_GRAD[0]
        _DIRECT_GRAD[0] = self.on
        return self

    def __exit__(self, *exc):
        _DIRECT_GRAD[0] = self.prev
        return False


def grad_target(p):
    """``p.grad`` if gradients of ``p`` may be accumulated in place right now, else None."""
    if not _DIRECT_GRAD[0] or p is None or not isinstance(p, torch.nn.Parameter):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


class LinearFn(torch.autograd.Function):
    """act(x . W^T + b) with nn.Linear's (out, in) weight layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool, grad_masked: bool = False):
        x = rowmajor(x)
        weight = weight.contiguous()
        n_out, k = weight.shape
        _mat(x, "x", cols=k)
        n = x.shape[0]
        out = seg_gemm([0], [n], x, weight, [0], None, None, None,
                       None if bias is None else bias.contiguous(), [0], n_out, True, relu, n, k, 0)
        ctx.relu = relu and not grad_masked
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        n_out, k = weight.shape
        n = x.shape[0]
        g = rowmajor(dout)
        if ctx.relu:
            g = relu_bwd_(g.clone(), out)
        dw = db = dx = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            tw, tb = grad_target(ctx.params[0]), grad_target(ctx.params[1])
            dw = tw if tw is not None else torch.zeros_like(weight)
            db = None
            if ctx.has_bias:
                db = tb if tb is not None else torch.zeros(n_out, dtype=torch.float32, device=g.device)
            if n > 0:
                seg_gemm_wgrad([0], [n], x, g, dw, [0], db, [0], True)
            if tw is not None:
                dw = None
            if tb is not None:
                db = None
        if ctx.needs_input_grad[0]:
            dx = seg_gemm([0], [n], g, weight, [0], None, None, None, None, None, k, False, False, n,
                          n_out, 0)
        return dx, dw, db, None, None


class StandardLossFn(torch.autograd.Function):
    """mean(w * loss(outputs, labels)) with the gradient produced in the same pass."""

    @staticmethod
    def forward(ctx, outputs, labels, weights, kind: int):
        loss, dlogits, _ = loss_fwd_bwd(kind, outputs, labels, weights)
        ctx.save_for_backward(dlogits)
        ctx.shape = outputs.shape
        return loss

    @staticmethod
    def backward(ctx, gl):
        (dlogits,) = ctx.saved_tensors
        return (dlogits * gl).reshape(ctx.shape), None, None, None


class SoftmaxFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits):
        p = softmax_lastdim(logits)
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, gp):
        (p,) = ctx.saved_tensors
        return p * (gp - (gp * p).sum(-1, keepdim=True))

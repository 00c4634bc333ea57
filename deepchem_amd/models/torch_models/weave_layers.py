"""WeaveLayer / WeaveGather with the reference's layer contract
(deepchem/models/torch_models/layers.py:4135-4429 and :4432-4648), computed by libgcmi.so.

Same constructors, same attribute names (``W_AA`` ... ``b_P`` are plain tensors, the ``*_bn`` are
``nn.BatchNorm1d`` modules: tests and ``Weave.__init__`` assign and re-initialise them in place),
same ``forward(inputs)`` with NumPy arrays or tensors.  Reference behaviour kept on purpose:

* every BatchNorm of this path runs in eval mode (layers.py:4361 ``self.AA_bn.eval()`` etc.), i.e.
  it is an affine map from the running statistics;
* ``forward`` re-wraps its inputs (``torch.tensor(inputs[0])``, layers.py:4350-4353 and :4581), so
  no gradient reaches the weave weights or flows from one layer into the previous one.

What is different is the arithmetic: the BatchNorms are folded into the weights, the pair -> atom
reduction and the Gaussian-histogram gather are single fused kernels, and the atom -> pair products
are evaluated once per ATOM (U = A.W_AP[:Fa], V = A.W_AP[Fa:]) and gathered per pair instead of a
``(n_pairs, 2*Fa)`` gathered matmul per ordering (include/gcmi.h, Weave section).
"""
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init as initializers

from deepchem_amd import ops
from deepchem_amd._lib import GcmiError


def _dev_f32(x, device) -> torch.Tensor:
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    return ops.rowmajor(t.detach().to(device=device, dtype=torch.float32))


def _host_i64(x) -> np.ndarray:
    if torch.is_tensor(x):
        x = x.detach().cpu().numpy()
    return np.asarray(x).astype(np.int64)


def _csr_from_sorted(ids: np.ndarray, n: int, what: str) -> np.ndarray:
    if ids.size and (np.any(np.diff(ids) < 0) or ids[0] < 0 or ids[-1] >= n):
        raise ValueError("%s must be ascending and inside [0, %d)" % (what, n))
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(ids, minlength=n), out=ptr[1:])
    return ptr.astype(np.int32)


class _PairPlan:
    """Device copies of the pair index arrays of one batch (validated once)."""

    def __init__(self, pair_split, atom_to_pair, n_atoms: int, device):
        ps = _host_i64(pair_split)
        a2p = _host_i64(atom_to_pair).reshape(-1, 2)
        n_pairs = ps.shape[0]
        if a2p.shape[0] != n_pairs:
            raise ValueError("pair_split / atom_to_pair do not match (%d vs %d pairs)" % (n_pairs, a2p.shape[0]))
        if n_pairs and (a2p.min() < 0 or a2p.max() >= n_atoms):
            raise ValueError("atom_to_pair refers to atoms outside [0, %d)" % n_atoms)
        if n_pairs and (np.any(np.diff(ps) < 0) or ps[0] < 0 or ps[-1] >= n_atoms):
            raise ValueError("pair_split must be ascending and inside [0, %d)" % n_atoms)
        if n_atoms and np.bincount(ps, minlength=n_atoms).min() == 0:
            raise ValueError("every atom needs at least one pair (its self pair)")  # reference: shape error at the concat
        self.n_pairs, self.n_atoms = n_pairs, n_atoms
        self.pair_src = torch.from_numpy(ps.astype(np.int32)).to(device)
        self.a2p = torch.from_numpy(a2p.astype(np.int32)).to(device).contiguous().view(-1)


_plan_cache = []  # [(pair_split object, atom_to_pair object, n_atoms, plan)], newest first


def _pair_plan(pair_split, atom_to_pair, n_atoms: int, device) -> _PairPlan:
    """The stacked layers of a model receive the same index arrays: build their device form once."""
    for ps, ap, n, plan in _plan_cache:
        if ps is pair_split and ap is atom_to_pair and n == n_atoms and plan.pair_src.device == device:
            return plan
    plan = _PairPlan(pair_split, atom_to_pair, n_atoms, device)
    _plan_cache.insert(0, (pair_split, atom_to_pair, n_atoms, plan))
    del _plan_cache[2:]
    return plan


def _require_relu(activation: str, what: str):
    if activation != 'relu':
        raise GcmiError("%s: only activation='relu' has a kernel (got %r)" % (what, activation))


class WeaveLayer(nn.Module):
    """One weave module: atom and pair features exchange information (A->A, P->A, A->P, P->P)."""

    def __init__(self, n_atom_input_feat: int = 75, n_pair_input_feat: int = 14, n_atom_output_feat: int = 50,
                 n_pair_output_feat: int = 50, n_hidden_AA: int = 50, n_hidden_PA: int = 50, n_hidden_AP: int = 50,
                 n_hidden_PP: int = 50, update_pair: bool = True, init_: str = 'xavier_uniform_',
                 activation: str = 'relu', batch_normalize: bool = True, device=None, **kwargs):
        super(WeaveLayer, self).__init__(**kwargs)
        _require_relu(activation, "WeaveLayer")
        self.init = init_
        self.activation = activation
        self.update_pair = update_pair
        self.n_hidden_AA, self.n_hidden_PA = n_hidden_AA, n_hidden_PA
        self.n_hidden_AP, self.n_hidden_PP = n_hidden_AP, n_hidden_PP
        self.n_hidden_A = n_hidden_AA + n_hidden_PA
        self.n_hidden_P = n_hidden_AP + n_hidden_PP
        self.batch_normalize = batch_normalize
        self.n_atom_input_feat, self.n_pair_input_feat = n_atom_input_feat, n_pair_input_feat
        self.n_atom_output_feat, self.n_pair_output_feat = n_atom_output_feat, n_pair_output_feat
        self.device = torch.device("cuda:0") if device is None else torch.device(device)
        init = getattr(initializers, self.init)

        def weight(k, n):
            return init(torch.empty(k, n)).to(self.device)

        def bias(n):
            return torch.zeros((n,), device=self.device)

        def bn(n):
            return nn.BatchNorm1d(num_features=n, eps=1e-3, momentum=0.99, affine=True,
                                  track_running_stats=True).to(self.device)

        self.W_AA, self.b_AA, self.AA_bn = weight(n_atom_input_feat, n_hidden_AA), bias(n_hidden_AA), bn(n_hidden_AA)
        self.W_PA, self.b_PA, self.PA_bn = weight(n_pair_input_feat, n_hidden_PA), bias(n_hidden_PA), bn(n_hidden_PA)
        self.W_A, self.b_A, self.A_bn = (weight(self.n_hidden_A, n_atom_output_feat), bias(n_atom_output_feat),
                                         bn(n_atom_output_feat))
        if self.update_pair:
            self.W_AP, self.b_AP, self.AP_bn = (weight(n_atom_input_feat * 2, n_hidden_AP), bias(n_hidden_AP),
                                                bn(n_hidden_AP))
            self.W_PP, self.b_PP, self.PP_bn = (weight(n_pair_input_feat, n_hidden_PP), bias(n_hidden_PP),
                                                bn(n_hidden_PP))
            self.W_P, self.b_P, self.P_bn = (weight(self.n_hidden_P, n_pair_output_feat), bias(n_pair_output_feat),
                                             bn(n_pair_output_feat))
        self.built = True

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}(n_atom_input_feat:{self.n_atom_input_feat},'
                f'n_pair_input_feat:{self.n_pair_input_feat},n_atom_output_feat:{self.n_atom_output_feat},'
                f'n_pair_output_feat:{self.n_pair_output_feat},n_hidden_AA:{self.n_hidden_AA},'
                f'n_hidden_PA:{self.n_hidden_PA},n_hidden_AP:{self.n_hidden_AP},n_hidden_PP:{self.n_hidden_PP},'
                f'batch_normalize:{self.batch_normalize},update_pair:{self.update_pair},init:{self.init},'
                f'activation:{self.activation})')

    def _folded(self, w: torch.Tensor, b: torch.Tensor, bn: nn.BatchNorm1d):
        """(W', b') with the eval-mode BatchNorm that follows the product folded in."""
        w = ops.rowmajor(w.detach().to(self.device, torch.float32))
        b = b.detach().to(self.device, torch.float32).contiguous()
        if not self.batch_normalize:
            return w, b
        scale, shift = ops.bn_fold_eval(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        return ops.fold_affine(w, b, scale, shift)

    @staticmethod
    def _linear(x, w, b, relu: bool, x2=None, w2=None):
        n = x.shape[0]
        return ops.seg_gemm([0], [n], x, w.reshape(-1), [0], x2, None if w2 is None else w2.reshape(-1),
                            None if w2 is None else [0], b, None if b is None else [0], w.shape[1], False, relu, n,
                            x.shape[1], 0 if x2 is None else x2.shape[1])

    def forward(self, inputs: List) -> List[torch.Tensor]:
        """inputs = [atom_features, pair_features, pair_split, atom_to_pair] -> [A, P]."""
        A = _dev_f32(inputs[0], self.device)
        Pf = _dev_f32(inputs[1], self.device)
        n_atoms, n_pairs = A.shape[0], Pf.shape[0]
        plan = _pair_plan(inputs[2], inputs[3], n_atoms, self.device)
        if plan.n_pairs != n_pairs:
            raise ValueError("pair_split / atom_to_pair do not match the %d pairs" % n_pairs)

        w, b = self._folded(self.W_AA, self.b_AA, self.AA_bn)
        AA = self._linear(A, w, b, True)
        w, b = self._folded(self.W_PA, self.b_PA, self.PA_bn)
        PA = ops.weave_pair_to_atom(Pf, plan.pair_src, n_atoms, w, b)
        w, b = self._folded(self.W_A, self.b_A, self.A_bn)
        A_out = self._linear(AA, w[:self.n_hidden_AA].contiguous(), b, True, PA, w[self.n_hidden_AA:].contiguous())
        if not self.update_pair:
            return [A_out, Pf]
        Fa = self.n_atom_input_feat
        w, b_ap = self._folded(self.W_AP, self.b_AP, self.AP_bn)
        # Z = [atom->pair block | pair->pair block], each padded to a multiple of four columns so that every
        # row piece the kernels touch is 16-byte addressable (pad columns are exact zeros: zero weights, zero
        # biases, relu(0) = 0).  U = A.W_AP[:Fa] and V = A.W_AP[Fa:] come from ONE atom-level product; the
        # pair kernel does the four-row gather per pair, the pair->pair block is an ordinary product.
        H, H2 = w.shape[1], self.n_hidden_PP
        Hp, H2p = (H + 3) // 4 * 4, (H2 + 3) // 4 * 4
        w_uv = torch.zeros((Fa, 2 * Hp), dtype=torch.float32, device=self.device)
        w_uv[:, :H] = w[:Fa]
        w_uv[:, Hp:Hp + H] = w[Fa:]
        UV = self._linear(A, w_uv, None, False)
        b_ap_p = torch.zeros(Hp, dtype=torch.float32, device=self.device)
        b_ap_p[:H] = b_ap
        Z = torch.empty((n_pairs, Hp + H2p), dtype=torch.float32, device=self.device)
        ops.weave_pair_features(UV[:, :Hp], UV[:, Hp:], b_ap_p, Pf, None, None, plan.a2p, out=Z[:, :Hp])
        w_pp, b_pp = self._folded(self.W_PP, self.b_PP, self.PP_bn)
        w_pp_p = torch.zeros((w_pp.shape[0], H2p), dtype=torch.float32, device=self.device)
        w_pp_p[:, :H2] = w_pp
        b_pp_p = torch.zeros(H2p, dtype=torch.float32, device=self.device)
        b_pp_p[:H2] = b_pp
        ops.seg_gemm([0], [n_pairs], Pf, w_pp_p.reshape(-1), [0], None, None, None, b_pp_p, [0], H2p, False, True,
                     n_pairs, Pf.shape[1], 0, out=Z[:, Hp:])
        w, b = self._folded(self.W_P, self.b_P, self.P_bn)
        w_p = torch.zeros((Hp + H2p, w.shape[1]), dtype=torch.float32, device=self.device)
        w_p[:H] = w[:H]
        w_p[Hp:Hp + H2] = w[H:]
        # the output block is padded to a multiple of four columns too: the product runs on the split-bf16 kernel
        # (which wants 16-byte output quads) and the next layer reads 16-byte addressable pair rows; callers see
        # the (n_pairs, n_pair_output_feat) view
        Ho = w.shape[1]
        Hop = (Ho + 3) // 4 * 4
        w_po = torch.zeros((Hp + H2p, Hop), dtype=torch.float32, device=self.device)
        w_po[:, :Ho] = w_p
        b_po = torch.zeros(Hop, dtype=torch.float32, device=self.device)
        b_po[:Ho] = b
        P_out = self._linear(Z, w_po, b_po, True)[:, :Ho]
        return [A_out, P_out]


class WeaveGather(nn.Module):
    """Molecule fingerprints from atom features: per-molecule sum, after an 11-bin Gaussian
    histogram expansion of every feature when ``gaussian_expand`` (reference layers.py:4432-4648)."""

    def __init__(self, batch_size: int, n_input: int = 128, gaussian_expand: bool = True,
                 compress_post_gaussian_expansion: bool = False, init_: str = 'xavier_uniform_',
                 activation: str = 'tanh', device=None, **kwargs):
        super(WeaveGather, self).__init__(**kwargs)
        self.n_input = n_input
        self.batch_size = batch_size
        self.gaussian_expand = gaussian_expand
        self.compress_post_gaussian_expansion = compress_post_gaussian_expansion
        self.init = init_
        self.activation = activation
        self.device = torch.device("cuda:0") if device is None else torch.device(device)
        if self.compress_post_gaussian_expansion:
            if activation != 'tanh':
                raise GcmiError("WeaveGather: only activation='tanh' has a kernel (got %r)" % activation)
            init = getattr(initializers, self.init)
            self.W = init(torch.empty([self.n_input * 11, self.n_input])).to(self.device)
            self.b = torch.zeros((self.n_input,), device=self.device)
        self.built = True

    def __repr__(self):
        return (f'{self.__class__.__name__}(batch_size:{self.batch_size},n_input:{self.n_input},'
                f'gaussian_expand:{self.gaussian_expand},init:{self.init},activation:{self.activation},'
                f'compress_post_gaussian_expansion:{self.compress_post_gaussian_expansion})')

    def forward(self, inputs: List) -> torch.Tensor:
        """inputs = [atom_features, atom_split] -> one row per molecule present in atom_split."""
        x = _dev_f32(inputs[0], self.device)
        atom_split = _host_i64(inputs[1])
        if atom_split.shape[0] != x.shape[0]:
            raise ValueError("atom_split does not match the %d atoms" % x.shape[0])
        n_mols = int(atom_split.max()) + 1 if atom_split.size else 0
        ptr = _csr_from_sorted(atom_split, n_mols, "atom_split")
        if n_mols and int(np.diff(ptr).min()) == 0:
            raise ValueError("atom_split must name consecutive molecules")  # reference: rows in order of appearance
        out = ops.weave_gather(x, torch.from_numpy(ptr).to(self.device), self.gaussian_expand)
        if self.compress_post_gaussian_expansion:
            W = ops.rowmajor(self.W.detach().to(self.device, torch.float32))
            b = self.b.detach().to(self.device, torch.float32).contiguous()
            out = ops.seg_gemm([0], [n_mols], out, W.reshape(-1), [0], None, None, None, b, [0], W.shape[1], False,
                               False, n_mols, W.shape[0], 0)
            out = ops.tanh_(out)
        return out

    def gaussian_histogram(self, x) -> torch.Tensor:
        """(N, n_feat) -> (N, 11*n_feat): the expansion alone (each row is its own segment)."""
        x = _dev_f32(x, self.device)
        ptr = torch.arange(x.shape[0] + 1, dtype=torch.int32, device=self.device)
        return ops.weave_gather(x, ptr, True)

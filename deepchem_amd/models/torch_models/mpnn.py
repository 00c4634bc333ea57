"""``MPNNModel``: message-passing network of Gilmer et al. with the interface of DeepChem's MPNNModel.

Which reference.  The torch ``MPNNModel`` of the reference delegates to ``dgllife.model.MPNNPredictor``
(deepchem/models/torch_models/mpnn.py:125; dgllife is neither under /root/reference nor installed), so the
in-tree definition of the model is the Keras one: ``MPNNModel`` (deepchem/models/graph_models.py:1045-1247) over
``MessagePassing`` (T rounds of ``EdgeNetwork`` + ``GatedRecurrentUnit``, models/layers.py:3648-3799), a
``Dense(n_hidden)`` atom embedding, ``SetGather`` (set2set, M rounds, :3802-3887), ``Dense(2 n_hidden, relu)`` and
the task head; same constructor arguments, generator contract (``[atom_features, pair_features, atom_split,
atom_to_pair, n_samples]``), outputs and losses.  **Model-level parity is unpinned** (no TensorFlow here); the
sub-layers are pinned by the reference's torch ports and their assets (tests/test_gpu_mpnn.py), the model by an
autograd restatement on torch-CPU (oracle/mpnn_oracle.py: ``MPNNOracle``).

What runs where.  All arithmetic is in libgcmi.so.  EdgeNetwork is re-associated so that the weights meet the data
in ONE atom-level product (per-atom moments ``T = [sum_p pf_pk h_src(p) | sum_p h_src(p)]``, then ``m = T . M^T``;
mpnn_layers.py), and its backward uses the same re-association on the transposed pair list:

    dh_j = sum_k W_k^T (sum_i pf_ijk dm_i) + B^T sum_i dm_i   =   moments over the pairs ENDING in j of dm, times M2^T

so neither pass builds a d x d matrix per pair nor a (K+1) d gradient row per pair.  GRU gates, the set2set
attention (softmax recomputed in the backward) and the LSTM cell have fused elementwise / per-molecule kernels in
both directions (csrc/mpnn.hip); every matrix product runs on the segmented-GEMM kernels.
"""
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from deepchem_amd import _lib, ops
from deepchem_amd.metrics import to_one_hot
from deepchem_amd.models.losses import L2Loss, SoftmaxCrossEntropy
from deepchem_amd.models.torch_models.torch_model import TorchModel
from deepchem_amd.models.torch_models.weave_layers import _csr_from_sorted
from deepchem_amd.ops import _ptr, _stream


# ---------------------------------------------------------------------------------------------- autograd pieces
class MatmulFn(torch.autograd.Function):
    """``x @ W (+ x2 @ W2) + b`` with (in, out) weight layout (the layout of the Keras kernels)."""

    @staticmethod
    def forward(ctx, x, W, b, x2, W2):
        x, W = ops.rowmajor(x), W.contiguous()
        n, k = x.shape
        n_out = W.shape[1]
        dual = x2 is not None
        if dual:
            x2, W2 = ops.rowmajor(x2), W2.contiguous()
        out = ops.seg_gemm([0], [n], x, W.reshape(-1), [0], x2 if dual else None, W2.reshape(-1) if dual else None,
                           [0] if dual else None, None if b is None else b.contiguous(), None if b is None else [0],
                           n_out, False, False, n, k, x2.shape[1] if dual else 0)
        ctx.dual, ctx.has_bias = dual, b is not None
        ctx.params = (W, b, W2)
        ctx.save_for_backward(x, W, x2 if dual else x, W2 if dual else W)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W, x2, W2 = ctx.saved_tensors
        g = ops.rowmajor(g)
        n, n_out = g.shape
        dev = g.device

        def dgrad(Wm):  # g @ Wm^T: Wm (k, n_out) read as an nn.Linear matrix of a layer n_out -> k
            return ops.seg_gemm([0], [n], g, Wm.reshape(-1), [0], None, None, None, None, None, Wm.shape[0], True, False,
                                n, n_out, 0)

        def wgrad(a, Wm, want_bias, pW, pb):
            # (the kernels add into their outputs: with ops.direct_param_grads on, straight into p.grad)
            tw, tb = ops.grad_target(pW), ops.grad_target(pb) if want_bias else None
            dw = tw if tw is not None else torch.zeros_like(Wm)
            db = None
            if want_bias:
                db = tb if tb is not None else torch.zeros(n_out, dtype=torch.float32, device=dev)
            if n > 0:
                ops.seg_gemm_wgrad([0], [n], a, g, dw, [0], db, [0] if want_bias else None, False)
            return (None if tw is not None else dw), (None if tb is not None else db)

        pW, pb, pW2 = ctx.params
        dx = dgrad(W) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = wgrad(x, W, ctx.has_bias, pW, pb)
        dx2 = dW2 = None
        if ctx.dual:
            dx2 = dgrad(W2) if ctx.needs_input_grad[3] else None
            if ctx.needs_input_grad[4]:
                dW2, _ = wgrad(x2, W2, False, pW2, None)
        return dx, dW, db, dx2, dW2


class GruGatesFn(torch.autograd.Function):
    """(zp, rp, h) -> (z = sigmoid(zp), r = sigmoid(rp), hr = h r), in place on zp / rp."""

    @staticmethod
    def forward(ctx, zp, rp, h):
        h = h.contiguous()
        hr = ops.gru_gates_(zp, rp, h)
        ctx.mark_dirty(zp, rp)
        ctx.save_for_backward(zp, rp, h)
        return zp, rp, hr

    @staticmethod
    def backward(ctx, dz, dr_unused, dhr):
        z, r, h = ctx.saved_tensors
        dz = torch.zeros_like(z) if dz is None else dz.contiguous()
        dhr = torch.zeros_like(z) if dhr is None else dhr.contiguous()
        dzp, drp, dh = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        _lib.call("gcmi_gru_gates_bwd", _ptr(z), _ptr(r), _ptr(h), _ptr(dz), _ptr(dhr), _ptr(dzp), _ptr(drp), _ptr(dh),
                  z.numel(), _stream())
        return dzp, drp, dh


class GruOutFn(torch.autograd.Function):
    """(z, hpre, x) -> (1 - z) tanh(hpre) + z x."""

    @staticmethod
    def forward(ctx, z, hpre, x):
        z, hpre, x = z.contiguous(), hpre.contiguous(), x.contiguous()
        ctx.save_for_backward(z, hpre, x)
        return ops.gru_out(z, hpre, x)

    @staticmethod
    def backward(ctx, dout):
        z, hpre, x = ctx.saved_tensors
        dout = dout.contiguous()
        dz, dhpre, dx = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        _lib.call("gcmi_gru_out_bwd", _ptr(z), _ptr(hpre), _ptr(x), _ptr(dout), _ptr(dz), _ptr(dhpre), _ptr(dx),
                  z.numel(), _stream())
        return dz, dhpre, dx


class AttendFn(torch.autograd.Function):
    """One set2set attention step: (x, h) -> q_star = [h | sum_a softmax(<x_a, h>) x_a] per molecule."""

    @staticmethod
    def forward(ctx, x, h, mol_ptr):
        x, h = ops.rowmajor(x), ops.rowmajor(h)
        ctx.save_for_backward(x, h, mol_ptr)
        return ops.set2set_attend(x, mol_ptr, h)

    @staticmethod
    def backward(ctx, dq):
        x, h, mol_ptr = ctx.saved_tensors
        dq = ops.rowmajor(dq)
        dx = torch.empty_like(x)
        dh = torch.empty_like(h)
        n_mols = mol_ptr.numel() - 1
        _lib.call("gcmi_set2set_attend_bwd", _ptr(x), x.stride(0), x.shape[1], _ptr(mol_ptr), n_mols, _ptr(h), h.stride(0),
                  _ptr(dq), dq.stride(0), _ptr(dx), dx.stride(0), 0, _ptr(dh), dh.stride(0), _stream())
        return dx, dh, None


class LstmCellFn(torch.autograd.Function):
    """(z (B, 4H) gate pre-activations in order i, f, o, g; c) -> (h', c')."""

    @staticmethod
    def forward(ctx, z, c):
        z = ops.rowmajor(z)
        c_prev = c.contiguous()
        c_new = c_prev.clone()
        h_new = ops.lstm_cell_(z, c_new)
        ctx.save_for_backward(z, c_prev)
        return h_new, c_new

    @staticmethod
    def backward(ctx, dh, dc):
        z, c_prev = ctx.saved_tensors
        H = c_prev.shape[1]
        dh = torch.zeros_like(c_prev) if dh is None else dh.contiguous()
        dz = torch.empty_like(z)
        dc_prev = torch.empty_like(c_prev)
        _lib.call("gcmi_lstm_cell_bwd", _ptr(z), z.stride(0), H, z.shape[0], _ptr(c_prev), _ptr(dh),
                  _ptr(dc.contiguous()) if dc is not None else None, _ptr(dz), _ptr(dc_prev), _stream())
        return dz, dc_prev


class PairPlan:
    """The pair list of a batch on the device, in both directions: pairs sorted by their first atom (``dst``; the
    generator's order) with the second atoms ``src``, and the same pairs sorted by their second atom for the
    backward pass (with the pair features permuted once).  Built with device ops (bincount / cumsum / a stable
    argsort): sorting 3 x 10^5 pairs on the host cost more than the whole optimizer step."""

    def __init__(self, atom_to_pair, pair_features: torch.Tensor, n_atoms: int, device, mol_ptr=None, max_mol_atoms: int = 0):
        a2p = torch.as_tensor(np.ascontiguousarray(np.asarray(atom_to_pair), np.int64).reshape(-1, 2)) \
            if not torch.is_tensor(atom_to_pair) else atom_to_pair.reshape(-1, 2).to(torch.int64)
        a2p = a2p.to(device, non_blocking=True)
        if a2p.shape[0] != pair_features.shape[0]:
            raise ValueError("atom_to_pair does not match the %d pairs" % pair_features.shape[0])
        dst, src = a2p[:, 0].contiguous(), a2p[:, 1].contiguous()
        if a2p.numel():
            bad = (a2p.min() < 0) | (a2p.max() >= n_atoms) | (dst[1:] < dst[:-1]).any()
            if bool(bad):  # one read-back per batch
                raise ValueError("atom_to_pair must list pairs by ascending first atom, atoms inside [0, %d)" % n_atoms)
        self.n_atoms = n_atoms
        self.pf = pair_features
        self.mol_ptr = mol_ptr  # int32 CSR of the atoms per molecule, or None: the moments kernel stages molecules in LDS
        self.max_mol_atoms = int(max_mol_atoms)  # (0: unknown)

        def csr(ids):
            ptr = torch.zeros(n_atoms + 1, dtype=torch.int64, device=device)
            if ids.numel():
                torch.cumsum(torch.bincount(ids, minlength=n_atoms), 0, out=ptr[1:])
            return ptr.to(torch.int32)
        self.dst_ptr = csr(dst)
        self.src = src.to(torch.int32)
        perm = torch.argsort(src, stable=True)
        self.src_ptr = csr(src)  # bincount does not need the sorted order
        self.dst_of_sorted = dst.index_select(0, perm).to(torch.int32)
        self.pf_t = pair_features.index_select(0, perm).contiguous()


class EdgeMatsFn(torch.autograd.Function):
    """(W (K, d d), b (d d)) -> M (d, (K+1) d) with M[r, k d + c] = W_k[r, c] (k = K: the bias block): the matrix the
    per-atom moments are multiplied with.  Built ONCE per forward, not per message round; its backward folds the
    gradient accumulated over the T rounds back into W's and b's layouts."""

    @staticmethod
    def forward(ctx, W, b, d: int):
        K = W.shape[0]
        blocks = torch.cat([W.reshape(K, d, d), b.reshape(1, d, d)])  # [k][r][c]
        ctx.K, ctx.d = K, d
        return blocks.permute(1, 0, 2).reshape(d, (K + 1) * d).contiguous()

    @staticmethod
    def backward(ctx, dM):
        K, d = ctx.K, ctx.d
        dB = dM.reshape(d, K + 1, d).permute(1, 0, 2)
        return dB[:K].reshape(K, d * d).contiguous(), dB[K].reshape(d * d).contiguous(), None


class EdgeAccum:
    """Per forward pass: M, its transposed twin for the backward (M2[c, k d + r] = W_k[r, c]), and ONE accumulator
    that the T rounds' weight-gradient launches add into; the round that runs its backward last hands it to autograd."""

    def __init__(self, M: torch.Tensor, K: int, d: int):
        self.M, self.K, self.d = M, K, d
        self.M2 = None
        self.dM = None
        self.pending = 0


class EdgeNetworkFn(torch.autograd.Function):
    """m_i = sum_{pairs (i, j)} A(pf_ij) h_j with A(pf) = reshape(pf . W + b, (d, d))
    (models/layers.py:3744-3752), computed as moments + one product, both ways."""

    @staticmethod
    def forward(ctx, h, M, plan: PairPlan, acc: EdgeAccum):
        h = ops.rowmajor(h)
        d, K = acc.d, acc.K
        T = ops.edge_network_moments(h, plan.pf, plan.dst_ptr, plan.src, plan.mol_ptr, plan.max_mol_atoms)
        n = T.shape[0]
        m = ops.seg_gemm([0], [n], T, M.reshape(-1), [0], None, None, None, None, None, d, True, False, n, (K + 1) * d, 0)
        ctx.plan, ctx.acc = plan, acc
        acc.pending += 1
        ctx.save_for_backward(T, M)
        return m

    @staticmethod
    def backward(ctx, dm):
        T, M = ctx.saved_tensors
        plan, acc = ctx.plan, ctx.acc
        K, d = acc.K, acc.d
        dm = ops.rowmajor(dm)
        n = dm.shape[0]
        dh = dM = None
        acc.pending -= 1
        if ctx.needs_input_grad[1]:
            if acc.dM is None:
                acc.dM = torch.zeros((d, (K + 1) * d), dtype=torch.float32, device=dm.device)
            if n > 0:  # dM[r, k d + c] += sum_i dm_ir T_i[k d + c]
                ops.seg_gemm_wgrad([0], [n], T, dm, acc.dM, [0], None, None, True)
            if acc.pending == 0:
                dM, acc.dM = acc.dM, None
        if ctx.needs_input_grad[0]:
            if acc.M2 is None:
                acc.M2 = M.reshape(d, K + 1, d).permute(2, 1, 0).reshape(d, (K + 1) * d).contiguous()
            Tt = ops.edge_network_moments(dm, plan.pf_t, plan.src_ptr, plan.dst_of_sorted, plan.mol_ptr, plan.max_mol_atoms)  # [sum_i pf_ijk dm_i | sum_i dm_i]
            dh = ops.seg_gemm([0], [n], Tt, acc.M2.reshape(-1), [0], None, None, None, None, None, d, True, False, n,
                              (K + 1) * d, 0)
        return dh, dM, None, None


# ---------------------------------------------------------------------------------------------- the network
class _MPNNTorchModel(nn.Module):

    def __init__(self, n_tasks: int, n_atom_feat: int = 70, n_pair_feat: int = 8, n_hidden: int = 100, T: int = 5,
                 M: int = 10, mode: str = "regression", n_classes: int = 2, batch_size: int = 100,
                 uncertainty: bool = False):
        super().__init__()
        self.uncertainty = uncertainty
        if n_atom_feat > n_hidden:
            raise ValueError("Too large initial feature vector")
        if n_hidden > 128 or n_pair_feat > 16:
            raise ValueError("MPNNModel on libgcmi.so supports n_hidden <= 128 and n_pair_feat <= 16")
        self.n_tasks, self.n_classes, self.mode = n_tasks, n_classes, mode
        self.n_atom_feat, self.n_pair_feat, self.n_hidden, self.T, self.M = n_atom_feat, n_pair_feat, n_hidden, T, M
        self.batch_size = batch_size
        d = n_hidden
        glorot = nn.init.xavier_uniform_
        # EdgeNetwork (models/layers.py:3728-3741)
        self.edge_W = nn.Parameter(glorot(torch.empty(n_pair_feat, d * d)))
        self.edge_b = nn.Parameter(torch.zeros(d * d))
        # GatedRecurrentUnit (:3770-3787)
        for name in ("Wz", "Wr", "Wh", "Uz", "Ur", "Uh"):
            setattr(self, "gru_" + name, nn.Parameter(glorot(torch.empty(d, d))))
        for name in ("bz", "br", "bh"):
            setattr(self, "gru_" + name, nn.Parameter(torch.zeros(d)))
        self.atom_embed = nn.Linear(d, d)
        # SetGather (:3836-3848): orthogonal U, forget-gate bias one
        self.set_U = nn.Parameter(nn.init.orthogonal_(torch.empty(2 * d, 4 * d)))
        self.set_b = nn.Parameter(torch.cat([torch.zeros(d), torch.ones(d), torch.zeros(d), torch.zeros(d)]))
        self.dense1 = nn.Linear(2 * d, 2 * d)
        self.head = nn.Linear(2 * d, n_tasks * n_classes if mode == "classification" else n_tasks)
        dense_layers = [self.atom_embed, self.dense1, self.head]
        if uncertainty:  # a second head on the same features predicts the log-variance (graph_models.py:1147-1151)
            self.log_var_head = nn.Linear(2 * d, n_tasks)
            dense_layers.append(self.log_var_head)
        for lin in dense_layers:  # Keras Dense: glorot uniform, zero bias
            glorot(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, inputs) -> List[torch.Tensor]:
        atom_features, pair_features, atom_split, atom_to_pair, n_samples = inputs
        dev = self.edge_W.device
        x = torch.as_tensor(atom_features, dtype=torch.float32, device=dev)
        pf = torch.as_tensor(pair_features, dtype=torch.float32, device=dev).contiguous()
        if not x.is_cuda:
            raise _lib.GcmiError("MPNNModel: inputs must be on the GPU (no CPU path in deepchem_amd)")
        n, d = x.shape[0], self.n_hidden
        if x.shape[1] != self.n_atom_feat or pf.shape[1] != self.n_pair_feat:
            raise ValueError("MPNNModel: feature widths do not match the model")
        split = atom_split.cpu().numpy() if torch.is_tensor(atom_split) else np.asarray(atom_split)
        mol_ptr = torch.from_numpy(_csr_from_sorted(np.asarray(split, np.int64), self.batch_size, "atom_split")).to(dev)  # int32
        biggest = int(np.bincount(np.asarray(split, np.int64)).max()) if len(split) else 0
        plan = PairPlan(atom_to_pair, pf, n, dev, mol_ptr, biggest)
        h = torch.zeros((n, d), dtype=torch.float32, device=dev)
        h[:, :self.n_atom_feat] = x  # zero padding up to n_hidden (MessagePassing.call, :3697-3706)
        edge_M = EdgeMatsFn.apply(self.edge_W, self.edge_b, d)
        edge_acc = EdgeAccum(edge_M, self.n_pair_feat, d)
        for _ in range(self.T):
            m = EdgeNetworkFn.apply(h, edge_M, plan, edge_acc)
            zp = MatmulFn.apply(m, self.gru_Wz, self.gru_bz, h, self.gru_Uz)
            rp = MatmulFn.apply(m, self.gru_Wr, self.gru_br, h, self.gru_Ur)
            z, r, hr = GruGatesFn.apply(zp, rp, h)
            hpre = MatmulFn.apply(m, self.gru_Wh, self.gru_bh, hr, self.gru_Uh)
            # Keras GatedRecurrentUnit.call (models/layers.py:3796-3799): inputs = [out, message] and the carry is
            # z * inputs[0] = z * h (the torch port of the layer, torch_models/layers.py:2912, carries z * message;
            # the stand-alone GatedRecurrentUnit of mpnn_layers.py keeps that, its reference asset pins it)
            h = GruOutFn.apply(z, hpre, h)
        emb = ops.LinearFn.apply(h, self.atom_embed.weight, self.atom_embed.bias, False, False)
        B = self.batch_size
        c = torch.zeros((B, d), dtype=torch.float32, device=dev)
        hs = torch.zeros((B, d), dtype=torch.float32, device=dev)
        q_star = None
        for _ in range(self.M):
            q_star = AttendFn.apply(emb, hs, mol_ptr)
            z4 = MatmulFn.apply(q_star, self.set_U, self.set_b, None, None)
            hs, c = LstmCellFn.apply(z4, c)
        dense1 = ops.LinearFn.apply(q_star, self.dense1.weight, self.dense1.bias, True, False)
        out = ops.LinearFn.apply(dense1, self.head.weight, self.head.bias, False, False)
        n_samples = int(n_samples)
        if self.mode == "classification":
            logits = out.reshape(-1, self.n_tasks, self.n_classes)[0:n_samples]
            return [ops.SoftmaxFn.apply(logits), logits]
        output = out[0:n_samples]
        if self.uncertainty:
            log_var = ops.LinearFn.apply(dense1, self.log_var_head.weight, self.log_var_head.bias, False, False)[0:n_samples]
            return [output, torch.exp(log_var), output, log_var]
        return [output]


class MPNNModel(TorchModel):
    """Message Passing Neural Network (Gilmer et al. 2017) with set2set readout (Vinyals et al. 2015): constructor,
    batches and outputs of deepchem.models.MPNNModel (graph_models.py:1045-1247)."""

    def __init__(self, n_tasks: int, n_atom_feat: int = 70, n_pair_feat: int = 8, n_hidden: int = 100, T: int = 5,
                 M: int = 10, mode: str = "regression", dropout: float = 0.0, n_classes: int = 2,
                 uncertainty: bool = False, batch_size: int = 100, **kwargs):
        if mode not in ['classification', 'regression']:
            raise ValueError("mode must be either 'classification' or 'regression'")
        if uncertainty:
            if mode != "regression":
                raise ValueError("Uncertainty is only supported in regression mode")
            if dropout == 0.0:
                raise ValueError('Dropout must be included to predict uncertainty')
        self.n_tasks, self.n_atom_feat, self.n_pair_feat, self.n_hidden = n_tasks, n_atom_feat, n_pair_feat, n_hidden
        self.T, self.M, self.mode, self.n_classes, self.uncertainty = T, M, mode, n_classes, uncertainty
        model = _MPNNTorchModel(n_tasks, n_atom_feat, n_pair_feat, n_hidden, T, M, mode, n_classes, batch_size,
                                uncertainty)
        if mode == "classification":
            output_types, loss = ['prediction', 'loss'], SoftmaxCrossEntropy()
        elif uncertainty:
            # the Keras model validates `dropout` but builds no Dropout layer (graph_models.py:1110-1170), so every
            # mask of predict_uncertainty gives the same prediction and the reported deviation is the predicted
            # (aleatoric) one; kept as it is
            output_types = ['prediction', 'variance', 'loss', 'loss']

            def loss(outputs, labels, weights):
                diff = outputs[0] - labels[0].reshape(outputs[0].shape)
                losses = torch.square(diff) / torch.exp(outputs[1]) + outputs[1]
                w = weights[0]
                if w.dim() < losses.dim():
                    w = w.reshape(tuple(w.shape) + (1,) * (losses.dim() - w.dim()))
                return torch.mean(losses * w)
        else:
            output_types, loss = ['prediction'], L2Loss()
        super(MPNNModel, self).__init__(model, loss, output_types=output_types, batch_size=batch_size, **kwargs)
        self._flat_step = True  # parameters, gradients and Adam moments in flat buffers (TorchModel._ensure_built)

    def _to_device(self, x):
        # index arrays stay on the host: the pair plan (two CSR directions) is built there
        if not torch.is_tensor(x) and np.asarray(x).dtype.kind in "iu":
            return np.asarray(x)
        return super(MPNNModel, self)._to_device(x)

    def default_generator(self, dataset, epochs: int = 1, mode: str = 'fit', deterministic: bool = True,
                          pad_batches: bool = True):
        """``([atom_features, pair_features, atom_split, atom_to_pair, n_samples], [y], [w])`` per batch
        (graph_models.py:1197-1247): all n x n ordered pairs of every molecule, in meshgrid order."""
        from deepchem_amd.data.datasets import pad_features
        for _ in range(epochs):
            for X_b, y_b, w_b, _ids in dataset.iterbatches(batch_size=self.batch_size, deterministic=deterministic,
                                                           pad_batches=pad_batches):
                n_samples = np.array(X_b.shape[0])
                X_b = pad_features(self.batch_size, X_b)
                if y_b is not None and self.mode == 'classification':
                    y_b = to_one_hot(y_b.flatten(), self.n_classes).reshape(-1, self.n_tasks, self.n_classes)
                atom_feat, pair_feat, atom_split, atom_to_pair = [], [], [], []
                start = 0
                for im, mol in enumerate(X_b):
                    n_atoms = mol.get_num_atoms()
                    atom_split.extend([im] * n_atoms)
                    first = np.repeat(np.arange(n_atoms), n_atoms)
                    second = np.tile(np.arange(n_atoms), n_atoms)
                    atom_to_pair.append(np.stack([first + start, second + start], axis=1))
                    start += n_atoms
                    atom_feat.append(mol.get_atom_features())
                    pair_feat.append(np.reshape(mol.get_pair_features(), (n_atoms * n_atoms, self.n_pair_feat)))
                inputs = [np.concatenate(atom_feat, axis=0), np.concatenate(pair_feat, axis=0), np.array(atom_split),
                          np.concatenate(atom_to_pair, axis=0), n_samples]
                yield (inputs, [y_b], [w_b])

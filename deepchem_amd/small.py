"""SmallBatchEngine: many optimizer steps (or prediction batches) per C call.

At the batch sizes the reference actually trains with (100 molecules by default,
graphconvmodel.py:292; 64 in MolNet's presets) a step moves ~2 000 atom rows: the large-batch kernels
leave the GPU almost empty and the Python loop of ``fit_generator`` (torch_model.py:423-445) costs more
than the arithmetic.  ``gcmi_small_fit`` / ``gcmi_small_predict`` (include/gcmi.h, csrc/smallstep.hip)
run the whole step -- forward, loss, backward, Adam, BatchNorm running statistics -- as 8-12 launches
on 16-row degree tiles, for a whole list of collated batches per call.  This class owns the host side:
the descriptor array, the workspace, and the hand-over of the optimizer state (the flat moment buffers
of ``GcmiAdam``, so a checkpoint taken afterwards is indistinguishable from one written by the
per-batch path).
"""
import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from deepchem_amd import _lib
from deepchem_amd._lib import GcmiModelIO, GcmiSmallBatch
from deepchem_amd.graph import _stream

# batches up to this many atoms run on the small engine (beyond, the streaming kernels win)
SMALL_MAX_ATOMS = 16384


class SmallUnsupported(Exception):
    pass


class SmallBatchEngine:

    def __init__(self, native):
        """``native``: the model's ``deepchem_amd.native.NativeNet`` (flat parameter arena + description)."""
        self.native = native
        self._ws: Optional[torch.Tensor] = None
        self._ws_shape = (0, 0)
        lib = _lib.load()
        if lib.gcmi_small_workspace_floats(ctypes.byref(native.desc), 1, 2) < 0:
            raise SmallUnsupported(lib.gcmi_last_error().decode())

    # ------------------------------------------------------------------ buffers
    def _workspace(self, max_atoms: int, max_mols: int) -> Tuple[torch.Tensor, int, int]:
        a, b = self._ws_shape
        if self._ws is None or max_atoms > a or max_mols > b or self._ws.device != self.native.flat.device:
            a = max(int(max_atoms * 1.25) + 64, a)
            b = max(int(max_mols * 1.25) + 2, b)
            need = int(_lib.load().gcmi_small_workspace_floats(ctypes.byref(self.native.desc), a, b))
            if need < 0:
                raise SmallUnsupported("gcmi_small_workspace_floats rejected the model description")
            self._ws = torch.empty(need + 16, dtype=torch.float32, device=self.native.flat.device)
            self._ws_shape = (a, b)
        return self._ws, self._ws_shape[0], self._ws_shape[1]

    def _io(self, ws: torch.Tensor) -> GcmiModelIO:
        io = GcmiModelIO()
        io.d_workspace = ws.data_ptr()
        d = self.native.desc
        if d.batch_norm:
            for i in range(self.native.n_layers + 1):
                bn = self.native.module.batch_norms[i]
                io.d_bn_running_mean[i] = bn.running_mean.data_ptr()
                io.d_bn_running_var[i] = bn.running_var.data_ptr()
                io.d_bn_batches_tracked[i] = bn.num_batches_tracked.data_ptr()
        return io

    @staticmethod
    def describe(batch, labels=None, weights=None, n_rows: Optional[int] = None) -> GcmiSmallBatch:
        """A ``DeviceBatch`` (+ device labels / weights) as the struct the C side reads."""
        g = batch.graph
        if g.mol_runs is None:
            raise ValueError("the batch has no readout plan (set_mols)")
        sb = GcmiSmallBatch()
        ctypes.memmove(ctypes.byref(sb.graph), ctypes.byref(g.c), ctypes.sizeof(g.c))
        x = batch.atom_features
        sb.d_atom_features = x.data_ptr() if x.numel() else None
        sb.ld_features = int(x.stride(0)) if x.shape[0] > 1 else int(x.shape[1])
        sb.d_labels = labels.data_ptr() if labels is not None else None
        sb.d_weights = weights.data_ptr() if weights is not None else None
        sb.n_rows = int(batch.n_samples if n_rows is None else n_rows)
        return sb

    # ------------------------------------------------------------------ calls
    # gcmi_grad_sync_fn (include/gcmi.h): int (*)(void* ctx, float* d_grad_range, int64_t n_floats, void* stream)
    _SYNC_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)

    def fit(self, descs: Sequence[GcmiSmallBatch], optimizer, max_atoms: int, max_mols: int,
            grad_sync=None) -> torch.Tensor:
        """One Adam step per descriptor, in order.  Returns the per-step losses (device, float32).

        ``grad_sync`` (data parallel, ``deepchem_amd.dist.FlatGradAllReduce``): its ``reduce_flat`` is called once
        per step on the trained range of the gradient arena, between the backward launches and the Adam launch of
        that step (``gcmi_small_fit_dp``); torch.distributed orders the collective on the current stream, which is the
        stream the library launches on."""
        nat = self.native
        n = len(descs)
        losses = torch.empty(n, dtype=torch.float32, device=nat.flat.device)
        if n == 0:
            return losses
        if optimizer._flat is None or optimizer._flat["p"].data_ptr() != nat.flat.data_ptr():
            optimizer.attach_flat(nat.flat, nat.grad_flat, nat._slices)
        d = nat.desc
        L = nat.n_layers
        lo = 0 if d.grad_mode == 1 else (d.off_bn_gamma[L - 1] if d.batch_norm else d.off_dense_w)
        hi = d.n_params
        f = optimizer._flat
        if f.get("range") != (lo, hi):
            optimizer._setup_flat_range(lo, hi)
        group = optimizer.param_groups[0]
        beta1, beta2 = group["betas"]
        first_step = int(f["step_t"].item()) + 1
        arr = (GcmiSmallBatch * n)(*descs)
        ws, a, b = self._workspace(max_atoms, max_mols)
        io = self._io(ws)
        glo, ghi = ctypes.c_int64(0), ctypes.c_int64(0)
        if grad_sync is None:
            _lib.call("gcmi_small_fit", ctypes.byref(d), ctypes.c_void_p(nat.flat.data_ptr()),
                      ctypes.c_void_p(nat.grad_flat.data_ptr()), ctypes.c_void_p(f["m"].data_ptr()),
                      ctypes.c_void_p(f["v"].data_ptr()), ctypes.byref(io), ctypes.cast(arr, ctypes.c_void_p), n, a, b,
                      float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), first_step,
                      ctypes.c_void_p(losses.data_ptr()), ctypes.byref(glo), ctypes.byref(ghi), _stream())
        else:
            bucket = nat.grad_flat[lo:hi]
            failure = []

            def _sync(ctx, ptr, count, stream):
                try:
                    if ptr != bucket.data_ptr() or count != bucket.numel():
                        raise RuntimeError("gradient range mismatch")
                    grad_sync.reduce_flat(bucket)
                    return 0
                except BaseException as e:  # never unwind through the C frames
                    failure.append(e)
                    return 1

            cb = self._SYNC_FN(_sync)
            try:
                _lib.call("gcmi_small_fit_dp", ctypes.byref(d), ctypes.c_void_p(nat.flat.data_ptr()),
                          ctypes.c_void_p(nat.grad_flat.data_ptr()), ctypes.c_void_p(f["m"].data_ptr()),
                          ctypes.c_void_p(f["v"].data_ptr()), ctypes.byref(io), ctypes.cast(arr, ctypes.c_void_p), n, a, b,
                          float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), first_step,
                          ctypes.c_void_p(losses.data_ptr()), ctypes.byref(glo), ctypes.byref(ghi),
                          ctypes.cast(cb, ctypes.c_void_p), None, _stream())
            except _lib.GcmiError:
                if failure:
                    raise failure[0]
                raise
        assert (glo.value, ghi.value) == (lo, hi)
        f["step_t"] += n
        nat.grad_range = (lo, hi)
        return losses

    def predict(self, descs: Sequence[GcmiSmallBatch], max_atoms: int, max_mols: int) -> None:
        """Eval-mode forward of every descriptor into the output pointers it names."""
        n = len(descs)
        if n == 0:
            return
        arr = (GcmiSmallBatch * n)(*descs)
        ws, a, b = self._workspace(max_atoms, max_mols)
        io = self._io(ws)
        _lib.call("gcmi_small_predict", ctypes.byref(self.native.desc), ctypes.c_void_p(self.native.flat.data_ptr()),
                  ctypes.byref(io), ctypes.cast(arr, ctypes.c_void_p), n, a, b, _stream())


# ---------------------------------------------------------------------------------------------- chunks of batches
import collections  # noqa: E402

import numpy as np  # noqa: E402

from deepchem_amd._lib import GcmiGraph  # noqa: E402


class Chunk:
    """The batches of one C call, collated into one device arena."""

    def __init__(self):
        self.n_batches = 0
        self.max_atoms = 0
        self.descs = None      # (GcmiSmallBatch * n)
        self.keep = []         # device tensors the descriptors point into
        self.symmetric = True
        self.n_real: List[int] = []


class ChunkCollator:
    """index batches -> ``Chunk``: ``gcmi_collate_batches`` into a pinned arena (threads over batches), one
    H2D copy, atom codes expanded by one launch, descriptors bound natively (``gcmi_small_bind``)."""

    def __init__(self, packed, device: torch.device, mols_out: int, max_deg: int = 10):
        from deepchem_amd.data.collate import PinnedRing
        self.packed = packed
        self.device = device
        self.mols_out = int(mols_out)
        self.max_deg = max_deg
        self.coded = getattr(packed, "atom_codes", None) is not None
        self.n_feat = 2 if self.coded else packed.n_feat
        self.ld = 2 if self.coded else (self.n_feat + 3) // 4 * 4
        self.feats = packed.atom_codes.view(np.float32) if self.coded else \
            np.ascontiguousarray(packed.atom_features, np.float32)
        self.atom_ptr = np.ascontiguousarray(packed.atom_ptr, np.int64)
        self.adj_ptr = np.ascontiguousarray(packed.adj_ptr, np.int64)
        self.adj_idx = np.ascontiguousarray(packed.adj_idx, np.int32)
        self.ring = PinnedRing(6)  # a worker may be two chunks ahead of the copies

    def collate_host(self, idx_batches: Sequence[np.ndarray], n_real: Sequence[int]):
        """The CPU half (may run on a worker thread: the native call releases the GIL): index batches -> one
        pinned arena + per-batch graph structs and offsets."""
        lib = _lib.load()
        n = len(idx_batches)
        sel = np.ascontiguousarray(np.concatenate(idx_batches), np.int64)
        if sel.size and (sel.min() < 0 or sel.max() >= self.packed.n_mols):
            raise IndexError("molecule index outside the set")
        batch_ptr = np.zeros(n + 1, np.int64)
        np.cumsum([len(i) for i in idx_batches], out=batch_ptr[1:])
        parts = np.zeros((n, 5), np.int64)
        counts = np.zeros((n, 3), np.int64)
        words = int(lib.gcmi_collate_batches_layout(self.atom_ptr.ctypes.data, self.adj_ptr.ctypes.data, sel.ctypes.data,
                                                    batch_ptr.ctypes.data, n, self.ld, self.max_deg, self.mols_out,
                                                    parts.ctypes.data, counts.ctypes.data))
        if words < 0:
            _lib.check(words, "gcmi_collate_batches_layout")
        # the molecule indices ride behind the arena (int64 pairs of words): one pinned H2D copy carries everything a
        # chunk needs -- a separate copy from pageable memory would block the host until the stream has drained
        total = words + 2 * int(sel.shape[0]) + 4
        arena = self.ring.get(total)[:total]
        slot = self.ring._last
        arena[words:].numpy().view(np.int64)[:sel.shape[0]] = sel
        graphs = (GcmiGraph * n)()
        sym = np.ones(n, np.int32)
        _lib.call("gcmi_collate_batches", self.feats.ctypes.data, self.n_feat, self.atom_ptr.ctypes.data,
                  self.adj_ptr.ctypes.data, self.adj_idx.ctypes.data, sel.ctypes.data, batch_ptr.ctypes.data, n,
                  self.max_deg, self.ld, self.mols_out, arena.data_ptr(), parts.ctypes.data, counts.ctypes.data,
                  ctypes.cast(graphs, ctypes.c_void_p), sym.ctypes.data, 0)
        return dict(n=n, sel=sel, words=words, parts=parts, counts=counts, arena=arena, slot=slot, graphs=graphs,
                    sym=sym, n_real=[int(r) for r in n_real])

    def to_device(self, h) -> Chunk:
        """The stream half (consumer thread): one H2D copy, atom codes expanded by one launch."""
        n, counts = h["n"], h["counts"]
        dev_arena = h["arena"].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.ring.events[h["slot"]] = ev  # the pinned buffer may be refilled once this copy has run
        ch = Chunk()
        ch.n_batches = n
        ch.max_atoms = int(counts[:, 0].max()) if n else 0
        ch.symmetric = bool(h["sym"].all())
        ch.n_real = h["n_real"]
        ch.sel = h["sel"]
        ch.sel_dev = dev_arena[h["words"]:].view(torch.int64)[:h["sel"].shape[0]]
        rows = int(counts[-1, 2] + (counts[-1, 0] + 1) // 2 * 2) if n else 0
        if self.coded:
            from deepchem_amd import ops
            feats = ops.expand_atom_codes(dev_arena[:rows * 2].view(torch.uint8).view(rows, 8), 76) if rows else \
                torch.zeros((0, 76), dtype=torch.float32, device=self.device)
            ld = 76
        else:
            feats, ld = dev_arena, self.ld
        ch.keep = [dev_arena, feats]
        ch._bind = (h["graphs"], h["parts"], counts, dev_arena, feats, ld)
        return ch

    def collate(self, idx_batches: Sequence[np.ndarray], n_real: Sequence[int]) -> Chunk:
        return self.to_device(self.collate_host(idx_batches, n_real))

    def bind(self, ch: Chunk, n_rows: Sequence[int], labels=None, label_stride=0, weights=None, weight_stride=0,
             logits=None, probs=None, logit_stride=0, fp=None, fp_stride=0) -> None:
        graphs, parts, counts, dev_arena, feats, ld = ch._bind
        n = ch.n_batches
        ch.descs = (GcmiSmallBatch * n)()
        nr = np.ascontiguousarray(n_rows, np.int64)

        def ptr(t):
            return ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _lib.call("gcmi_small_bind", ctypes.cast(ch.descs, ctypes.c_void_p), ctypes.cast(graphs, ctypes.c_void_p),
                  parts.ctypes.data, counts.ctypes.data, n, ptr(dev_arena), ptr(feats), ld, self.mols_out,
                  nr.ctypes.data, ptr(labels), label_stride, ptr(weights), weight_stride, ptr(logits), ptr(probs),
                  logit_stride, ptr(fp), fp_stride)
        ch.keep += [t for t in (labels, weights, logits, probs, fp) if t is not None]


class HeldChunks:
    """Chunks stay referenced until an event recorded after their last launch has passed (their device buffers
    were filled on this stream and are read by kernels still queued on it)."""

    def __init__(self, device):
        self.device = device
        self.q = collections.deque()

    def hold(self, ch: Chunk) -> None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.q.append((ev, ch))
        while len(self.q) > 2 and self.q[0][0].query():
            self.q.popleft()
        while len(self.q) > 6:  # bound the memory: wait for the oldest
            self.q[0][0].synchronize()
            self.q.popleft()

    def drain(self) -> None:
        if self.q:
            self.q[-1][0].synchronize()
        self.q.clear()


def chunks_ahead(collator: ChunkCollator, index_batches, chunk_batches: int, cut_after=None, depth: int = 2):
    """Iterate host-collated chunks (``collate_host`` results) produced by a worker thread ``depth`` chunks ahead of the
    consumer.  ``cut_after(k)``: True when the k-th batch of the whole iteration (1-based) must end its chunk
    (checkpoint steps)."""
    import queue
    import threading
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    stop = threading.Event()

    def put(item) -> bool:
        while not stop.is_set():
            try:
                q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            buf_idx, buf_real, k = [], [], 0
            limit = min(chunk_batches, 8)  # short first chunks: the GPU starts after 8 batches are collated, not 48
            for idx, n_real in index_batches:
                if stop.is_set():
                    return
                k += 1
                buf_idx.append(idx)
                buf_real.append(int(n_real))
                cut = cut_after is not None and cut_after(k)
                if len(buf_idx) >= limit or cut:
                    if not put((collator.collate_host(buf_idx, buf_real), cut)):
                        return
                    buf_idx, buf_real = [], []
                    limit = min(chunk_batches, 2 * limit)
            if buf_idx:
                if not put((collator.collate_host(buf_idx, buf_real), False)):
                    return
            put(None)
        except BaseException as e:  # surface in the consumer
            put(e)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        th.join(timeout=5.0)

"""Kernel micro-benchmarks on the Tox21 degree mix (R1 of SURVEY.md 8d): every hot kernel
alone, HIP-event timed, with its algorithmic GB/s or TFLOP/s.  Development tool:
    python tools/kbench.py [--mols 65536] [--only wgrad,seg_gemm] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from deepchem_amd import ops  # noqa: E402
from deepchem_amd._lib import K_SEG_GEMM, K_WGRAD  # noqa: E402
from deepchem_amd.data.collate import collate_to_device  # noqa: E402
from deepchem_amd.utils.synthetic import synthetic_molecules  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def timeit_kernel(fn, iters, kernel_id):
    """Device time of the library's own launches (HIP events around them on the launch stream): the Python
    wrappers of the product entry points cost more host time per call than the kernels run, so wall-clock
    events around a loop of calls measure the host."""
    from deepchem_amd import ops
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ops.timing_enable(kernel_id, True)
    ops.timing_read(kernel_id, reset=True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    n, ms = ops.timing_read(kernel_id, reset=True)
    ops.timing_enable(kernel_id, False)
    return ms * 1e3 / max(iters, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--win-cap", type=int, default=None)
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    dev = torch.device("cuda:0")
    packed = synthetic_molecules(args.mols, seed=1)
    b = collate_to_device(packed, None, dev, **({} if args.win_cap is None else {"win_cap": args.win_cap}))
    print("windows:", b.graph.c.n_win, "big:", b.graph.c.n_win_big, "alloc:", b.graph.c.win_alloc,
          b.graph.c.win_alloc_big, flush=True)
    g = b.graph
    g.ensure_rev_pos()
    N, E, B = g.n_atoms, g.n_edges, g.n_mols
    res = {"n_atoms": N, "n_edges": E, "n_mols": B}
    gen = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape):
        return torch.randn(shape, generator=gen).to(dev)

    def want(name):
        return not only or any(name.startswith(o) for o in only)

    def rec(name, us, bytes_=None, flops=None):
        d = {"us": round(us, 1)}
        if bytes_ is not None:
            d["GBps"] = round(bytes_ / us / 1e3, 1)
        if flops is not None:
            d["TFLOPs"] = round(flops / us / 1e6, 2)
        res[name] = d
        print(name, d, flush=True)

    sb, se = list(g.seg_begin), list(g.seg_end)
    for F in (64, 76, 128):
        x = rnd(N, F)
        if want("gather_sum"):
            rec("gather_sum_F%d" % F, timeit(lambda: ops.gather_sum(g, x), args.iters), E * (4 * F + 4) + N * 4 * F)
        if want("gather_max") and F != 76:
            sc, sh = rnd(F), rnd(F)
            rec("gather_max_bn_F%d" % F, timeit(lambda: ops.gather_max(g, x, sc, sh), args.iters),
                E * (4 * F + 4) + 2 * N * 4 * F + N * F)
            out, arg = ops.gather_max(g, x, sc, sh)
            rec("gather_max_bwd_F%d" % F, timeit(lambda: ops.gather_max_bwd(g, x, arg), args.iters),
                E * (5 * F + 5) + 2 * N * 4 * F + N * F)
        if want("readout") and F == 128:
            rec("readout_F128", timeit(lambda: ops.readout(g, x, B, tanh=True), args.iters), N * (4 * F + 4) + B * 8 * F)
            out, arg = ops.readout(g, x, B, tanh=True)
            dout = rnd(B, 2 * F)
            rec("readout_bwd_F128", timeit(lambda: ops.readout_bwd(g, dout, out, arg, True), args.iters),
                N * 4 * F + N * 4 + B * 20 * F)
        if want("bn") and F != 76:
            gam, bet, rm, rv = rnd(F), rnd(F), torch.zeros(F, device=dev), torch.ones(F, device=dev)
            rec("bn_stats_F%d" % F, timeit(lambda: ops.bn_stats(x, gam, bet, rm, rv, 1e-3, 0.99), args.iters), N * 4 * F)
            mean, invstd, _, _ = ops.bn_stats(x, gam, bet, rm, rv, 1e-3, 0.99)
            dy = rnd(N, F)
            rec("bn_bwd_F%d" % F, timeit(lambda: ops.bn_bwd(dy, x, gam, mean, invstd, True, True), args.iters),
                5 * N * 4 * F)
    if want("seg_gemm"):
        for (k1, k2, n_out, trans, label) in ((75, 75, 64, False, "graphconv0_fwd"), (64, 64, 64, False, "graphconv1_fwd"),
                                             (64, 0, 64, True, "graphconv1_dgrad"), (64, 0, 128, True, "dense_fwd"),
                                             (128, 0, 64, False, "dense_dgrad")):
            ld1 = int(os.environ.get("KB_LD75", "76")) if k1 == 75 else k1  # KB_LD75: row stride of the 75-column operands
            a1 = rnd(N, ld1)[:, :k1]
            a2 = rnd(N, ld1)[:, :k2] if k2 else None
            w = rnd(21 * max(k1, 1) * n_out)
            off1 = [-1] + [(2 * (d - 1)) * k1 * n_out for d in range(1, 11)]
            off2 = [20 * k1 * n_out] + [(2 * (d - 1) + 1) * k1 * n_out for d in range(1, 11)]
            bias = rnd(11 * n_out)
            boff = [d * n_out for d in range(11)]
            if k2 == 0:
                off1 = [0] * 11
            fn = lambda: ops.seg_gemm(sb, se, a1, w, off1, a2, w if k2 else None, off2 if k2 else None, bias, boff,
                                      n_out, trans, True, N, k1, k2)
            rec("seg_gemm_" + label, timeit_kernel(fn, args.iters, K_SEG_GEMM), 4.0 * N * (k1 + k2 + n_out),
                2.0 * N * (k1 + k2) * n_out)
        fp = rnd(B, 256)
        wh = rnd(24 * 256)
        rec("seg_gemm_head_fwd", timeit(lambda: ops.seg_gemm([0], [B], fp, wh, [0], None, None, None, None, None, 24,
                                                             True, False, B, 256, 0), args.iters), None, 2.0 * B * 256 * 24)
    if want("wgrad"):
        for (k, n, trans, segs, label) in ((75, 64, False, True, "graphconv0"), (64, 64, False, True, "graphconv1"),
                                           (64, 128, True, False, "dense"), (256, 24, True, False, "head")):
            rows = N if label != "head" else B
            ld = 76 if k == 75 else k
            a = rnd(rows, ld)[:, :k]
            gg = rnd(rows, n)
            dw = torch.zeros(21 * k * n, device=dev)
            db = torch.zeros(11 * n, device=dev)
            if segs:
                fn = lambda: ops.seg_gemm_wgrad(sb, se, a, gg, dw, [0] + [(2 * (d - 1)) * k * n for d in range(1, 11)], db,
                                                [d * n for d in range(11)], trans)
            else:
                fn = lambda: ops.seg_gemm_wgrad([0], [rows], a, gg, dw, [0], db, [0], trans)
            rec("wgrad_" + label, timeit_kernel(fn, args.iters, K_WGRAD), rows * 4 * (k + n), 2.0 * rows * k * n)
    if want("mfma_peak"):
        import ctypes
        from deepchem_amd import _lib
        from deepchem_amd.graph import _stream
        o = torch.zeros(4, device=dev)
        for blocks in (256, 1024, 2048):
            iters = 2000
            fn = lambda: _lib.call("gcmi_diag_mfma_peak", blocks, iters, ctypes.c_void_p(o.data_ptr()), _stream())
            us = timeit(fn, 5)
            rec("mfma_peak_blocks%d" % blocks, us, None, blocks * 4 * iters * 32 * 4096.0)
    if want("adam"):
        p, gr, m, v = rnd(204504), rnd(204504), torch.zeros(204504, device=dev), torch.zeros(204504, device=dev)
        rec("adam_flat_204k", timeit(lambda: ops.adam_step_(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 3), args.iters), 204504 * 28)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

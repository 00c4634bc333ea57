"""Micro-benchmarks of the Weave and message-passing rows on an HIV-like synthetic
batch (25.5 atoms/mol, all ordered pairs incl. self pairs): every new kernel alone and the layers end
to end, HIP-event timed, with algorithmic GB/s, and the same layers on the CPU oracle (torch CPU, the
reference's op sequence) on a smaller sample.

    python tools/kbench_weave.py [--mols 2048] [--iters 10] [--cpu-mols 64]
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
from deepchem_amd.models.torch_models.layers import (EdgeNetwork, GatedRecurrentUnit, SetGather, WeaveGather,  # noqa: E402
                                                     WeaveLayer)
from oracle import mpnn_oracle as MO  # noqa: E402
from oracle import weave_oracle as WO  # noqa: E402


def hiv_like(n_mols, seed, fa=75, fp=14, mean_atoms=25.5, max_atoms=120):
    rng = np.random.RandomState(seed)
    sizes = np.clip(np.round(rng.lognormal(np.log(mean_atoms), 0.45, n_mols)), 2, max_atoms).astype(int)
    mols = []
    for n in sizes:
        src, dst = np.nonzero(np.ones((n, n), bool))
        mols.append((rng.standard_normal((n, fa)).astype(np.float32) * 0.5,
                     rng.standard_normal((n * n, fp)).astype(np.float32) * 0.5, np.stack([src, dst]).astype(np.int64)))
    return mols


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu-mols", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    mols = hiv_like(args.mols, 0)
    atom_feat, pair_feat, pair_split, atom_split, a2p = WO.weave_batch(mols)
    N, P = atom_feat.shape[0], pair_feat.shape[0]
    res = {"n_mols": args.mols, "n_atoms": int(N), "n_pairs": int(P)}
    A, Pf = torch.from_numpy(atom_feat).to(dev), torch.from_numpy(pair_feat).to(dev)

    def rec(name, us, bytes_=None, extra=None):
        d = {"us": round(us, 1)}
        if bytes_ is not None:
            d["GBps"] = round(bytes_ / us / 1e3, 1)
        if extra:
            d.update(extra)
        res[name] = d
        print(name, d, flush=True)

    # ---- Weave: kernels alone
    layer = WeaveLayer()
    H = 50
    psrc = torch.from_numpy(pair_split.astype(np.int32)).to(dev)
    w_pa, b_pa = layer.W_PA.contiguous(), layer.b_PA
    rec("weave_pair_to_atom", timeit(lambda: ops.weave_pair_to_atom(Pf, psrc, N, w_pa, b_pa), args.iters),
        P * 14 * 4 + N * H * 4)
    UV = torch.randn(N, 104, device=dev)  # the layer's layout: both halves padded to 52 columns
    a2p_d = torch.from_numpy(a2p.astype(np.int32)).to(dev).contiguous().view(-1)
    w_pp, b_pp = layer.W_PP.contiguous(), layer.b_PP
    Z = torch.empty(P, 104, device=dev)
    b52 = torch.zeros(52, device=dev)
    rec("weave_pair_features_atom_block",
        timeit(lambda: ops.weave_pair_features(UV[:, :52], UV[:, 52:], b52, Pf, None, None, a2p_d, out=Z[:, :52]), args.iters),
        P * (8 + 52 * 4) + P * 4 * 52 * 4)  # ids + output row + 4 gathered U/V rows (on-die)
    w52 = torch.zeros(14, 52, device=dev)
    w52[:, :50] = w_pp
    rec("weave_pair_features_pair_block",
        timeit(lambda: ops.seg_gemm([0], [P], Pf, w52.reshape(-1), [0], None, None, None, b52, [0], 52, False, True, P, 14,
                                    0, out=Z[:, 52:]), args.iters), P * (14 * 4 + 52 * 4))
    x128 = torch.tanh(torch.randn(N, 128, device=dev))
    mptr = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(atom_split, minlength=args.mols))]).astype(np.int32)).to(dev)
    rec("weave_gather_gaussian", timeit(lambda: ops.weave_gather(x128, mptr, True), args.iters),
        N * 128 * 4 + args.mols * 128 * 11 * 4)
    # ---- Weave: layers end to end
    inputs = [A, Pf, pair_split, a2p]
    rec("WeaveLayer_forward", timeit(lambda: layer(inputs), max(3, args.iters // 3)))
    gather = WeaveGather(args.mols, 128)
    rec("WeaveGather_forward", timeit(lambda: gather([x128, atom_split]), max(3, args.iters // 3)))

    # ---- message passing
    d = 100
    h = torch.randn(N, d, device=dev) * 0.3
    edge, gru = EdgeNetwork(14, d), GatedRecurrentUnit(d)
    rec("EdgeNetwork_forward", timeit(lambda: edge([Pf, h, a2p]), max(3, args.iters // 3)), None,
        {"reference_intermediate_GB": round(P * d * d * 4 / 1e9, 2)})
    msg = edge([Pf, h, a2p])
    rec("GatedRecurrentUnit_forward", timeit(lambda: gru([h, msg]), args.iters), 9 * N * d * 4)
    sg = SetGather(6, args.mols, d)
    rec("SetGather_forward_M6", timeit(lambda: sg([h, atom_split]), max(3, args.iters // 3)), 6 * 2 * N * d * 4)

    # ---- CPU oracle (the reference's op sequence on torch CPU) on a smaller sample
    cm = hiv_like(args.cpu_mols, 1)
    ca, cp, cps, cas, ca2p = WO.weave_batch(cm)
    p = {k: getattr(layer, k).cpu() for k in ("W_AA", "b_AA", "W_PA", "b_PA", "W_A", "b_A", "W_AP", "b_AP", "W_PP", "b_PP",
                                              "W_P", "b_P")}
    t0 = time.perf_counter()
    WO.weave_layer(ca, cp, cps, ca2p, p, None, True)
    t_w = time.perf_counter() - t0
    hc = torch.randn(ca.shape[0], d) * 0.3
    t0 = time.perf_counter()
    MO.edge_network(cp, hc, ca2p, edge.W.cpu(), edge.b.cpu())
    t_e = time.perf_counter() - t0
    res["cpu_oracle"] = {"mols": args.cpu_mols, "pairs": int(cp.shape[0]), "threads": torch.get_num_threads(),
                         "WeaveLayer_forward_us_per_mol": round(t_w / args.cpu_mols * 1e6, 1),
                         "EdgeNetwork_forward_us_per_mol": round(t_e / args.cpu_mols * 1e6, 1)}
    res["gpu_us_per_mol"] = {"WeaveLayer_forward": round(res["WeaveLayer_forward"]["us"] / args.mols, 3),
                             "EdgeNetwork_forward": round(res["EdgeNetwork_forward"]["us"] / args.mols, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

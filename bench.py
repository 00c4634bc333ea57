"""bench.py -- molecules/s of the GraphConvModel training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``
(RANK / LOCAL_RANK / WORLD_SIZE from the environment), or -- when WORLD_SIZE is not set -- this process
starts that launcher itself as a CHILD, before anything has touched the GPU, and exits with its code.
A WORLD_SIZE that disagrees with --gpus, or fewer GPUs than ranks, is an error, never a silent 1-rank run.

One STEP = one optimizer step of ``GraphConvModel`` (forward, loss, backward,
gradient all-reduce when N > 1, Adam) over one collated batch of Tox21-like
synthetic molecules that is already resident in HBM when the timed region
starts.  ``value`` = molecules processed by all ranks / wall time of K steps
(max over ranks, barrier + synchronize on both sides).

The JSON line also carries
  roofline     -- the GraphConv gather-sum kernel: algorithmic bytes
                  (E*(4F+4) + N*4F per forward launch) / its
                  average duration measured with HIP events on the launch
                  stream during the timed steps, against HBM peak 8 TB/s;
  cpu_baseline -- the oracle (CPU restatement of the reference, reference-
                  faithful op sequence, reference gradient semantics) timed on
                  this box's host cores on a bounded sample, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); measured float4 copy: 6290


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="molecules per GPU per step (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch molecules per GPU; strong: --global-batch molecules split over the GPUs")
    ap.add_argument("--global-batch", type=int, default=65536, help="molecules per step over all GPUs (strong scaling)")
    ap.add_argument("--gemm-mode", default="fast", choices=["fast", "exact"],
                    help="arithmetic of the matrix products (deepchem_amd.set_gemm_mode)")
    ap.add_argument("--tasks", type=int, default=12)
    ap.add_argument("--grad-mode", default="full", choices=["full", "reference"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fit-pipeline", type=int, default=1,
                    help="also time end-to-end fit() (collation + H2D + step) on a PackedDataset")
    ap.add_argument("--profile-only", action="store_true",
                    help="only the large-batch steps (for rocprofv3 passes): no small-batch, fit or CPU legs")
    ap.add_argument("--small-batch", type=int, default=100,
                    help="also report mol/s at the reference's default batch size (0 = skip)")
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16", "bf16+grads"],
                    help="activation storage of the timed step (the headline line is fp32; the default run adds a "
                         "bf16-storage line item beside it; --storage bf16 --profile-only is for the rocprofv3 passes)")
    args = ap.parse_args()
    if args.profile_only:
        args.small_batch, args.fit_pipeline, args.no_cpu_baseline = 0, 0, True
    return args


def gather_sum_bytes(n_atoms, n_edges, n_deg0, feats, accumulate):
    b = n_edges * (4 * feats + 4) + (n_atoms - n_deg0) * 4 * feats
    if accumulate:
        b += (n_atoms - n_deg0) * 4 * feats  # read-modify-write of the destination
    else:
        b += n_deg0 * 4 * feats  # zero rows of lone atoms
    return b


def _pmc_tag(name):
    """Which storage a committed PMC file (profiles/r*_pmc_traffic*.json) was taken in, by its name."""
    return "bf16+grads" if "bf16g" in name else "bf16" if "bf16" in name else "fp32"


def measured_traffic(storage="fp32"):
    """HBM bytes per gather-sum launch from the newest committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json:
    FETCH_SIZE x2 + WRITE_SIZE per the gfx950 corrections of MI355X_MICROARCH.md, separate passes, tools/pmc_passes.sh
    over THIS command with --profile-only) -- counters cannot be read from inside the process.  Returns
    (bytes per launch, file name) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")))
    files = [f for f in files if "R1" not in os.path.basename(f) and _pmc_tag(os.path.basename(f)) == storage]
    measured_traffic.step_bytes = None
    for path in reversed(files):
        try:
            with open(path) as f:
                kernels = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        tot, n = 0, 0
        for name, rec in kernels.items():
            if ("SumOp<false>" in name or "SumOpH" in name or "SumOpFH" in name) and "hbm_bytes_per_launch" in rec:  # the forward gather-sum launches
                k = rec.get("fetch_launches", 1)
                tot += rec["hbm_bytes_per_launch"] * k
                n += k
        if n:
            # the whole step: every kernel's bytes per launch x its launches, over the profiled steps (the readout
            # runs once per step)
            steps = max((rec.get("fetch_launches", 0) for name, rec in kernels.items() if "readout_fwd" in name),
                        default=0)
            step_bytes = None
            if steps:
                step_bytes = int(sum(rec.get("hbm_bytes_per_launch", 0) *
                                     max(rec.get("fetch_launches", 0), rec.get("write_launches", 0))
                                     for rec in kernels.values()) / steps)
            measured_traffic.step_bytes = step_bytes
            return int(tot / n), os.path.basename(path)
    return None, None


# kernel families of the event timers (include/gcmi.h GCMI_K_*) by kernel name, for the PMC file
FAMILY_PATTERNS = {
    "gather_sum": ("SumOp<false>", "SumOpFH", "SumOpH"),
    "gather_max": ("MaxOp<", "MaxOpH<"),
    "gather_max_bwd": ("MaxBwdOp", "SumAccMaxBwdOp", "SumOp<true>"),
    "readout": ("readout_fwd",),
    "seg_gemm": ("fwd_reg_kernel", "fwd_fused_kernel", "fwd_h_kernel", "fwd_hd_kernel", "seg_gemm"),
    "wgrad": ("wgrad",),
    "batchnorm": ("bn_", "col_sums"),
    "fused_bwd": ("fused_bwd_kernel",),
}


def family_traffic(storage="fp32"):
    """HBM bytes per STEP of every event-timed kernel family from the same committed PMC passes as measured_traffic
    (FETCH_SIZE x 2 + WRITE_SIZE per launch x launches, over the profiled steps)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")))
    files = [f for f in files if "R1" not in os.path.basename(f) and _pmc_tag(os.path.basename(f)) == storage]
    for path in reversed(files):
        try:
            with open(path) as f:
                kernels = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        steps = max((rec.get("fetch_launches", 0) for name, rec in kernels.items() if "readout_fwd" in name), default=0)
        if not steps:
            continue
        out = {}
        for fam, pats in FAMILY_PATTERNS.items():
            tot = 0
            for name, rec in kernels.items():
                if any(p in name for p in pats):
                    tot += rec.get("hbm_bytes_per_launch", 0) * max(rec.get("fetch_launches", 0), rec.get("write_launches", 0))
            out[fam] = int(tot / steps)
        return out, os.path.basename(path)
    return None, None


measured_traffic.step_bytes = None


def make_workload(args, rank, device, batch, storage="fp32", widths=(64, 64), dense=128):
    import deepchem_amd as dc
    from deepchem_amd.data.collate import collate_to_device
    from deepchem_amd.metrics import to_one_hot
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    packed = synthetic_molecules(batch, seed=1000 + rank)
    y, w = synthetic_labels(batch, args.tasks, "classification", seed=1000 + rank)
    dbatch = collate_to_device(packed, None, device)
    labels = torch.as_tensor(to_one_hot(y.flatten(), 2).reshape(-1, args.tasks, 2).astype(np.float32),
                             device=device)
    weights = torch.as_tensor(w.astype(np.float32), device=device)
    model = dc.models.torch_models.GraphConvModel(args.tasks, number_input_features=[75] + list(widths[:-1]),
                                                  graph_conv_layers=list(widths), dense_layer_size=dense,
                                                  batch_size=batch, mode="classification",
                                                  grad_mode=args.grad_mode, device=device,
                                                  learning_rate=1e-3, log_frequency=10**9, activation_storage=storage)
    return model, dbatch, labels, weights


def run_steps(model, dbatch, labels, weights, n):
    def gen():
        for _ in range(n):
            yield (dbatch, [labels], [weights])
    return model.fit_generator(gen(), checkpoint_interval=0)


def usable_cores() -> int:
    """Host cores this process may actually use (affinity mask and cgroup CPU quota), not
    the machine's core count: oversubscribing torch's thread pool makes the CPU arm look
    absurdly slow."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 64))


MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak (MI355X_MICROARCH.md); fp32 matrix peak: 157.3
MFMA_F32_PEAK_TFLOPS = 157.3


def head_gemm_utilisation(device, args, iters: int = 50):
    """BASELINE metric (iii): the post-readout task head (B, 256) x (256, tasks * classes) + bias -- the one product
    of the path that is a plain dense GEMM -- timed alone with events on the stream the library
    launches on, for the Tox21 head at this run's batch and the PCBA-shaped head (128 two-class tasks, 8 192
    molecules per GPU).  ``tflops`` counts 2 B K n useful flops; ``mfma_frac`` prices them against what the matrix
    pipe can deliver IN THIS ARITHMETIC: fast mode spends six bf16 MFMAs per fp32-equivalent product (peak / 6),
    exact mode runs the fp32 MFMA.  The head reads B x 256 floats for 2 * 256 * n flops per row, so it is bound by
    HBM (or, at small B, by the launch): ``hbm_frac`` is the fraction that matters."""
    from deepchem_amd import ops
    peak = MFMA_BF16_PEAK_TFLOPS / 6.0 if args.gemm_mode == "fast" else MFMA_F32_PEAK_TFLOPS
    res = {"gemm_mode": args.gemm_mode, "mfma_peak_tflops_this_arithmetic": round(peak, 1)}
    gen = torch.Generator(device="cpu").manual_seed(11)
    scratch = torch.empty(ops.task_head_scratch_floats(), dtype=torch.float32, device=device)

    def timed(run):
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    for name, rows, n_out in (("tox21", args.batch, args.tasks * 2), ("pcba", 8192, 256)):
        a = torch.randn((rows, 256), generator=gen).to(device)
        w = (torch.randn((n_out, 256), generator=gen) * 0.05).to(device)  # nn.Linear layout, as the model holds it
        b = torch.zeros(n_out, device=device)
        o = torch.empty((rows, n_out), device=device)
        # the head as the whole-model step runs it (gcmi_task_head_forward): with more than 32 outputs the head matrix
        # is split into fragment images first (one small launch, counted) and the product runs from them
        us = timed(lambda: ops.task_head_forward(a, w, b, scratch, o))
        # ... and the plain segmented product on the same operands (no scratch: weights split per workgroup)
        us_plain = timed(lambda: ops.seg_gemm([0], [rows], a, w.reshape(-1), [0], None, None, None, b, [0], n_out, True, False,
                                              rows, k1=256, out=o))
        flops = 2.0 * rows * 256 * n_out
        nbytes = 4.0 * (rows * (256 + n_out) + 256 * n_out + n_out)
        res[name] = {"rows": rows, "k": 256, "n_out": n_out, "us": round(us, 2), "us_seg_gemm": round(us_plain, 2),
                     "tflops": round(flops / (us * 1e-6) / 1e12, 2),
                     "mfma_frac": round(flops / (us * 1e-6) / 1e12 / peak, 4),
                     "GBps": round(nbytes / (us * 1e-6) / 1e9, 1),
                     "hbm_frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "bound": "hbm" if flops / nbytes < peak * 1e12 / (HBM_PEAK_GBS * 1e9) else "mfma"}
    return res


def cpu_baseline(args, seconds, batch=100, faithful=True, grad_mode="reference"):
    """The oracle's training step (reference-faithful ops, reference gradient cut, Adam) at the
    reference's default batch size on pre-collated batches; threads = all host cores."""
    from deepchem_amd.feat.mol_graphs import collate_packed
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    from oracle import graphconv_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    B = batch
    n_batches = 8 if batch <= 1000 else 2
    packed = synthetic_molecules(B * n_batches, seed=77)
    y, w = synthetic_labels(B * n_batches, args.tasks, "classification", seed=77)
    cfg = O.ModelConfig(args.tasks, batch_size=B)
    tr = O.OracleTrainer(cfg, O.init_state(cfg, 0), grad_mode=grad_mode, faithful=faithful)
    batches = []
    for b in range(n_batches):
        sel = np.arange(b * B, (b + 1) * B)
        m = collate_packed(packed, sel)
        multi = dict(atom_features=m.get_atom_features(), deg_slice=m.deg_slice, membership=m.membership,
                     deg_adj_lists=m.get_deg_adjacency_lists())
        batches.append(O.batch_tensors(multi, B, y[sel], w[sel], cfg))
    tr.train_step(*batches[0])  # warm-up
    t0 = time.time()
    steps = 0
    while time.time() - t0 < seconds:
        tr.train_step(*batches[steps % n_batches])
        steps += 1
    wall = time.time() - t0
    return {
        "value": round(steps * B / wall, 1),
        "unit": "molecules/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d training steps (fwd+bwd+Adam, grad_mode=%s, %s segment max, batch %d, pre-collated "
                  "Tox21-like synthetic batches, %d tasks) in %.1f s; torch threads=%d" %
                  (steps, grad_mode, "faithful O(B N F)" if faithful else "O(N F)", B, args.tasks, wall,
                   torch.get_num_threads()),
    }


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start ``torch.distributed.run`` with N ranks of this script as a child
    process.  Nothing in this process has initialised the GPU yet (``import torch`` and ``device_count`` do not),
    and the parent only waits, so no GPU context is ever replaced."""
    import socket
    import subprocess
    backend = os.environ.get("GCMI_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        raise SystemExit("bench.py --gpus %d: this node has %d GPU(s); refusing to run fewer ranks than asked "
                         "(GCMI_BENCH_BACKEND=gloo rehearses the rank logic on shared devices)" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


TOX21_TASKS = ['NR-AR', 'NR-AR-LBD', 'NR-AhR', 'NR-Aromatase', 'NR-ER', 'NR-ER-LBD', 'NR-PPAR-gamma', 'SR-ARE',
               'SR-ATAD5', 'SR-HSE', 'SR-MMP', 'SR-p53']


def tox21_real(device, epochs=10):
    """BASELINE.json config 2 on the real file (tests/golden/tox21.csv.gz, a copy of the reference tree's
    datasets/tox21.csv.gz): MolNet recipe -- native featurizer, index split 80/10/10, BalancingTransformer -- and
    fit() / predict() timed end to end (shuffle, collation, H2D, every optimizer step) at MolNet's preset batch 64
    (molnet/preset_hyper_parameters.py:49-56) and the reference's default batch 100 (graphconvmodel.py:292)."""
    import deepchem_amd as dc
    from deepchem_amd.data.data_loader import convert_df_to_numpy, load_csv_files
    path = os.path.join(ROOT, "tests", "golden", "tox21.csv.gz")
    if not os.path.exists(path):
        return None
    t0 = time.perf_counter()
    df = next(iter(load_csv_files([path], shard_size=8192)))
    packed, keep = dc.feat.ConvMolFeaturizer().featurize_packed(df["smiles"].tolist())
    y, w = convert_df_to_numpy(df, TOX21_TASKS)
    y, w = y[keep], w[keep]
    featurize_s = time.perf_counter() - t0
    n = packed.n_mols
    a, b = int(0.8 * n), int(0.9 * n)
    train = dc.data.PackedDataset(packed.select(np.arange(a)), y[:a], w[:a])
    _, _, w_bal, _ = dc.trans.BalancingTransformer(dataset=train).transform_array(None, train.y, train.w, None)
    train = dc.data.PackedDataset(train.packed, train.y, w_bal)
    valid = dc.data.PackedDataset(packed.select(np.arange(a, b)), y[a:b], w[a:b])
    rec = {"molecules": n, "atoms": packed.n_atoms, "train_molecules": a, "load_and_featurize_s": round(featurize_s, 3),
           "timed_epochs": epochs}
    for B, lr in ((64, 5e-4), (100, 1e-3)):
        for gm in ("reference", "full"):
            for engine in (True, False):
                if not engine and gm == "full":
                    continue
                model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=B,
                                                              learning_rate=lr, grad_mode=gm, device=device,
                                                              log_frequency=10**9)
                model.small_batch_engine = engine
                np.random.seed(123)
                model.fit(train, nb_epoch=1, checkpoint_interval=0)  # warm-up: label upload, workspace, first launches
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.fit(train, nb_epoch=epochs, checkpoint_interval=0)
                torch.cuda.synchronize()
                wall = time.perf_counter() - t1
                key = "fit_molecules_per_s_batch_%d_%s%s" % (B, gm, "" if engine else "_per_batch_path")
                rec[key] = round(epochs * a / wall, 1)
                if gm == "reference":
                    model.predict(valid)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(5):
                        model.predict(valid)
                    rec["predict_molecules_per_s_batch_%d%s" % (B, "" if engine else "_per_batch_path")] = round(
                        5 * len(valid) / (time.perf_counter() - t1), 1)
    # the opt-in bf16 activation storage, beside -- never instead of -- the fp32 numbers above
    bf = {"dtype": "bf16-storage", "what": "GraphConv outputs, pooled rows and the dense output stored as bfloat16; fp32 "
                                           "operands, accumulation, parameters, gradients and optimizer state"}
    for B, lr in ((64, 5e-4), (100, 1e-3)):
        model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=B, learning_rate=lr,
                                                      device=device, log_frequency=10**9, activation_storage="bf16")
        np.random.seed(123)
        model.fit(train, nb_epoch=1, checkpoint_interval=0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.fit(train, nb_epoch=epochs, checkpoint_interval=0)
        torch.cuda.synchronize()
        bf["fit_molecules_per_s_batch_%d_reference" % B] = round(epochs * a / (time.perf_counter() - t1), 1)
    rec["bf16_storage"] = bf
    # the whole MolNet preset (40 epochs at batch 64), wall time next to examples/stable_results.csv:5 (165.2 s)
    model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=64, learning_rate=5e-4,
                                                  device=device, log_frequency=10**9)
    np.random.seed(123)
    t1 = time.perf_counter()
    model.fit(train, nb_epoch=40, checkpoint_interval=0)
    torch.cuda.synchronize()
    rec["molnet_preset_40_epochs_wall_s"] = round(time.perf_counter() - t1, 3)
    rec["reference_published_wall_s"] = 165.2
    return rec


def tox21_real_dp(device, rank, world, epochs=10):
    """The real-Tox21 fit() legs on N ranks: every rank fits its contiguous share of the train split
    (equal shares, so equal batch counts) under shard_model -- the small-batch engine with the gradient all-reduce
    between backward and Adam of every in-library step (gcmi_small_fit_dp).  Rate = molecules of ALL ranks / slowest rank."""
    import torch.distributed as dist
    import deepchem_amd as dc
    from deepchem_amd.data.data_loader import convert_df_to_numpy, load_csv_files
    from deepchem_amd.dist import shard_model
    path = os.path.join(ROOT, "tests", "golden", "tox21.csv.gz")
    if not os.path.exists(path):
        return None
    df = next(iter(load_csv_files([path], shard_size=8192)))
    packed, keep = dc.feat.ConvMolFeaturizer().featurize_packed(df["smiles"].tolist())
    y, w = convert_df_to_numpy(df, TOX21_TASKS)
    y, w = y[keep], w[keep]
    a = int(0.8 * packed.n_mols)
    per = a // world
    lo = rank * per
    mine = dc.data.PackedDataset(packed.select(np.arange(lo, lo + per)), y[lo:lo + per], w[lo:lo + per])
    rec = {"ranks": world, "train_molecules_per_rank": per, "timed_epochs": epochs}
    for B, lr in ((64, 5e-4), (100, 1e-3)):
        for gm in ("reference", "full"):
            model = dc.models.torch_models.GraphConvModel(12, number_input_features=[75, 64], batch_size=B, learning_rate=lr,
                                                          grad_mode=gm, device=device, log_frequency=10**9)
            shard_model(model)
            np.random.seed(123)
            model.fit(mine, nb_epoch=1, checkpoint_interval=0)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            model.fit(mine, nb_epoch=epochs, checkpoint_interval=0)
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rec["fit_molecules_per_s_batch_%d_%s" % (B, gm)] = round(epochs * per * world / float(t.item()), 1)
            rec["engine_batch_%d_%s" % (B, gm)] = model.__dict__.get("_small") is not None
    return rec


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; the two must agree" % (args.gpus, world))
    if args.scaling == "strong":
        if args.global_batch % world:
            raise SystemExit("bench.py: --global-batch %d is not divisible by %d ranks" % (args.global_batch, world))
        args.batch = args.global_batch // world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU implementation)")
    import torch.distributed as dist
    # one rank per GPU over RCCL.  GCMI_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer
    # GPUs than ranks (ranks then share devices round-robin; the numbers mean nothing, the code path
    # -- parameter broadcast, flat-bucket all-reduce, barriers, max over ranks -- is the real one)
    backend = os.environ.get("GCMI_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if world > 1 else 0
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import deepchem_amd
    from deepchem_amd import ops
    deepchem_amd.set_gemm_mode(args.gemm_mode)
    from deepchem_amd._lib import (K_BATCHNORM, K_FUSED_BWD, K_GATHER_MAX, K_GATHER_MAX_BWD, K_GATHER_SUM, K_READOUT,
                                   K_SEG_GEMM, K_WGRAD)

    model, dbatch, labels, weights = make_workload(args, rank, device, args.batch, args.storage)
    if world > 1:
        from deepchem_amd.dist import shard_model
        shard_model(model)
    run_steps(model, dbatch, labels, weights, max(args.warmup, 1))

    kernel_ids = {"gather_sum": K_GATHER_SUM, "gather_max": K_GATHER_MAX, "gather_max_bwd": K_GATHER_MAX_BWD,
                  "readout": K_READOUT, "seg_gemm": K_SEG_GEMM, "wgrad": K_WGRAD, "batchnorm": K_BATCHNORM,
                  "fused_bwd": K_FUSED_BWD}
    # Inside the timed region only the roofline kernel is event-timed: every timed launch is
    # bracketed by two event records on the launch stream, which costs ~10 us of queue time each.
    # The other kernel families are measured in a second, untimed pass.
    ops.timing_enable(K_GATHER_SUM, True)
    ops.timing_read(K_GATHER_SUM, reset=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run_steps(model, dbatch, labels, weights, args.steps)
    barrier()
    wall = time.perf_counter() - t0
    gather_time = ops.timing_read(K_GATHER_SUM, reset=True)
    breakdown_steps = min(args.steps, 5)
    for kid in kernel_ids.values():
        ops.timing_enable(kid, True)
        ops.timing_read(kid, reset=True)
    run_steps(model, dbatch, labels, weights, breakdown_steps)
    torch.cuda.synchronize()
    ktimes = {name: ops.timing_read(kid, reset=True) for name, kid in kernel_ids.items()}
    for kid in kernel_ids.values():
        ops.timing_enable(kid, False)
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    g = dbatch.graph
    n0 = g.deg_counts[0]
    # gather-sum launches of one step: forward layer 0 (F=75) and forward layer 1 (F=64).  (The backward's gather of dS
    # is part of the two-stage window pass with the GraphPool backward below it, timed with that family; when that
    # pass is switched off it is a third, accumulating launch here.)
    per_step = gather_sum_bytes(g.n_atoms, g.n_edges, n0, 75, False) + \
        gather_sum_bytes(g.n_atoms, g.n_edges, n0, 64, False)
    launches_per_step = 2
    if args.grad_mode == "full" and gather_time[0] >= 3 * args.steps:
        per_step += gather_sum_bytes(g.n_atoms, g.n_edges, n0, 64, True)
        launches_per_step = 3
    n_launch, ms = gather_time
    achieved = (per_step * args.steps) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, traffic_file = measured_traffic(args.storage)
    h = args.storage != "fp32"
    if h:  # bf16 rows: 2 bytes per gathered / written element ("replace 4 F by 2 F for feature reads")
        e_b = lambda f: 2 * f
        per_step = (g.n_edges * (4 * 76 + 4) + g.n_atoms * (2 * e_b(80))) + \
                   (g.n_edges * (e_b(64) + 4) + g.n_atoms * e_b(64))
        achieved = (per_step * args.steps) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

    out = {
        "metric": "molecules/sec fwd+bwd GraphConvModel",
        "value": round(args.batch * world * args.steps / wall, 1),
        "unit": "molecules/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32" if not h else "f32 (bf16 activation storage)",
        "data": "synthetic",
        "config": {
            "workload": "Tox21-like synthetic molecules (18.2 atoms/mol, E/N 2.08), 12 binary tasks, "
                        "GraphConvModel [64,64]/dense 128, BatchNorm on, fwd+bwd+Adam per step",
            "molecules_per_gpu_per_step": args.batch,
            "atoms_per_gpu_per_step": g.n_atoms,
            "directed_edges_per_gpu_per_step": g.n_edges,
            "grad_mode": args.grad_mode,
            "gemm_mode": args.gemm_mode + (" (split-bf16 x3 products on v_mfma_f32_32x32x16_bf16, fp32-accurate)"
                                           if args.gemm_mode == "fast" else " (v_mfma_f32_32x32x2_f32 chain)"),
            "parallelism": "dp%d (molecules sharded by rank, one flat all-reduce per step)" % world,
        },
        "roofline": {
            "kernel": ("win_kernel<512, {16|19}, false, SumOp<false>> (gather_lds.hip: GraphConv.sum_neigh over LDS "
                       "molecule windows, forward of both layers)") if not h else
                      "win_kernel<512, 19, false, SumOpFH> + win_kernel<512, 8, false, SumOpH> (bf16 rows)",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_file,
            # the same launches priced by the bytes they actually move (every atom row is read from HBM once and the
            # ~3-fold re-reads of SURVEY 8d's count are served from LDS): PMC bytes per launch / measured launch time
            "achieved_real": round(traffic / (ms * 1e-3 / max(n_launch, 1)) / 1e9, 1) if traffic and ms > 0 else None,
            "frac_real": round(traffic / (ms * 1e-3 / max(n_launch, 1)) / 1e9 / HBM_PEAK_GBS, 4) if traffic and ms > 0 else None,
            "algorithmic_bytes_per_step": int(per_step),
            "launches_per_step": launches_per_step,
            "avg_launch_us": round(ms * 1e3 / max(n_launch, 1), 2),
        },
        "kernel_ms_per_step": {k: round(v[1] / breakdown_steps, 4) for k, v in ktimes.items()},
    }
    if measured_traffic.step_bytes:
        # all kernels of the step together: measured bytes (same counter passes) over the measured step time
        sb = measured_traffic.step_bytes
        out["step_traffic"] = {"hbm_bytes_per_step": sb, "source": traffic_file,
                               "achieved": round(sb / (wall / args.steps) / 1e9, 1), "unit": "GB/s",
                               "frac_of_hbm_peak": round(sb / (wall / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
    # Every event-timed kernel family against the same HBM peak (second, untimed pass): algorithmic bytes per
    # step = operands read once + results written once (DESIGN.md section 7 lists the terms).
    N, E, B = g.n_atoms, g.n_edges, args.batch
    fam_bytes = {
        "gather_sum": per_step,
        "gather_max": 2 * (E * (4 * 64 + 4) + 2 * N * 4 * 64 + N * 64),
        # GraphPool backward of the last block; in "full" mode also the two-stage pass (gather of dS onto the self
        # part + the GraphPool backward of block 0; its dX never leaves LDS: minus one write and one read of N x 64)
        "gather_max_bwd": (E * (5 * 64 + 5) + 2 * N * 4 * 64 + N * 64) +
        ((E * (5 * 64 + 5) + 2 * N * 4 * 64 + N * 64) + gather_sum_bytes(N, E, n0, 64, True) - 2 * N * 4 * 64
         if args.grad_mode == "full" else 0),
        "readout": N * (4 * 128 + 4) + B * 8 * 128,
        # forward: GraphConv 0/1, dense, head; backward: head (the blocks' backward products: fused_bwd)
        "seg_gemm": 4 * (N * (75 + 75 + 64) + N * (64 + 64 + 64) + N * (64 + 128) + B * (256 + 24) + B * (24 + 256)),
        # dW = A^T dY of the head (the blocks' weight gradients: fused_bwd)
        "wgrad": 4 * B * (256 + 24),
        # forward statistics ride in the epilogue of the producing product, the backward sums come from per-molecule
        # data (dense layer) or from the block above (GraphConv layers): what is left reads B x 640 floats
        "batchnorm": 4 * B * (256 + 128 + 256),
        # one pass per block: dense (reads dense 128 + pool 64, writes dpool 64; + ~77 floats per atom of per-molecule
        # gradient rows fetched once per degree run), GraphConv 1 (dy, gc, S, x -> dS, dXs), GraphConv 0 (dy, gc, S, x)
        "fused_bwd": 4 * (N * (128 + 64 + 64) + (N * (6 * 64) + N * (64 + 64 + 76 + 76) if args.grad_mode == "full" else 0)),
    }
    fam_real, fam_file = family_traffic(args.storage)
    out["roofline_by_kernel"] = {}
    for k in ktimes:
        ms_k = ktimes[k][1] / breakdown_steps
        rec = {"ms": round(ms_k, 4)}
        if not h:  # SURVEY 8d's count (fp32 rows; the LDS windows beat it, hence fractions above 1 for the gathers)
            rec["algorithmic_bytes"] = int(fam_bytes[k])
            rec["frac_algorithmic"] = round(fam_bytes[k] / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) if ms_k > 0 else 0.0
        # what the family really moved: PMC bytes of the committed passes over the time measured in THIS run
        if fam_real is not None and ms_k > 0:
            rec["hbm_bytes"] = fam_real[k]
            rec["GBps_real"] = round(fam_real[k] / (ms_k * 1e-3) / 1e9, 1)
            rec["frac_real"] = round(fam_real[k] / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
        out["roofline_by_kernel"][k] = rec
    out["roofline_by_kernel_source"] = fam_file

    if rank == 0 and world == 1 and not args.profile_only:
        out["head_gemm"] = head_gemm_utilisation(device, args)

    if rank == 0 and world == 1 and not args.profile_only and args.storage == "fp32" and args.gemm_mode == "fast":
        # BASELINE config 2 "bf16/fp32": the SAME step with every activation the step writes and reads back stored as
        # bf16 (gcmi_model_* storage = 1; fp32 arithmetic, parameters, gradients, Adam) -- beside the f32 headline,
        # never instead of it
        for st_name, key in (("bf16", "bf16_storage"), ("bf16+grads", "bf16_storage_and_gradient_streams")):
            mb, db, lb, wb = make_workload(args, rank, device, args.batch, st_name)
            run_steps(mb, db, lb, wb, max(args.warmup, 1))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(mb, db, lb, wb, args.steps)
            torch.cuda.synchronize()
            wall_b = time.perf_counter() - t1
            for kid in kernel_ids.values():
                ops.timing_enable(kid, True)
                ops.timing_read(kid, reset=True)
            run_steps(mb, db, lb, wb, breakdown_steps)
            torch.cuda.synchronize()
            kt_b = {name: ops.timing_read(kid, reset=True) for name, kid in kernel_ids.items()}
            for kid in kernel_ids.values():
                ops.timing_enable(kid, False)
            tb, tb_file = measured_traffic(st_name)
            item = {"dtype": "bf16-storage" if st_name == "bf16" else "bf16-storage+gradient-streams", "value": round(args.batch * args.steps / wall_b, 1), "unit": "molecules/s",
                    "ms_per_step": round(wall_b / args.steps * 1e3, 4), "steps": args.steps,
                    "what": "atom-feature copy, neighbour sums, GraphConv outputs, pooled rows and the dense output stored as "
                            "bfloat16 (one rounding per stored element); fp32 accumulation, fp64 BatchNorm sums of the "
                            "rounded values, fp32 parameters / parameter gradients / Adam; gradient streams between kernels "
                            + ("fp32" if st_name == "bf16" else "bfloat16 too (dpool, dy, dS, dXs)"),
                    "kernel_ms_per_step": {k: round(v[1] / breakdown_steps, 4) for k, v in kt_b.items()}}
            if measured_traffic.step_bytes:
                sb = measured_traffic.step_bytes
                item["step_traffic"] = {"hbm_bytes_per_step": sb, "source": tb_file,
                                        "achieved": round(sb / (wall_b / args.steps) / 1e9, 1), "unit": "GB/s",
                                        "frac_of_hbm_peak": round(sb / (wall_b / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
            else:
                item["step_traffic"] = None
            fb, fb_file = family_traffic(st_name)
            if fb is not None:
                item["roofline_by_kernel"] = {
                    k: {"ms": round(v[1] / breakdown_steps, 4), "hbm_bytes": fb[k],
                        "frac_real": round(fb[k] / (v[1] / breakdown_steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) if v[1] > 0 else 0.0}
                    for k, v in kt_b.items()}
                item["roofline_by_kernel_source"] = fb_file
            out[key] = item
            del mb, db, lb, wb
            measured_traffic(args.storage)

    if rank == 0 and args.small_batch and world == 1:
        m2, b2, l2, w2 = make_workload(args, 0, device, args.small_batch)
        run_steps(m2, b2, l2, w2, 10)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_small = 100
        run_steps(m2, b2, l2, w2, n_small)
        torch.cuda.synchronize()
        out["config"]["molecules_per_s_at_batch_%d" % args.small_batch] = round(
            args.small_batch * n_small / (time.perf_counter() - t1), 1)

    if rank == 0 and world == 1 and args.fit_pipeline:
        # end-to-end fit(): native collation + H2D (prefetched on a worker thread) + training step
        import deepchem_amd as dc
        from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
        n_fit = 32768
        packed = synthetic_molecules(n_fit, seed=5)
        y, w = synthetic_labels(n_fit, args.tasks, "classification", seed=5)
        ds = dc.data.PackedDataset(packed, y, w)
        for bsz in (100, 4096):
            m3 = dc.models.torch_models.GraphConvModel(args.tasks, number_input_features=[75, 64],
                                                       batch_size=bsz, grad_mode=args.grad_mode,
                                                       device=device, log_frequency=10**9)
            n_part = 4096 if bsz == 100 else n_fit
            part = dc.data.PackedDataset(packed.select(np.arange(n_part)), y[:n_part], w[:n_part])
            m3.fit(part, nb_epoch=1, checkpoint_interval=0)  # warm-up
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_ep = 2 if bsz == 100 else 8
            m3.fit(part, nb_epoch=n_ep, checkpoint_interval=0)
            torch.cuda.synchronize()
            out["config"]["fit_molecules_per_s_batch_%d" % bsz] = round(n_ep * n_part / (time.perf_counter() - t1), 1)
        # the same loop on a set whose rows are featurizer-shaped (one-hot blocks): kept, collated and copied as
        # 8-byte atom codes and expanded on the GPU (deepchem_amd/feat/atom_codes.py)
        from deepchem_amd.utils.synthetic import PackedMols
        rng = np.random.RandomState(5)
        big = packed.select(np.arange(4 * args.batch) % n_fit)
        codes = np.stack([rng.randint(0, hi, big.n_atoms) for hi in (44, 11, 7, 1, 1, 5, 2, 5)], axis=1).astype(np.uint8)
        coded = PackedMols(None, big.atom_ptr, big.adj_ptr, big.adj_idx, codes)
        yb, wb = synthetic_labels(big.n_mols, args.tasks, "classification", seed=6)
        m4 = dc.models.torch_models.GraphConvModel(args.tasks, number_input_features=[75, 64], batch_size=args.batch,
                                                   grad_mode=args.grad_mode, device=device, log_frequency=10**9)
        ds4 = dc.data.PackedDataset(coded, yb, wb)
        m4.fit(ds4, nb_epoch=1, checkpoint_interval=0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fit_epochs = 8  # 32 batches: the fill of the batch pipeline (threads, pinned buffers, first batch) is amortised
        m4.fit(ds4, nb_epoch=fit_epochs, checkpoint_interval=0)
        torch.cuda.synchronize()
        out["config"]["fit_molecules_per_s_batch_%d_atom_codes" % args.batch] = round(
            fit_epochs * big.n_mols / (time.perf_counter() - t1), 1)

    if not args.profile_only and args.scaling == "weak" and not (args.batch == 8192 and args.tasks == 128):
        # BASELINE config 3's per-GPU shape beside the Tox21 headline: 8 192 molecules x 128 two-class tasks per rank,
        # same step (fwd + bwd + Adam, one flat all-reduce per step for N > 1), same timing rules (collective leg)
        import copy
        pa = copy.copy(args)
        pa.tasks = 128
        pm, pb, pl, pw = make_workload(pa, rank, device, 8192, args.storage)
        if world > 1:
            from deepchem_amd.dist import shard_model
            shard_model(pm)
        run_steps(pm, pb, pl, pw, 3)
        barrier()
        t1 = time.perf_counter()
        p_steps = 20
        run_steps(pm, pb, pl, pw, p_steps)
        barrier()
        p_wall = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([p_wall], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            p_wall = float(t.item())
        if rank == 0:
            out["config"]["pcba_shape"] = {
                "workload": "PCBA-like: 8 192 synthetic molecules x 128 binary tasks per GPU, GraphConvModel [64,64]/dense 128",
                "molecules_per_s": round(8192 * world * p_steps / p_wall, 1), "ms_per_step": round(p_wall / p_steps * 1e3, 4),
                "steps": p_steps, "n_gpus": world, "grad_mode": args.grad_mode}
        del pm, pb, pl, pw
    if rank == 0 and world == 1 and not args.profile_only and args.storage == "fp32":
        # MolNet's preset widths ([128, 128] GraphConv layers, dense 256: molnet/preset_hyper_parameters.py:128-135) at the
        # headline batch: these shapes run on the general product / gradient kernels, not the one-pass block kernels
        wm, wb, wl, ww = make_workload(args, rank, device, args.batch, "fp32", widths=(128, 128), dense=256)
        run_steps(wm, wb, wl, ww, 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(wm, wb, wl, ww, 5)
        torch.cuda.synchronize()
        w_wall = time.perf_counter() - t1
        out["config"]["widths_128_128_dense_256"] = {"molecules_per_s": round(args.batch * 5 / w_wall, 1),
                                                     "ms_per_step": round(w_wall / 5 * 1e3, 4), "steps": 5,
                                                     "molecules_per_step": args.batch}
        del wm, wb, wl, ww
    if rank == 0 and world == 1 and not args.profile_only and args.gemm_mode == "fast":
        # the same step on the exact-fp32 matrix-core chain (the arithmetic whose trajectories track the reference)
        deepchem_amd.set_gemm_mode("exact")
        try:
            run_steps(model, dbatch, labels, weights, 2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(model, dbatch, labels, weights, 5)
            torch.cuda.synchronize()
            out["config"]["value_exact_gemm_mode"] = round(args.batch * 5 / (time.perf_counter() - t1), 1)
        finally:
            deepchem_amd.set_gemm_mode("fast")
    if rank == 0 and world == 1 and args.fit_pipeline:
        real = tox21_real(device)
        if real is not None:
            out["config"]["tox21_real"] = real
    if world > 1 and args.fit_pipeline:  # (collective: every rank takes part)
        real = tox21_real_dp(device, rank, world)
        if rank == 0 and real is not None:
            out["config"]["tox21_real"] = real
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        # beside it (not the headline baseline): the oracle at a large batch, where its O(B N F) faithful segment max
        # cannot finish -- the same results by the O(N F) form (faithful=False), same gradient mode as the GPU line
        out["cpu_baseline_large_batch"] = cpu_baseline(args, min(args.cpu_seconds, 10.0), batch=4096, faithful=False,
                                                       grad_mode=args.grad_mode)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The data-parallel path of the REAL model on two ranks: each rank runs the fused native
step of ``GraphConvModel`` on its half of a global batch, the trained range of the flat gradient arena
goes through ``FlatGradAllReduce.reduce_flat`` (one collective per step), Adam steps on the result.

Two ranks share the one GPU of the test box, so the process group is ``gloo`` (RCCL needs one device per
rank); on the 8-GPU node ``bench.py`` runs the same code with backend nccl.  Checked: the averaged
gradient equals the single-process gradient of the concatenated batch (BatchNorm off: its statistics are
per rank by design), both ranks hold identical parameters before and after the step, and ``bench.py
--gpus 2`` really runs two ranks (or refuses), never silently one."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, world, out_dir, grad_mode = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1], sys.argv[2]
dist.init_process_group("gloo")
torch.cuda.set_device(0)
import deepchem_amd as dc
from deepchem_amd.dist import shard_indices, shard_model
from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
dc.set_gemm_mode("exact")
n, T = 16, 3
packed = synthetic_molecules(n, seed=12, max_atoms=30)
y, w = synthetic_labels(n, T, "classification", 12, pos_rate=0.4)
idx = shard_indices(np.arange(n))
torch.manual_seed(50 + rank)  # different initial weights per rank: the broadcast must fix that
model = dc.models.torch_models.GraphConvModel(T, number_input_features=[75, 64], batch_size=len(idx),
                                              batch_normalize=False, grad_mode=grad_mode,
                                              device=torch.device("cuda:0"), learning_rate=1e-3)
shard_model(model)
model.small_batch_engine = False  # the per-batch path: it leaves the step's (reduced) gradients in the arena
before = {k: v.detach().cpu().clone() for k, v in model.model.state_dict().items()}
ds = dc.data.PackedDataset(packed.select(idx), y[idx], w[idx])
loss = model.fit(ds, nb_epoch=1, deterministic=True, checkpoint_interval=0)
nat = model.model.__dict__.get("_native")
assert nat is not None, "the native step did not run"
lo, hi = nat.grad_range
torch.save({"before": before, "after": {k: v.detach().cpu() for k, v in model.model.state_dict().items()},
            "grad": nat.grad_flat[lo:hi].detach().cpu(), "range": (lo, hi), "loss": loss, "idx": idx},
           os.path.join(out_dir, "rank%%d.pt" %% rank))
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize("grad_mode", ["reference", "full"])
def test_two_ranks_of_the_real_model_average_to_the_global_batch_gradient(tmp_path, grad_mode):
    import deepchem_amd as dc
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script), str(tmp_path), grad_mode]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    r0 = torch.load(str(tmp_path / "rank0.pt"), weights_only=False)
    r1 = torch.load(str(tmp_path / "rank1.pt"), weights_only=False)
    assert list(r0["idx"]) == list(range(0, 8)) and list(r1["idx"]) == list(range(8, 16))
    for k in r0["before"]:  # rank 0's parameters everywhere, before and after the step
        assert torch.equal(r0["before"][k], r1["before"][k]), k
        assert torch.equal(r0["after"][k], r1["after"][k]), k
    assert torch.equal(r0["grad"], r1["grad"]) and r0["range"] == r1["range"]
    # single process, the concatenated batch, same initial parameters
    n, T = 16, 3
    packed = synthetic_molecules(n, seed=12, max_atoms=30)
    y, w = synthetic_labels(n, T, "classification", 12, pos_rate=0.4)
    dc.set_gemm_mode("exact")
    try:
        model = dc.models.torch_models.GraphConvModel(T, number_input_features=[75, 64], batch_size=n,
                                                      batch_normalize=False, grad_mode=grad_mode,
                                                      device=torch.device("cuda:0"), learning_rate=1e-3)
        model.model.load_state_dict({k: v.clone() for k, v in r0["before"].items()})
        model.small_batch_engine = False  # the per-batch path leaves the step's gradients in the arena
        model.fit(dc.data.PackedDataset(packed, y, w), nb_epoch=1, deterministic=True, checkpoint_interval=0)
    finally:
        dc.set_gemm_mode("fast")
    nat = model.model.__dict__["_native"]
    assert nat.grad_range == r0["range"]
    lo, hi = nat.grad_range
    whole = nat.grad_flat[lo:hi].detach().cpu()
    scale = float(whole.abs().max())
    assert float((whole - r0["grad"]).abs().max()) <= 1e-4 * scale, (float((whole - r0["grad"]).abs().max()), scale)
    if grad_mode == "reference":  # "dense-head gradients only": the bucket is the dense layer + head
        assert hi - lo < 64 * 128 + 128 + 256 * 2 * T + 2 * T + 64


ENGINE_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, world, out_dir, grad_mode = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1], sys.argv[2]
dist.init_process_group("gloo")
torch.cuda.set_device(0)
import deepchem_amd as dc
from deepchem_amd.dist import shard_model
from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
n, T, B = 48, 3, 8
packed = synthetic_molecules(n, seed=31, max_atoms=30)
y, w = synthetic_labels(n, T, "classification", 31, pos_rate=0.4)
# global batch k = molecules [16 k, 16 k + 16): rank r takes its half of every global batch (dist.shard_indices)
idx = np.concatenate([np.arange(16 * k + 8 * rank, 16 * k + 8 * rank + 8) for k in range(n // 16)])
torch.manual_seed(70 + rank)  # different initial weights per rank: the broadcast must fix that
model = dc.models.torch_models.GraphConvModel(T, number_input_features=[75, 64], batch_size=B,
                                              batch_normalize=False, grad_mode=grad_mode,
                                              device=torch.device("cuda:0"), learning_rate=1e-3, log_frequency=1)
shard_model(model)
before = {k: v.detach().cpu().clone() for k, v in model.model.state_dict().items()}
ds = dc.data.PackedDataset(packed.select(idx), y[idx], w[idx])
losses = []
model.fit(ds, nb_epoch=2, deterministic=True, checkpoint_interval=0, all_losses=losses)
eng = model.__dict__.get("_small")
assert eng is not None, "the small-batch engine did not run"
nat = model.model.__dict__.get("_native")
lo, hi = nat.grad_range
assert float(nat.grad_flat[lo:hi].abs().max()) == 0.0  # the engine leaves its gradient scratch clean: it was the engine
torch.save({"before": before, "after": {k: v.detach().cpu() for k, v in model.model.state_dict().items()},
            "losses": losses, "steps": model.get_global_step()},
           os.path.join(out_dir, "rank%%d.pt" %% rank))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(600)
@pytest.mark.parametrize("grad_mode", ["reference", "full"])
def test_two_ranks_on_the_small_batch_engine_reproduce_the_single_process_run(tmp_path, grad_mode):
    """Data parallel INSIDE the small-batch engine (gcmi_small_fit_dp: the all-reduce sits between the backward
    launches and the Adam launch of every step of the in-library loop).  Two ranks, each with its half of every
    global batch of 16, six optimizer steps over two epochs, BatchNorm off (its statistics are per rank by design):
    both ranks end with identical parameters, and those equal a single process training on the whole batches."""
    import deepchem_amd as dc
    from deepchem_amd.utils.synthetic import synthetic_labels, synthetic_molecules
    script = tmp_path / "engine_worker.py"
    script.write_text(ENGINE_WORKER % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script), str(tmp_path), grad_mode]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    r0 = torch.load(str(tmp_path / "rank0.pt"), weights_only=False)
    r1 = torch.load(str(tmp_path / "rank1.pt"), weights_only=False)
    assert r0["steps"] == r1["steps"] == 6
    for k in r0["before"]:
        assert torch.equal(r0["before"][k], r1["before"][k]), k
        assert torch.equal(r0["after"][k], r1["after"][k]), k
    n, T = 48, 3
    packed = synthetic_molecules(n, seed=31, max_atoms=30)
    y, w = synthetic_labels(n, T, "classification", 31, pos_rate=0.4)
    model = dc.models.torch_models.GraphConvModel(T, number_input_features=[75, 64], batch_size=16,
                                                  batch_normalize=False, grad_mode=grad_mode,
                                                  device=torch.device("cuda:0"), learning_rate=1e-3, log_frequency=1)
    model.model.load_state_dict({k: v.clone() for k, v in r0["before"].items()})
    losses = []
    model.fit(dc.data.PackedDataset(packed, y, w), nb_epoch=2, deterministic=True, checkpoint_interval=0,
              all_losses=losses)
    assert model.get_global_step() == 6
    # the mean of the two ranks' losses is the whole batch's loss, step by step
    both = 0.5 * (np.array(r0["losses"]) + np.array(r1["losses"]))
    assert np.allclose(both, np.array(losses), rtol=2e-4, atol=1e-6), (both, losses)
    after = {k: v.detach().cpu() for k, v in model.model.state_dict().items()}
    changed = 0
    for k, v in after.items():
        if not v.is_floating_point():
            continue
        a, b = r0["after"][k].double(), v.double()
        scale = max(float(b.abs().max()), 1e-3)
        d = (a - b).abs()
        # Adam moves an entry whose gradient is at rounding level by lr * sign(noise): single entries may differ by a
        # few learning-rate steps in ANY two runs that sum in a different order; the bulk must agree
        assert float(d.median()) <= 2e-5 * scale and float(d.max()) <= 6 * 1e-3 * 2, (k, float(d.median()), float(d.max()))
        changed += int(not torch.equal(r0["before"][k], r0["after"][k]))
    assert changed >= (4 if grad_mode == "reference" else 30)  # (degrees the 48 molecules do not have keep their weights)


FAMILY_COMMON = r'''
def make(family, B, seed):
    import numpy as np, torch
    import deepchem_amd as dc
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    if family == "mpnn":
        from deepchem_amd.models.torch_models.mpnn import MPNNModel
        from tests.test_gpu_mpnn_model import qm9_like
        model = MPNNModel(2, n_atom_feat=20, n_pair_feat=6, n_hidden=32, T=2, M=3, mode="regression", batch_size=B,
                          device=dev, learning_rate=1e-3)
        X = qm9_like(8, 20, 6, seed=9)
        rng = np.random.RandomState(10)
        y, w = rng.randn(8, 2), (rng.rand(8, 2) < 0.8).astype(float)
    else:
        from deepchem_amd.models.torch_models import WeaveModel, WeaveMol
        from tests.util import load_golden
        M = load_golden("weave_model.npz")
        X = np.empty(8, dtype=object)
        for i in range(8):
            X[i] = WeaveMol(M["mol%d_nodes" % i], M["mol%d_pairs" % i], M["mol%d_edges" % i])
        y, w = M["classification_y"][:8], M["classification_w"][:8]
        model = WeaveModel(2, fully_connected_layer_sizes=[40, 20], batch_size=B, mode="classification",
                           learning_rate=1e-3, device=dev)
    return model, X, y, w
'''

FAMILY_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, out_dir, family = int(os.environ["RANK"]), sys.argv[1], sys.argv[2]
dist.init_process_group("gloo")
torch.cuda.set_device(0)
import deepchem_amd as dc
from deepchem_amd.dist import shard_model
dc.set_gemm_mode("exact")
exec(open(os.path.join(out_dir, "family_common.py")).read())
model, X, y, w = make(family, 4, 80 + rank)   # different initial weights per rank: the broadcast must fix that
model._ensure_built()
shard_model(model)
assert model._grad_arena is not None, "no flat gradient arena"
calls = {"flat": 0, "per_tensor": 0}
rf, pc = model._grad_sync.reduce_flat, type(model._grad_sync).__call__
def counting_flat(bucket):
    calls["flat"] += 1
    return rf(bucket)
model._grad_sync.reduce_flat = counting_flat
before = {k: v.detach().cpu().clone() for k, v in model.model.state_dict().items()}
sel = np.arange(4 * rank, 4 * rank + 4)
loss = model.fit(dc.data.NumpyDataset(X[sel], y[sel], w[sel]), nb_epoch=1, deterministic=True, checkpoint_interval=0)
assert calls["flat"] == 1, calls   # ONE zero-copy collective on the arena, not the per-tensor copy-in / copy-out
arena = model._grad_arena
assert arena.intact()
grads = {n: v.detach().cpu().clone() for (n, p), (q, v) in zip([(n, p) for n, p in model.model.named_parameters()
                                                                 if p.requires_grad], arena.views)}
torch.save({"before": before, "after": {k: v.detach().cpu() for k, v in model.model.state_dict().items()},
            "grads": grads, "loss": float(loss)}, os.path.join(out_dir, "rank%%d.pt" %% rank))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(600)
@pytest.mark.parametrize("family", ["mpnn", "weave"])
def test_two_ranks_of_mpnn_and_weave_exchange_one_flat_bucket(tmp_path, family):
    """BASELINE configs 4 (MPNNModel, 1 -> 8 GPUs) and 5 (WeaveModel, 2 GPUs) through ``shard_model``: their backward
    runs through autograd, so ``shard_model`` puts ONE flat arena behind every ``p.grad`` (dist.FlatGradArena) and the
    step's exchange is ``reduce_flat`` on it -- one collective -- instead of the per-tensor copy-in / copy-out.  Two
    ranks with four molecules each: identical parameters before and after, and the averaged gradient equals the
    single-process gradient of the eight-molecule batch (1e-4 of each tensor's scale)."""
    import deepchem_amd as dc
    (tmp_path / "family_common.py").write_text(FAMILY_COMMON)
    script = tmp_path / "family_worker.py"
    script.write_text(FAMILY_WORKER % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script), str(tmp_path), family]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    r0 = torch.load(str(tmp_path / "rank0.pt"), weights_only=False)
    r1 = torch.load(str(tmp_path / "rank1.pt"), weights_only=False)
    for k in r0["before"]:
        assert torch.equal(r0["before"][k], r1["before"][k]), k
        assert torch.equal(r0["after"][k], r1["after"][k]), k
    for k in r0["grads"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k
    # single process: the eight molecules as one batch, same initial parameters
    ns = {}
    exec(FAMILY_COMMON, ns)
    dc.set_gemm_mode("exact")
    try:
        model, X, y, w = ns["make"](family, 8, 5)
        model._ensure_built()
        model.model.load_state_dict({k: v.clone() for k, v in r0["before"].items()})
        model.model.train()
        (inputs, labels, weights), = list(model.default_generator(dc.data.NumpyDataset(X, y, w), pad_batches=True))
        inputs, labels, weights = model._prepare_batch((inputs, labels, weights))
        outputs = model._forward_lists(model._unwrap_single(inputs))
        if model._roles.declared:
            outputs = [outputs[i] for i in model._roles.loss]
        loss = model._loss_fn(outputs, labels, weights)
        loss.backward()
    finally:
        dc.set_gemm_mode("fast")
    assert abs(float(loss) - 0.5 * (r0["loss"] + r1["loss"])) <= 1e-4 * max(1.0, abs(float(loss)))
    checked = 0
    for name, p in model.model.named_parameters():
        if not p.requires_grad:
            continue
        a = r0["grads"][name].double()
        b = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().double()
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 1e-4 * scale + 1e-9, (name, float((a - b).abs().max()), scale)
        checked += int(float(b.abs().max()) > 0)
    assert checked >= 4, checked


@pytest.mark.timeout(900)
def test_bench_gpus_2_runs_two_ranks_or_refuses(tmp_path):
    """The driver's command line.  On a 1-GPU box with the default backend bench.py must refuse loudly; with the
    gloo rehearsal switch it must run TWO ranks and say n_gpus = 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch",
            "512", "--no-cpu-baseline", "--fit-pipeline", "0", "--small-batch", "0"]
    if torch.cuda.device_count() < 2:
        refused = subprocess.run(base, env=env, capture_output=True, text=True, timeout=300)
        assert refused.returncode != 0 and "refusing" in (refused.stderr + refused.stdout)
    env["GCMI_BENCH_BACKEND"] = "gloo"
    done = subprocess.run(base, env=env, capture_output=True, text=True, timeout=800)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    line = [l for l in done.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["molecules_per_gpu_per_step"] == 512
    # WORLD_SIZE that disagrees with --gpus is an error
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run(base, env=env2, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "must agree" in (bad.stderr + bad.stdout)
    # strong scaling: a fixed global batch split over the ranks
    # ... and the real-Tox21 fit() legs on the ranks (small-batch engine under data parallel: VERDICT r2 item 4)
    args = [a for a in base]
    args[args.index("--fit-pipeline") + 1] = "1"
    strong = subprocess.run(args + ["--scaling", "strong", "--global-batch", "1024"], env=env, capture_output=True,
                            text=True, timeout=800)
    assert strong.returncode == 0, strong.stdout[-2000:] + strong.stderr[-4000:]
    rec = json.loads([l for l in strong.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["config"]["molecules_per_gpu_per_step"] == 512
    real = rec["config"]["tox21_real"]
    assert real["ranks"] == 2 and real["engine_batch_64_reference"] and real["engine_batch_100_full"]
    assert real["fit_molecules_per_s_batch_64_reference"] > 0

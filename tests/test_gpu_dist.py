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
    strong = subprocess.run(base + ["--scaling", "strong", "--global-batch", "1024"], env=env, capture_output=True,
                            text=True, timeout=800)
    assert strong.returncode == 0, strong.stdout[-2000:] + strong.stderr[-4000:]
    rec = json.loads([l for l in strong.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["config"]["molecules_per_gpu_per_step"] == 512

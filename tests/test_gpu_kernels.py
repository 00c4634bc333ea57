"""GPU parity, kernel by kernel: every entry point of the C ABI (through
deepchem_amd.ops -> ctypes -> libgcmi.so) against the oracle on the same seeded
inputs.  fp32 tolerance: 1e-4 relative (BASELINE.json north_star); indices,
arg-max winners and tie-breaking must match exactly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from deepchem_amd.feat.mol_graphs import collate_packed
from deepchem_amd.utils.synthetic import (concat_packed, single_atom_and_edge_cases,
                                          synthetic_molecules)
from oracle import graphconv_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def make_batch(n_mols=60, seed=0, n_feat=75, high_degree=True, int_features=False):
    """(cpu reference-layout inputs, BatchGraph, x on GPU)."""
    from deepchem_amd.graph import BatchGraph
    sets = [synthetic_molecules(n_mols, seed=seed, n_feat=n_feat)]
    if high_degree:
        sets.append(single_atom_and_edge_cases(n_feat, seed))
        sets.append(synthetic_molecules(6, seed=seed + 1, n_feat=n_feat, mean_atoms=14, max_atoms=40,
                                        parent_weights=(1,) * 10, ring_deg=10, ring_p_deg3=1.0,
                                        rings_per_atom=0.8))
    packed = concat_packed(sets)
    multi = collate_packed(packed)
    rng = np.random.RandomState(seed)
    x = rng.standard_normal((multi.get_num_atoms(), n_feat)).astype(np.float32)
    if int_features:
        x = rng.randint(-2, 3, size=x.shape).astype(np.float32)  # ties everywhere
    cpu = [torch.from_numpy(x), torch.from_numpy(np.asarray(multi.deg_slice)),
           torch.from_numpy(multi.membership)] + \
          [torch.from_numpy(a) for a in multi.get_deg_adjacency_lists()[1:]]
    dev = torch.device("cuda:0")
    g = BatchGraph.from_layer_inputs(cpu[1], cpu[2], cpu[3:], dev)
    g.symmetric = True
    return cpu, g, cpu[0].to(dev), multi.get_num_molecules()


def long_adj(cpu):
    return cpu[:3] + [a.long() for a in cpu[3:]]


@pytest.mark.parametrize("n_feat", [75, 64, 76, 128, 2])
def test_gather_sum(n_feat):
    from deepchem_amd import ops
    cpu, g, x, _ = make_batch(n_feat=n_feat, seed=n_feat)
    s = ops.gather_sum(g, x)
    ref = O.sum_neigh(cpu[0], long_adj(cpu)[3:])
    n0 = g.deg_start[1]
    assert float(s[:n0].abs().max()) == 0.0 if n0 else True
    assert rel(s[n0:], torch.cat(ref, 0)) < TOL
    # accumulate form
    base = torch.ones_like(s)
    s2 = ops.gather_sum(g, x, base.clone(), accumulate=True)
    assert rel(s2, s + 1) < TOL


def test_gather_sum_strided_rows():
    """leading dimension > n_feat (a column slice of a wider matrix)."""
    from deepchem_amd import ops
    cpu, g, x, _ = make_batch(n_feat=64, seed=3)
    wide = torch.zeros((x.shape[0], 96), device=x.device)
    wide[:, 16:80] = x
    s = ops.gather_sum(g, wide[:, 16:80])
    assert rel(s, ops.gather_sum(g, x)) == 0.0


@pytest.mark.parametrize("n_feat", [75, 64])
def test_scatter_add_equals_gather_on_symmetric_graph(n_feat):
    from deepchem_amd import ops
    cpu, g, x, _ = make_batch(n_feat=n_feat, seed=5)
    # autograd of the oracle's gather-sum: d/dx sum(S * ds)
    ds = torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
    xc = cpu[0].clone().requires_grad_(True)
    S = torch.cat([torch.zeros((g.deg_start[1], n_feat))] + O.sum_neigh(xc, long_adj(cpu)[3:]), 0)
    (S * ds).sum().backward()
    dx = torch.zeros_like(x)
    ops.scatter_add(g, ds.cuda(), dx)
    assert rel(dx, xc.grad) < TOL
    assert rel(ops.gather_sum(g, ds.cuda()), xc.grad) < TOL  # bonds listed from both ends


@pytest.mark.parametrize("n_feat,ints", [(75, False), (64, False), (64, True), (7, True)])
def test_gather_max_fwd_bwd(n_feat, ints):
    from deepchem_amd import ops
    cpu, g, x, _ = make_batch(n_feat=n_feat, seed=11, int_features=ints)
    out, arg = ops.gather_max(g, x)
    xc = cpu[0].clone().requires_grad_(True)
    ref = O.graph_pool([xc] + long_adj(cpu)[1:])
    assert rel(out, ref) == 0.0  # max is exact
    dout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
    ref.backward(dout)
    dx = ops.gather_max_bwd(g, dout.cuda(), arg)
    assert rel(dx, xc.grad) < TOL  # first-max tie rule: self, then neighbours in table order


def test_pool_backward_gather_form_equals_atomic_form_and_detects_asymmetry():
    from deepchem_amd import ops
    from deepchem_amd.feat.mol_graphs import ConvMol
    from deepchem_amd.graph import BatchGraph
    dev = torch.device("cuda:0")
    cpu, g, x, _ = make_batch(n_feat=64, seed=23, int_features=True)
    g.symmetric = None  # unknown provenance: must be detected on the device
    out, arg = ops.gather_max(g, x)
    dout = torch.randn(x.shape, generator=torch.Generator().manual_seed(4)).cuda()
    dx_gather = ops.gather_max_bwd(g, dout, arg)
    assert g.symmetric is True and g.rev_pos is not None
    g2 = BatchGraph.from_layer_inputs(cpu[1], cpu[2], cpu[3:], dev)
    g2.symmetric = False  # force the atomic scatter
    dx_atomic = ops.gather_max_bwd(g2, dout, arg)
    assert g2.rev_pos is None
    assert rel(dx_gather, dx_atomic) < 1e-5
    # self-loops with multiplicity (the reference's null molecule, mol_graphs.py:236-254)
    np.random.seed(0)
    null = ConvMol.agglomerate_mols([ConvMol.get_null_mol(8), ConvMol.get_null_mol(8)])
    gn = BatchGraph.from_layer_inputs(torch.from_numpy(np.asarray(null.deg_slice)),
                                      torch.from_numpy(null.membership),
                                      [torch.from_numpy(a) for a in null.get_deg_adjacency_lists()[1:]], dev)
    xn = torch.from_numpy(null.get_atom_features().astype(np.float32)).to(dev)
    on, an = ops.gather_max(gn, xn)
    dn = torch.randn(xn.shape, generator=torch.Generator().manual_seed(5)).cuda()
    d1 = ops.gather_max_bwd(gn, dn, an)
    assert gn.symmetric is True
    assert rel(d1, dn) < 1e-6  # every candidate is the atom itself: the gradient goes to self
    # a one-directional bond: not symmetric -> atomics, still the oracle's answer
    deg_slice = torch.tensor([[0, 1], [1, 1]] + [[2, 0]] * 9)  # atom 0: degree 0, atom 1: degree 1
    adj = [torch.tensor([[0]], dtype=torch.int32)] + [torch.zeros((0, d), dtype=torch.int32) for d in range(2, 11)]
    ga = BatchGraph.from_layer_inputs(deg_slice, torch.tensor([0, 0], dtype=torch.int32), adj, dev)
    xa = torch.tensor([[1.0, 5.0], [2.0, 3.0]]).to(dev)
    oa, aa = ops.gather_max(ga, xa)
    da = ops.gather_max_bwd(ga, torch.ones_like(xa), aa)
    assert ga.symmetric is False
    assert torch.equal(oa.cpu(), torch.tensor([[1.0, 5.0], [2.0, 5.0]]))
    assert torch.equal(da.cpu(), torch.tensor([[1.0, 2.0], [1.0, 0.0]]))


def test_gather_max_with_folded_batchnorm():
    from deepchem_amd import ops
    cpu, g, x, _ = make_batch(n_feat=64, seed=13)
    scale = torch.randn(64).cuda()  # negative scales too: max does not commute with them
    shift = torch.randn(64).cuda()
    out, _ = ops.gather_max(g, x, scale, shift)
    y = cpu[0] * scale.cpu() + shift.cpu()
    ref = O.graph_pool([y] + long_adj(cpu)[1:])
    assert rel(out, ref) < TOL


@pytest.mark.parametrize("n_feat,tanh,ints", [(128, True, False), (75, False, False), (64, False, True)])
def test_readout_fwd_bwd(n_feat, tanh, ints):
    from deepchem_amd import ops
    cpu, g, x, n_mols = make_batch(n_feat=n_feat, seed=17, int_features=ints)
    batch_size = n_mols + 3  # empty molecules at the end: (0, -inf) -> tanh -> (0, -1)
    out, arg = ops.readout(g, x, batch_size, tanh=tanh)
    xc = cpu[0].clone().requires_grad_(True)
    ref = O.graph_gather([xc, cpu[1], cpu[2]], batch_size, activation=torch.tanh if tanh else None)
    assert out.shape == (batch_size, 2 * n_feat)
    if tanh:
        assert rel(out, ref) < TOL
        assert float(out[n_mols:, :n_feat].abs().max()) == 0.0
        assert float((out[n_mols:, n_feat:] + 1).abs().max()) == 0.0
    else:
        assert torch.isneginf(out[n_mols:, n_feat:]).all()
        fin = torch.isfinite(ref)
        assert rel(out.cpu()[fin], ref[fin]) < TOL
    dout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
    dout[n_mols:] = 0
    ref2 = torch.where(torch.isfinite(ref), ref, torch.zeros_like(ref))
    (ref2 * dout).sum().backward()
    dx = ops.readout_bwd(g, dout.cuda(), out, arg, tanh)
    assert rel(dx, xc.grad) < TOL


@pytest.mark.parametrize("n_feat,n_mols,ints", [(128, 60, False), (128, 3000, True), (64, 500, True)])
def test_readout_pipelined_walk_is_bit_identical_to_the_plain_walk(n_feat, n_mols, ints):
    """readout.hip: run bounds in registers, rows walked as one sequence of four-row rounds with two rounds in
    flight (GCMI_OPT_READOUT_PIPELINED, the default) -- the same rows in the same order as the run-by-run walk, so
    sums, maxima and the first-maximum arg-max agree bit for bit; single atoms, degree-10 atoms, empty molecules at
    the end, integer features (ties everywhere)."""
    import ctypes
    from deepchem_amd import _lib, ops
    cpu, g, x, n = make_batch(n_mols=n_mols, n_feat=n_feat, seed=29, int_features=ints)
    was = ctypes.c_int32(-1)
    _lib.call("gcmi_get_option", _lib.GCMI_OPT_READOUT_PIPELINED, ctypes.byref(was))
    assert was.value in (0, 1)  # 1 unless the process was started with GCMI_READOUT_PRE=0
    got = {}
    try:
        for mode in (1, 0):
            _lib.call("gcmi_set_option", _lib.GCMI_OPT_READOUT_PIPELINED, mode)
            out, arg = ops.readout(g, x, n + 5, tanh=False)
            got[mode] = (out.clone(), arg.clone())
    finally:
        _lib.call("gcmi_set_option", _lib.GCMI_OPT_READOUT_PIPELINED, was.value)
    assert torch.equal(got[1][0], got[0][0])
    assert torch.equal(got[1][1], got[0][1])
    # (the oracle's O(N F) segment max: equal to its faithful O(B N F) loop bit for bit, tests/test_oracle_golden.py;
    # the loop would take a minute at 3 000 molecules)
    ref = O.graph_gather(cpu[:3], n + 5, activation=None, faithful=False)
    fin = torch.isfinite(ref)
    assert rel(got[1][0].cpu()[fin], ref[fin]) < TOL


def test_mol_runs_flag_for_unsorted_membership():
    from deepchem_amd.graph import BatchGraph
    dev = torch.device("cuda:0")
    mem = torch.tensor([0, 2, 1], dtype=torch.int32, device=dev)  # not ascending in the block
    g = BatchGraph([3] + [0] * 10, torch.empty(0, dtype=torch.int32, device=dev), mem)
    with pytest.raises(ValueError):
        g.set_mols(3)


@pytest.mark.parametrize("n,f", [(1000, 64), (777, 75), (5, 128), (4096, 128)])
def test_batchnorm_stats_fold_bwd(n, f):
    from deepchem_amd import ops
    gen = torch.Generator().manual_seed(n)
    x = torch.randn((n, f), generator=gen) * 2 + 3
    gamma = torch.rand(f, generator=gen) + 0.5
    beta = torch.randn(f, generator=gen)
    rm, rv = torch.zeros(f), torch.ones(f)
    xc = x.clone().requires_grad_(True)
    gc, bc = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.batch_norm(xc, rm, rv, gc, bc, True, 0.99, 1e-3)
    drm, drv = torch.zeros(f).cuda(), torch.ones(f).cuda()
    mean, invstd, scale, shift = ops.bn_stats(x.cuda(), gamma.cuda(), beta.cuda(), drm, drv, 1e-3, 0.99)
    assert rel(drm, rm) < TOL and rel(drv, rv) < TOL
    yd = ops.bn_apply(x.cuda(), scale, shift)
    assert float((yd.cpu() - y.detach()).abs().max()) < 1e-4
    dy = torch.randn((n, f), generator=gen)
    y.backward(dy)
    dgamma, dbeta, dx = ops.bn_bwd(dy.cuda(), x.cuda(), gamma.cuda(), mean, invstd, True)
    assert rel(dgamma, gc.grad) < TOL and rel(dbeta, bc.grad) < TOL
    assert float((dx.cpu() - xc.grad).abs().max()) < 1e-4 * max(1.0, float(xc.grad.abs().max()))
    _, _, dxm = ops.bn_bwd(dy.cuda(), x.cuda(), gamma.cuda(), mean, invstd, True, relu_mask=True)
    assert torch.equal(dxm, torch.where(x.cuda() > 0, dx, torch.zeros_like(dx)))
    # eval fold
    s2, h2 = ops.bn_fold_eval(gamma.cuda(), beta.cuda(), drm, drv, 1e-3)
    ye = F.batch_norm(x, rm, rv, gamma, beta, False, 0.99, 1e-3)
    assert float((ops.bn_apply(x.cuda(), s2, h2).cpu() - ye).abs().max()) < 1e-4


@pytest.mark.parametrize("k1,k2,n_out,trans,relu", [(75, 75, 64, False, True), (64, 0, 128, True, True),
                                                     (256, 0, 24, True, False), (64, 64, 2, False, False),
                                                     (128, 0, 75, True, False), (33, 7, 1, False, False)])
def test_seg_gemm(k1, k2, n_out, trans, relu):
    from deepchem_amd import ops
    gen = torch.Generator().manual_seed(k1 * 7 + n_out)
    bounds = [0, 5, 5, 130, 131, 400, 1000]  # ragged segments incl. an empty one and a 1-row one
    n = bounds[-1]
    n_seg = len(bounds) - 1
    a1 = torch.randn((n, k1), generator=gen)
    a2 = torch.randn((n, k2), generator=gen) if k2 else None
    w1 = torch.randn((n_seg, n_out, k1) if trans else (n_seg, k1, n_out), generator=gen)
    w2 = torch.randn((n_seg, n_out, k2) if trans else (n_seg, k2, n_out), generator=gen) if k2 else None
    bias = torch.randn((n_seg, n_out), generator=gen)
    skip1 = 2 if k2 else -1  # drop the a1 term of one segment (degree-0 rows have no neighbour sum)
    ref = torch.zeros((n, n_out))
    for s in range(n_seg):
        r = slice(bounds[s], bounds[s + 1])
        acc = bias[s].expand(bounds[s + 1] - bounds[s], n_out).clone()
        if s != skip1:
            acc += a1[r] @ (w1[s].T if trans else w1[s])
        if k2:
            acc += a2[r] @ (w2[s].T if trans else w2[s])
        ref[r] = F.relu(acc) if relu else acc
    w1_off = [(-1 if s == skip1 else s * k1 * n_out) for s in range(n_seg)]
    w2_off = [s * k2 * n_out for s in range(n_seg)]
    out = ops.seg_gemm(bounds[:-1], bounds[1:], a1.cuda(), w1.cuda().reshape(-1), w1_off,
                       a2.cuda() if k2 else None, w2.cuda().reshape(-1) if k2 else None,
                       w2_off if k2 else None, bias.cuda().reshape(-1),
                       [s * n_out for s in range(n_seg)], n_out, trans, relu, n, k1, k2)
    assert rel(out, ref) < TOL


@pytest.mark.parametrize("k,n_cols,trans", [(75, 64, False), (64, 128, True), (256, 24, True), (64, 64, False),
                                            (300, 40, False), (5, 1, True)])
def test_seg_gemm_wgrad(k, n_cols, trans):
    from deepchem_amd import ops
    gen = torch.Generator().manual_seed(k + n_cols)
    bounds = [0, 3, 3, 700, 701, 2100, 5000]
    n = bounds[-1]
    n_seg = len(bounds) - 1
    a = torch.randn((n, k), generator=gen)
    g = torch.randn((n, n_cols), generator=gen)
    dw = torch.zeros((n_seg, n_cols, k) if trans else (n_seg, k, n_cols)).cuda()
    db = torch.zeros((n_seg, n_cols)).cuda()
    ops.seg_gemm_wgrad(bounds[:-1], bounds[1:], a.cuda(), g.cuda(), dw.view(-1),
                       [s * k * n_cols for s in range(n_seg)], db.view(-1),
                       [s * n_cols for s in range(n_seg)], trans)
    for s in range(n_seg):
        r = slice(bounds[s], bounds[s + 1])
        ref = a[r].T @ g[r]
        if trans:
            ref = ref.T
        scale = max(float(ref.abs().max()), 1.0)
        assert float((dw[s].cpu() - ref).abs().max()) / scale < TOL, s
        assert float((db[s].cpu() - g[r].sum(0)).abs().max()) / max(float(g[r].sum(0).abs().max()), 1.0) < TOL


def test_mfma_operand_maps_with_asymmetric_data():
    """A = I-like selector against an asymmetric B: a transposed or permuted
    operand/accumulator map cannot pass this."""
    from deepchem_amd import ops
    n, k, n_out = 64, 64, 64
    a = torch.zeros((n, k))
    for i in range(n):
        a[i, (i * 7 + 3) % k] = 1.0  # row i selects row (7i+3)%64 of W
    w = (torch.arange(k * n_out, dtype=torch.float32).reshape(k, n_out) * 0.01) + \
        torch.arange(n_out, dtype=torch.float32)[None, :] ** 2 * 1e-3
    out = ops.seg_gemm([0], [n], a.cuda(), w.cuda().reshape(-1), [0], None, None, None, None, None, n_out,
                       False, False, n, k, 0)
    ref = a @ w
    assert rel(out, ref) < 1e-6


def test_relu_bwd():
    from deepchem_amd import ops
    y = torch.randn((333, 64)).cuda()
    g = torch.randn((333, 64)).cuda()
    out = ops.relu_bwd_(g.clone(), y)
    assert torch.equal(out, torch.where(y > 0, g, torch.zeros_like(g)))


@pytest.mark.parametrize("kind", [0, 1])
def test_loss_fwd_bwd(kind):
    from deepchem_amd import ops
    gen = torch.Generator().manual_seed(kind)
    b, t, c = 37, 12, 2
    w = (torch.rand((b, t), generator=gen) > 0.17).float()
    if kind == 0:
        logits = torch.randn((b, t, c), generator=gen, requires_grad=True)
        y = F.one_hot((torch.rand((b, t), generator=gen) < 0.1).long(), c).float()
        cfg = O.ModelConfig(t)
    else:
        logits = torch.randn((b, t), generator=gen, requires_grad=True)
        y = torch.randn((b, t), generator=gen)
        cfg = O.ModelConfig(t, mode="regression")
    ref = O.batch_loss(cfg, [logits], y, w)
    ref.backward()
    loss, dlogits, probs = ops.loss_fwd_bwd(kind, logits.detach().cuda(), y.cuda(), w.cuda(), want_probs=True)
    assert abs(float(loss) - float(ref)) < TOL * max(1.0, abs(float(ref)))
    assert rel(dlogits, logits.grad) < TOL
    if kind == 0:
        assert rel(probs, F.softmax(logits.detach(), -1)) < TOL
        assert rel(ops.softmax_lastdim(logits.detach().cuda()), F.softmax(logits.detach(), -1)) < TOL


def test_adam_matches_torch():
    from deepchem_amd import ops
    gen = torch.Generator().manual_seed(0)
    p = torch.randn(1000, generator=gen)
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    dp, m, v = p.clone().cuda(), torch.zeros(1000).cuda(), torch.zeros(1000).cuda()
    for step in range(1, 6):
        g = torch.randn(1000, generator=gen)
        ref_p.grad = g.clone()
        opt.step()
        ops.adam_step_(dp, g.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, step)
    assert float((dp.cpu() - ref_p.detach()).abs().max()) < 2e-6


def test_cpu_tensors_are_refused():
    from deepchem_amd import ops
    from deepchem_amd._lib import GcmiError
    cpu, g, x, _ = make_batch(n_mols=5, high_degree=False)
    with pytest.raises(GcmiError):
        ops.gather_sum(g, cpu[0])


def test_gather_sum_large_against_oracle_and_linearity():
    """Tox21 degree mix at 2^20 atoms (SURVEY 8d R1 scaled to what the oracle does in seconds)."""
    from deepchem_amd import ops
    from deepchem_amd.graph import BatchGraph
    packed = synthetic_molecules(57000, seed=2)
    multi = collate_packed(packed)
    n = multi.get_num_atoms()
    assert n > 1_000_000
    dev = torch.device("cuda:0")
    g = BatchGraph.from_layer_inputs(torch.from_numpy(np.asarray(multi.deg_slice)),
                                     torch.from_numpy(multi.membership),
                                     [torch.from_numpy(a) for a in multi.get_deg_adjacency_lists()[1:]], dev)
    x = torch.randn((n, 64), generator=torch.Generator().manual_seed(0))
    s = ops.gather_sum(g, x.cuda())
    adj = [torch.from_numpy(a).long() for a in multi.get_deg_adjacency_lists()[1:]]
    ref = torch.cat(O.sum_neigh(x, adj), 0)
    assert rel(s[g.deg_start[1]:], ref) < TOL
    # linearity: gather(2x + y) == 2 gather(x) + gather(y)
    y = torch.randn((n, 64), generator=torch.Generator().manual_seed(1)).cuda()
    lhs = ops.gather_sum(g, 2 * x.cuda() + y)
    rhs = 2 * s + ops.gather_sum(g, y)
    assert rel(lhs, rhs) < TOL
    # readout of all-ones counts atoms per molecule
    ones = torch.ones((n, 4), device=dev)
    out, _ = ops.readout(g, ones, packed.n_mols)
    assert torch.equal(out[:, 0].cpu().long(), torch.from_numpy(np.diff(packed.atom_ptr)))


@pytest.mark.parametrize("n_feat,win_cap", [(64, 128), (75, 128), (128, 64), (64, 32), (64, 1000)])
def test_lds_window_kernels_equal_direct_kernels_and_oracle(n_feat, win_cap):
    """The LDS-window forms (gather_lds.hip) of sum_neigh / GraphPool / GraphPool backward on a
    natively collated batch: bit-identical to the direct-from-HBM kernels (same summation order)
    and equal to the oracle."""
    from deepchem_amd import ops
    from deepchem_amd.data.collate import collate_to_device
    packed = concat_packed([synthetic_molecules(300, seed=n_feat, n_feat=n_feat),
                            single_atom_and_edge_cases(n_feat, 1),
                            synthetic_molecules(6, seed=2, n_feat=n_feat, mean_atoms=14, max_atoms=40,
                                                parent_weights=(1,) * 10, ring_deg=10, ring_p_deg3=1.0,
                                                rings_per_atom=0.8)])
    dev = torch.device("cuda:0")
    b = collate_to_device(packed, None, dev, win_cap=win_cap)
    g = b.graph
    assert g.c.n_win > 0 and g.rev_pos is not None
    rng = np.random.RandomState(0)
    xf = rng.randint(-3, 4, size=(g.n_atoms, b.atom_features.shape[1])).astype(np.float32)  # ties
    xf[:, n_feat:] = 0
    x = torch.from_numpy(xf).to(dev)[:, :n_feat] if n_feat % 4 else torch.from_numpy(xf).to(dev)
    xr = torch.from_numpy(rng.standard_normal(xf.shape).astype(np.float32)).to(dev)
    sc = torch.from_numpy(rng.standard_normal(xf.shape[1]).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rng.standard_normal(xf.shape[1]).astype(np.float32)).to(dev)
    dout = torch.from_numpy(rng.standard_normal(xf.shape).astype(np.float32)).to(dev)

    def run():
        s = ops.gather_sum(g, xr)
        s_acc = ops.gather_sum(g, xr, torch.ones_like(xr), accumulate=True)
        o, a = ops.gather_max(g, x)
        o_bn, a_bn = ops.gather_max(g, xr, sc, sh)
        dx = ops.gather_max_bwd(g, dout[:, :o.shape[1]].contiguous(), a)
        return [s, s_acc, o, a, o_bn, a_bn, dx]

    n_win = g.c.n_win
    with_win = run()
    g.c.n_win = 0          # the direct kernels
    direct = run()
    g.c.n_win = n_win
    for w, d in zip(with_win, direct):
        assert torch.equal(w, d)
    # oracle on the same rows (the native collation keeps the reference's atom order)
    multi = collate_packed(packed)
    adjs = [torch.from_numpy(a).long() for a in multi.get_deg_adjacency_lists()[1:]]
    xc = x.cpu()[:, :n_feat].clone()
    ref = O.graph_pool([xc, torch.from_numpy(np.asarray(multi.deg_slice)),
                        torch.from_numpy(multi.membership)] + adjs)
    assert rel(with_win[2][:, :n_feat], ref) == 0.0
    ref_s = torch.cat(O.sum_neigh(xr.cpu(), adjs), 0)
    assert rel(with_win[0][g.deg_start[1]:], ref_s) < TOL

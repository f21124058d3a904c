"""Data-parallel step plumbing on CPU: world_size-2 gloo processes, one all-reduce per step.

KAT (viii) of SURVEY.md section 7: the all-reduced gradient equals the mean of the per-shard
gradients, parameters stay bit-identical across ranks after Adam steps, and FlatAdam follows
Chainer's Adam update rule (train_ddi_modify.py:289).  The model here is the torch-only MLP
link predictor (the HIP encoder needs a GPU); the DP code path is model-agnostic.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bmp.dp import FlatAdam, shard
from bmp.mlp import MLP
from bmp.predictor import sigmoid_cross_entropy


def _data():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 16, generator=g)
    t = (torch.rand(64, 1, generator=g) < 0.3).int()
    return x, t


def _model():
    torch.manual_seed(5)
    return MLP(1, (8, 4), in_dim=16)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, t = _data()
        model = _model()
        if rank == 1:                                # ranks start different; broadcast must fix it
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        opt = FlatAdam(model, alpha=1e-2, weight_decay_rate=1e-3)
        opt.broadcast_parameters(0)
        sl = shard(64, rank, world)
        grads = None
        for step in range(3):
            opt.zero_grad()
            loss = sigmoid_cross_entropy(model(x[sl]), t[sl])
            loss.backward()
            opt.all_reduce_grads()
            if step == 0:
                grads = opt.grad.clone()
            opt.step()
        out[rank] = (grads.numpy(), opt.flat.detach().clone().numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_mean_of_shards():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, p0 = out[0]
    g1, p1 = out[1]
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)          # bit-identical across ranks

    # single-process reference: mean of the two shards' gradients, same Adam
    x, t = _data()
    model = _model()
    ref = FlatAdam(model, alpha=1e-2, weight_decay_rate=1e-3)
    first = None
    for step in range(3):
        acc = torch.zeros_like(ref.grad)
        for r in range(world):
            ref.zero_grad()
            sigmoid_cross_entropy(model(x[shard(64, r, world)]), t[shard(64, r, world)]).backward()
            acc += ref.grad
        ref.grad.copy_(acc / world)
        if step == 0:
            first = ref.grad.clone()
        ref.step()
    assert np.allclose(g0, first.numpy(), rtol=1e-6, atol=1e-8)
    assert np.allclose(p0, ref.flat.detach().numpy(), rtol=1e-5, atol=1e-7)


def test_flat_adam_is_chainer_adam():
    from oracle.ref_cpu import chainer_adam_step
    model = _model()
    opt = FlatAdam(model, alpha=3e-3, weight_decay_rate=1e-2)
    params = [p.detach().clone() for p in model.parameters()]
    state = [dict(m=torch.zeros_like(p), v=torch.zeros_like(p)) for p in params]
    g = torch.Generator().manual_seed(1)
    for t in range(1, 4):
        grads = [torch.randn(p.shape, generator=g) for p in params]
        for p, gr in zip(model.parameters(), grads):
            p.grad.copy_(gr)
        opt.step()
        chainer_adam_step(params, grads, state, t, alpha=3e-3, weight_decay_rate=1e-2)
    for p, q in zip(model.parameters(), params):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)


def test_parameters_are_views_of_the_flat_buffer():
    model = _model()
    opt = FlatAdam(model)
    base = opt.flat.data_ptr()
    off = 0
    for p in model.parameters():
        assert p.data_ptr() == base + 4 * off and p.grad.data_ptr() == opt.grad.data_ptr() + 4 * off
        off += p.numel()
    assert off == opt.flat.numel()


def test_shard_partitions_the_batch():
    assert [shard(8192, r, 8) for r in range(8)] == [slice(1024 * r, 1024 * (r + 1)) for r in range(8)]


def test_functional_mode_matches_per_parameter_grads():
    """One flat gradient via a single autograd node == the per-parameter .grad views."""
    x, t = _data()
    m1, m2 = _model(), _model()
    o1, o2 = FlatAdam(m1, alpha=1e-2), FlatAdam(m2, alpha=1e-2)
    for _ in range(3):
        o1.zero_grad()
        sigmoid_cross_entropy(m1(x), t).backward()
        o1.step()
        loss = sigmoid_cross_entropy(o2.functional_forward(x), t)
        loss.backward()
        o2.collect_grads()
        assert torch.allclose(o1.grad, o2.grad, rtol=1e-6, atol=1e-8)
        o2.step()
    assert torch.allclose(o1.flat, o2.flat, rtol=1e-6, atol=1e-8)
    for p, q in zip(m1.parameters(), m2.parameters()):       # module parameters stay views of the flat buffer
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-8)


# ---- rehearsal of config C5 without hardware: global batch 8192 = 8 ranks x 1024 pairs ----
class _OraclePair(torch.nn.Module):
    """The CPU oracle's pair predictor as an nn.Module (test infrastructure: the HIP encoder needs a GPU; the
    data-parallel plumbing under test -- shard, per-shard padding, FlatAdam, ONE all-reduce -- is model-agnostic)."""

    def __init__(self, hidden=8, layers=1):
        super().__init__()
        from oracle import ref_cpu as O
        p = O.make_pair_params(777, hidden_dim=hidden, out_dim=hidden, n_layers=layers, attn=None, dtype=torch.float32)
        self.names = sorted(p)
        self.params = torch.nn.ParameterList([torch.nn.Parameter(p[n]) for n in self.names])
        self.layers = layers

    def forward(self, a1, j1, a2, j2):
        from oracle import ref_cpu as O
        p = dict(zip(self.names, self.params))
        return O.pair_forward(p, a1, j1, a2, j2, n_layers=self.layers, attn=None)[0]


def _c5_shard_batch(store, idx1, idx2, label, rank, world, G=8192):
    from bmp import synth
    sl = shard(G, rank, world)                                   # pairs [r*G/W, (r+1)*G/W) of the global batch
    a1, j1 = synth.concat_mols([store[k] for k in idx1[sl]])     # padding per rank shard (concat_mols on the shard)
    a2, j2 = synth.concat_mols([store[k] for k in idx2[sl]])
    T = torch.from_numpy
    return T(a1), T(j1), T(a2), T(j2), T(label[sl].reshape(-1, 1))


def _c5_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bmp import synth
        store = synth.make_store()
        idx1, idx2, label = synth.make_pairs(limit=8192)
        a1, j1, a2, j2, t = _c5_shard_batch(store, idx1, idx2, label, rank, world)
        model = _OraclePair()
        if rank:                                                  # ranks start apart; the broadcast must fix it
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.01 * rank)
        opt = FlatAdam(model, alpha=1e-2)
        opt.broadcast_parameters(0)
        n_coll = [0]
        orig = dist.all_reduce

        def counting(*a, **k):
            n_coll[0] += 1
            return orig(*a, **k)
        dist.all_reduce = counting
        first = None
        for step in range(2):
            loss = sigmoid_cross_entropy(opt.functional_forward(a1, j1, a2, j2), t)
            loss.backward()
            opt.collect_grads()
            local = opt.grad.clone()
            opt.all_reduce_grads()
            if step == 0:
                first = (local.numpy(), opt.grad.clone().numpy())
            opt.step()
        dist.all_reduce = orig
        out[rank] = (first[0], first[1], opt.flat.detach().clone().numpy(), n_coll[0], int(a1.shape[1]), int(a2.shape[1]))
    finally:
        dist.destroy_process_group()


def test_c5_partition_eight_ranks_gloo():
    """G = 8192 -> 8 x 1024, padding per rank shard, ONE all-reduce per step, ranks bit-identical, the reduced gradient
    = the mean of the eight shard gradients (not the gradient of a monolithic 8192 batch, whose padding differs)."""
    world = 8
    port = 31500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_c5_worker, args=(world, port, out), nprocs=world, join=True)
    res = [out[r] for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(res[0][1], res[r][1]) and np.array_equal(res[0][2], res[r][2])      # bit-identical ranks
    assert all(r[3] == 2 for r in res)                                # one collective per step, two steps
    mean = np.mean([r[0] for r in res], axis=0)
    assert np.allclose(res[0][1], mean, rtol=1e-5, atol=1e-8)
    # per-shard padding really differs between ranks and from the monolithic batch
    from bmp import synth
    store = synth.make_store()
    idx1, idx2, _ = synth.make_pairs(limit=8192)
    n = np.array([m.n for m in store])
    for r in range(world):
        sl = shard(8192, r, world)
        assert res[r][4] == n[idx1[sl]].max() and res[r][5] == n[idx2[sl]].max()
    assert len({(r[4], r[5]) for r in res}) > 1

"""The data-parallel step on the GPU path: two ranks (gloo collectives, both on cuda:0 -- one MI355X is all a test box
has) run the whole pair model through the layout plan, all-reduce the flat gradient once per step and update with the
fused Adam kernel (which folds the 1/world of the gradient mean in).  Ranks must stay bit-identical, and equal one
process that averages the two shards' gradients itself."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _setup():
    from bmp import packed, synth
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(40, seed=6, n_lo=4, n_hi=30, n_mean=12)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(1)
    i1, i2 = rs.randint(0, 40, 32), rs.randint(0, 40, 32)
    lab = (rs.uniform(size=(32, 1)) < 0.4).astype(np.int32)
    torch.manual_seed(9)
    model = build_pair_predictor(hidden_dim=64, out_dim=32, n_layers=2, attn="nie", head=4).to(dev)
    return dev, ms, i1, i2, lab, model


def _shard_batch(ms, i1, i2, lab, sl, dev, pad):
    from bmp import packed
    return packed.pack_from_store(ms, [i1[sl], i2[sl]], device=dev, pad_to=pad), torch.from_numpy(lab[sl]).to(dev)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bmp.dp import FlatAdam, shard
        dev, ms, i1, i2, lab, model = _setup()
        n = ms.n_atoms
        pad = [int(n[i1].max()), int(n[i2].max())]
        opt = FlatAdam(model, alpha=1e-2)
        opt.broadcast_parameters(0)
        pb, t = _shard_batch(ms, i1, i2, lab, shard(32, rank, world), dev, pad)
        for _ in range(3):
            y = opt.functional_forward(pb)
            model.loss(y, t).backward()
            opt.collect_grads()
            opt.all_reduce_grads()
            opt.step()
        out[rank] = opt.flat.detach().cpu().numpy()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_the_gpu_path():
    world = 2
    port = 29700 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert np.array_equal(out[0], out[1])                       # ranks bit-identical after three steps

    from bmp.dp import FlatAdam, shard
    dev, ms, i1, i2, lab, model = _setup()
    n = ms.n_atoms
    pad = [int(n[i1].max()), int(n[i2].max())]
    ref = FlatAdam(model, alpha=1e-2)
    shards = [_shard_batch(ms, i1, i2, lab, shard(32, r, world), dev, pad) for r in range(world)]
    for _ in range(3):
        acc = None
        for pb, t in shards:
            y = ref.functional_forward(pb)
            model.loss(y, t).backward()
            ref.collect_grads()
            acc = ref.grad.clone() if acc is None else acc + ref.grad
        ref.grad = acc / world
        ref.step()
    got, want = out[0], ref.flat.detach().cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()

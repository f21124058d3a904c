"""The headline configuration at its full size (BASELINE.json configs[1]: 1024 drug pairs of the 544-drug store,
GGNN 4-step d=128 + Nie co-attention + MLP) checked through properties that do not need a dense oracle of that
size, plus an oracle spot check on a handful of its pairs padded the way the full batch pads them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 1024


@pytest.fixture(scope="module")
def world():
    from bmp import packed, synth
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store()
    ms = packed.MolStore(store)
    i1, i2, lab = synth.make_pairs()
    p = O.make_pair_params(777, hidden_dim=128, out_dim=128, n_layers=4, attn="nie", head=8, dtype=torch.float32,
                           bias_scale=0.05)
    model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie", head=8).to(dev)
    load_param_dict(model, p)
    return dict(dev=dev, store=store, ms=ms, i1=i1[:B], i2=i2[:B], lab=lab[:B], p=p, model=model)


def _logits(w, i1, i2, pad_to=None):
    from bmp import packed
    pb = packed.pack_from_store(w["ms"], [i1, i2], device=w["dev"], pad_to=pad_to)
    with torch.no_grad():
        return w["model"](pb), pb


def test_pair_order_is_irrelevant(world):
    """Pairs are independent units: any permutation of the batch (which re-packs every tile) gives every pair the
    same logit up to fp32 summation order (the tile kernels rotate their K walk by the tile index)."""
    w = world
    y, _ = _logits(w, w["i1"], w["i2"])
    perm = np.random.RandomState(1).permutation(B)
    yp, _ = _logits(w, w["i1"][perm], w["i2"][perm])
    ref = y[torch.from_numpy(perm).to(y.device)]
    assert (yp - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1.0)
    assert torch.isfinite(y).all() and y.std().item() > 1e-3


def test_padding_affine_law_full_size(world):
    """The reference sums the readout over padded atoms too (models/ggnn.py:340): one more pad column adds the same
    vector (the pad atom's readout term) to every molecule of the side."""
    from bmp import packed
    w = world
    n = w["ms"].n_atoms
    A1, A2 = int(n[w["i1"]].max()), int(n[w["i2"]].max())
    enc = w["model"].graph_conv
    with torch.no_grad():
        g0 = enc(packed.pack_from_store(w["ms"], [w["i1"], w["i2"]], device=w["dev"], pad_to=[A1, A2]))
        g1 = enc(packed.pack_from_store(w["ms"], [w["i1"], w["i2"]], device=w["dev"], pad_to=[A1 + 1, A2 + 3]))
    diff = g1 - g0
    d1, d2 = diff[:B], diff[B:]
    scale = g0.abs().max().item()
    assert (d1 - d1[0]).abs().max().item() <= 1e-5 * scale
    assert (d2 - 3.0 * d1[0]).abs().max().item() <= 1e-5 * scale          # three extra pad atoms on side 2
    assert d1[0].abs().max().item() > 0


def test_gradient_of_the_batch_is_the_mean_over_shards(world):
    """KAT (viii), at full size on one GPU: with both halves padded like the whole batch, the gradient of the mean
    loss over 1024 pairs is the mean of the two 512-pair gradients (what the data-parallel all-reduce computes)."""
    from bmp import packed
    w = world
    n = w["ms"].n_atoms
    pad = [int(n[w["i1"]].max()), int(n[w["i2"]].max())]
    model = w["model"]
    t = torch.from_numpy(w["lab"].reshape(-1, 1)).to(w["dev"])

    def grad(sl):
        for p in model.parameters():
            p.grad = None
        pb = packed.pack_from_store(w["ms"], [w["i1"][sl], w["i2"][sl]], device=w["dev"], pad_to=pad)
        model.loss(model(pb), t[sl]).backward()
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])

    whole = grad(slice(0, B))
    halves = 0.5 * (grad(slice(0, B // 2)) + grad(slice(B // 2, B)))
    scale = whole.abs().max().item()
    assert (whole - halves).abs().max().item() <= 2e-5 * scale


def test_oracle_spot_check_inside_the_full_batch(world):
    """Six pairs of the 1024, through the dense CPU oracle with the full batch's padding, against their logits in
    the full-batch GPU run (1e-4 relative, the north star's fp32 tolerance)."""
    from oracle import ref_cpu as O
    w = world
    y, _ = _logits(w, w["i1"], w["i2"])
    n = w["ms"].n_atoms
    A1, A2 = int(n[w["i1"]].max()), int(n[w["i2"]].max())
    pick = np.random.RandomState(3).choice(B, 6, replace=False)

    def dense(idx, A):
        atoms = np.zeros((len(idx), A), np.int32); adj = np.zeros((len(idx), 4, A, A), np.float32)
        for b, k in enumerate(idx):
            m = w["store"][k]
            atoms[b, :m.n] = m.atoms; adj[b, :, :m.n, :m.n] = m.dense_adj()
        return torch.from_numpy(atoms), torch.from_numpy(adj)

    a1, j1 = dense(w["i1"][pick], A1)
    a2, j2 = dense(w["i2"][pick], A2)
    p64 = {k: v.double() for k, v in w["p"].items()}
    yo, _, _ = O.pair_forward(p64, a1, j1.double(), a2, j2.double(), n_layers=4, attn="nie")
    got = y[torch.from_numpy(pick).to(y.device)].double().cpu()
    scale = max(yo.abs().max().item(), 1.0)
    assert (got - yo).abs().max().item() <= 1e-4 * scale


def test_full_size_step_is_bitwise_reproducible(world):
    """No atomics anywhere on the path (deterministic split-K slabs, fixed-order reductions, one wavefront per molecule in
    the collate): the same 1024-pair batch, collated twice on the device and stepped twice from the same parameters, gives
    bit-identical logits and a bit-identical flat gradient."""
    from bmp import packed
    from bmp.dp import FlatAdam
    w = world
    ds = packed.DeviceMolStore(w["ms"], w["dev"])
    opt = FlatAdam(w["model"], alpha=0.0)
    outs = []
    for _ in range(2):
        pb, t = packed.pack_from_store_device(ds, [w["i1"], w["i2"]], labels=w["lab"].reshape(-1, 1))
        y = opt.functional_forward(pb)
        w["model"].loss(y, t).backward()
        opt.collect_grads()
        outs.append((y.detach().clone(), opt.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][1].abs().max().item() > 0


def test_many_steps_with_all_streams_match_the_one_stream_run(world, monkeypatch):
    """Twelve end-to-end training steps (device collate on its own stream, weight gradients on the low-priority side stream,
    readout beside the chain, forward as two chains of tiles, Adam) against the same steps with every launch in line on one
    stream: bit-identical parameters at the end -- what a race between iterations (a buffer handed back too early, a missing
    join) would break."""
    from bmp import packed
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    w = world
    rs = np.random.RandomState(11)
    perm = [rs.permutation(len(w["i1"])) for _ in range(3)]

    def run(one_stream):
        monkeypatch.setenv("BMP_ONE_STREAM", "1" if one_stream else "0")
        monkeypatch.setenv("BMP_COLLATE_STREAM", "0" if one_stream else "1")
        model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie", head=8).to(w["dev"])
        load_param_dict(model, w["p"])
        ds = packed.DeviceMolStore(w["ms"], w["dev"])
        opt = FlatAdam(model, alpha=1e-3)
        losses = []
        for it in range(12):
            p = perm[it % 3]
            sl = slice((it % 4) * 256, (it % 4) * 256 + 256)
            pb, t = packed.pack_from_store_device(ds, [w["i1"][p][sl], w["i2"][p][sl]], labels=w["lab"][p][sl].reshape(-1, 1))
            y = opt.functional_forward(pb)
            loss = model.loss(y, t)
            loss.backward()
            opt.collect_grads()
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        assert (opt.plan.side is None) == one_stream and (ds.stream is None) == one_stream
        return opt.flat.clone(), torch.stack(losses)

    p_multi, l_multi = run(False)
    p_one, l_one = run(True)
    assert torch.equal(l_multi, l_one) and torch.equal(p_multi, p_one)
    assert torch.isfinite(p_one).all() and (l_one[-1] < l_one[0]).item()


def test_planned_no_grad_forward_with_two_chains_is_the_one_chain_forward(world):
    """The planned forward under ``torch.no_grad()`` (predict / evaluators) with the encoder's tile-local launches split over
    two streams: nothing but the plan's state holds the step tensors handed to the second stream once each Function has
    returned (no autograd context), so the state must keep them until the join -- otherwise the caching allocator reuses
    their blocks for the next main-stream launch while the part stream still reads or writes them.  Bit-identical logits
    to the same forward with every launch on one stream, over several rounds with allocator churn in between."""
    from bmp import packed
    from bmp.dp import FlatAdam
    w = world
    opt = FlatAdam(w["model"], alpha=0.0)
    pb = packed.pack_from_store(w["ms"], [w["i1"], w["i2"]], device=w["dev"])
    assert pb.n_tiles >= 64
    with torch.no_grad():
        y0 = opt.functional_forward(pb).clone()
    plan = opt.plan
    assert plan is not None and plan.split is not None
    outs = []
    for rep in range(4):
        with torch.no_grad():
            y = opt.functional_forward(pb)
            junk = [torch.full((pb.n_rows, 128), float(rep), device=w["dev"]) for _ in range(3)]     # takes freed blocks at once
            outs.append(y.clone())
            del junk
    saved_split, saved_side = plan.split, plan.side
    plan.split = plan.side = None
    try:
        with torch.no_grad():
            y_one = opt.functional_forward(pb).clone()
    finally:
        plan.split, plan.side = saved_split, saved_side
    torch.cuda.synchronize()
    assert torch.isfinite(y_one).all()
    for y in [y0] + outs:
        assert torch.equal(y, y_one)


def test_planned_predict_is_the_training_forward_and_matches_the_oracle(world):
    """The evaluation callers' predict (eval_coattention.py:103-124: logits and the two molecule vectors under no-backprop)
    on the planned path at full size: the kernels keep nothing for a backward (no m / r|z / c / ij / C stores), yet the
    logits are bit for bit those of the training-mode forward; six pairs against the oracle's logits and molecule vectors."""
    from bmp import packed
    from bmp.dp import FlatAdam
    from oracle import ref_cpu as O
    from parity_util import close
    w = world
    opt = FlatAdam(w["model"], alpha=0.0)
    pb = packed.pack_from_store(w["ms"], [w["i1"], w["i2"]], device=w["dev"])
    y_train = opt.functional_forward(pb).detach().clone()
    y, (g1, g2) = opt.functional_predict(pb)
    torch.cuda.synchronize()
    assert not y.requires_grad and torch.equal(y, y_train)
    y2, _ = opt.functional_predict(pb)                      # again: nothing stale
    assert torch.equal(y2, y)
    n = w["ms"].n_atoms
    A1, A2 = int(n[w["i1"]].max()), int(n[w["i2"]].max())
    pick = np.random.RandomState(4).choice(B, 6, replace=False)

    def dense(idx, A):
        atoms = np.zeros((len(idx), A), np.int32); adj = np.zeros((len(idx), 4, A, A), np.float32)
        for b, k in enumerate(idx):
            m = w["store"][k]
            atoms[b, :m.n] = m.atoms; adj[b, :, :m.n, :m.n] = m.dense_adj()
        return torch.from_numpy(atoms), torch.from_numpy(adj)

    a1, j1 = dense(w["i1"][pick], A1)
    a2, j2 = dense(w["i2"][pick], A2)
    p64 = {k: v.double() for k, v in w["p"].items()}
    yo, g1o, g2o = O.pair_forward(p64, a1, j1.double(), a2, j2.double(), n_layers=4, attn="nie")
    sel = torch.from_numpy(pick).to(y.device)
    close(y[sel], yo, "predict logits"); close(g1[sel], g1o, "predict g1"); close(g2[sel], g2o, "predict g2")

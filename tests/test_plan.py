"""Layout plan (bmp/plan.py): the two gather tables against the modules' own layout code.

CPU: the tables are derived by evaluating prepared_layouts() / primary_layouts() / primary_grads() on
index-valued stand-ins; here they are replayed on random parameters and random kernel-gradient buffers and
must reproduce (i) the layout functions exactly and (ii) autograd's parameter gradients of the same functions.
GPU: a whole training step through the plan equals the eager autograd step (same kernels underneath).
"""
import numpy as np
import pytest
import torch

from bmp.coattention import NieFineCoattention, PoolingFineCoattention
from bmp.ggnn import GGNN
from bmp.plan import LayoutPlan, gather_sum_host


def _flat_of(mods):
    names, shapes, chunks = [], [], []
    for prefix, m in mods:
        for n, p in m.named_parameters():
            names.append(prefix + n); shapes.append(tuple(p.shape)); chunks.append(p.detach().reshape(-1))
    return names, shapes, torch.cat(chunks)


@pytest.mark.parametrize("tying,n_layers", [(True, 4), (False, 3), (True, 1)])
def test_plan_tables_match_layout_code(tying, n_layers):
    torch.manual_seed(11)
    enc = GGNN(out_dim=12, hidden_dim=8, n_layers=n_layers, weight_tying=tying)
    att = NieFineCoattention(hidden_dim=8, out_dim=12, head=3, activation="tanh")
    pool = PoolingFineCoattention(hidden_dim=8, out_dim=12)
    from bmp.relgcn import RelGCN
    rel = RelGCN(out_channels=12, ch_list=[8, 16, 8])
    rel2 = RelGCN(out_channels=12, ch_list=[64, 64])             # its layer takes the fused kernel's layouts
    with torch.no_grad():
        for m in (enc, att, pool, rel, rel2):
            for p in m.parameters():
                p.copy_(torch.randn_like(p))
    mods = [("graph_conv.", enc), ("attn.", att), ("pool.", pool), ("rel.", rel), ("rel2.", rel2)]
    names, shapes, flat = _flat_of(mods)
    plan = LayoutPlan(mods, names, shapes, "cpu")
    # (i) prepare == the layout functions, bit for bit
    plan.prepare(flat)
    for prefix, m in mods:
        with torch.no_grad():
            ref = m.prepared_layouts()
        assert set(ref) == set(plan.P[prefix])
        for k, v in ref.items():
            assert torch.equal(plan.P[prefix][k], v.contiguous()), (prefix, k)
    # (ii) collect == autograd through primary_layouts with the kernels' buffers as upstream gradients
    plan.gk.copy_(torch.randn_like(plan.gk))
    got = torch.zeros_like(flat)
    plan.collect(got)
    want = []
    for prefix, m in mods:
        params = list(m.parameters())
        prim = m.primary_layouts()
        pg = m.primary_grads(plan.G[prefix])
        loss = sum((prim[k] * t).sum() for k in prim for t in pg[k])
        gs = torch.autograd.grad(loss, params, allow_unused=True)
        want += [(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs, params)]
    want = torch.cat(want)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6), (got - want).abs().max()


def test_gather_sum_host_semantics():
    src = torch.tensor([1.0, 2.0, 4.0])
    tab = np.array([[0, 2, -1], [1, -1, -1]], dtype=np.int32)
    assert gather_sum_host(src, tab).tolist() == [3.0, 4.0, 0.0]


@pytest.mark.gpu
@pytest.mark.parametrize("attn,encoder", [("nie", "ggnn"), ("pool", "ggnn"), ("nie", "relgcn"), ("parallel", "relgcn")])
def test_planned_step_equals_eager_step(attn, encoder):
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(40, seed=3, n_lo=4, n_hi=40, n_mean=14)
    ms = packed.MolStore(store)
    i1, i2 = np.arange(0, 16), np.arange(16, 32)
    pb = packed.pack_from_store(ms, [i1, i2], device=dev)
    t = (torch.arange(16, device=dev) % 2).int().view(-1, 1)
    torch.manual_seed(1)
    model = build_pair_predictor(hidden_dim=64, out_dim=32, n_layers=3, attn=attn, head=4 if attn != "parallel" else 1,
                                 encoder=encoder).to(dev)
    # eager: module parameters, autograd through every layout op
    y = model(pb)
    model.loss(y, t).backward()
    # (the fine co-attention ignores g_1 / g_2, nie_coattention.py:335-370: the readout gets no gradient)
    eager = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()]).clone()
    y_eager = y.detach().clone()
    # planned: flat buffer, two gather launches
    opt = FlatAdam(model, alpha=1e-3)
    y2 = opt.functional_forward(pb)
    assert opt.plan is not None and "graph_conv." in opt.plan.P and (attn == "parallel" or "attn." in opt.plan.P)
    model.loss(y2, t).backward()
    opt.collect_grads()
    assert torch.equal(y2.detach(), y_eager)
    scale = eager.abs().max().item()
    assert (opt.grad - eager).abs().max().item() <= 1e-5 * scale
    # and the fused Adam kernel against the update rule written out
    p0, g = opt.flat.clone(), opt.grad.clone()
    opt.step()
    m = 0.1 * g; v = 0.001 * g * g
    a_t = 1e-3 * (1 - 0.999) ** 0.5 / (1 - 0.9)
    want = p0 - a_t * m / (v.sqrt() + 1e-8)
    assert torch.allclose(opt.flat, want, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_planned_step_with_the_reference_four_array_call():
    """The reference's call form runs the encoder once per side: the plan's gradient buffers must accumulate over both
    calls (embedding, readout, step groups), and nothing stale may survive from the step before."""
    from bmp import synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(30, seed=8, n_lo=4, n_hi=30, n_mean=10)
    a1, j1 = synth.concat_mols([store[k] for k in range(0, 10)])
    a2, j2 = synth.concat_mols([store[k] for k in range(10, 20)])
    t = (torch.arange(10, device=dev) % 2).int().view(-1, 1)
    torch.manual_seed(2)
    model = build_pair_predictor(hidden_dim=64, out_dim=32, n_layers=2, attn="parallel", head=1).to(dev)   # uses g1, g2: readout trains
    y = model(a1, j1, a2, j2)
    model.loss(y, t).backward()
    eager = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()]).clone()
    opt = FlatAdam(model, alpha=1e-3)
    for _ in range(2):                                   # twice: the second step must not see the first one's buffers
        y2 = opt.functional_forward(a1, j1, a2, j2)
        model.loss(y2, t).backward()
        opt.collect_grads()
        scale = eager.abs().max().item()
        assert (opt.grad - eager).abs().max().item() <= 1e-5 * scale
    assert opt.plan is not None and "graph_conv." in opt.plan.P


@pytest.mark.gpu
def test_graphed_training_steps_equal_eager_steps():
    """bmp.dp.GraphedTrainStep: a step recorded as a HIP graph and replayed leaves the same parameters as the eager
    step (same kernels, same order), over several steps and two alternating batches."""
    from bmp import packed, synth
    from bmp.dp import FlatAdam, GraphedTrainStep
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(40, seed=3, n_lo=4, n_hi=40, n_mean=14)
    ms = packed.MolStore(store)
    batches = []
    for k in range(2):
        i1, i2 = np.arange(16 * k, 16 * k + 8), np.arange(16 * k + 8, 16 * k + 16)
        batches.append((packed.pack_from_store(ms, [i1, i2], device=dev), (torch.arange(8, device=dev) % 2).int().view(-1, 1)))

    def run(graphed):
        torch.manual_seed(4)
        model = build_pair_predictor(hidden_dim=64, out_dim=32, n_layers=2, attn="nie", head=4).to(dev)
        opt = FlatAdam(model, alpha=1e-2)
        stepper = GraphedTrainStep(model, opt) if graphed else None
        losses = []
        for s in range(6):
            pb, t = batches[s % 2]
            if graphed:
                losses.append(float(stepper(pb, t).detach()))
            else:
                y = opt.functional_forward(pb)
                loss = model.loss(y, t)
                loss.backward()
                opt.collect_grads(); opt.all_reduce_grads(); opt.step()
                losses.append(float(loss.detach()))
        return opt.flat.clone(), losses, opt.t

    p_eager, l_eager, t_eager = run(False)
    p_graph, l_graph, t_graph = run(True)
    assert t_eager == t_graph == 6
    assert np.allclose(l_eager, l_graph, rtol=1e-5, atol=1e-6), (l_eager, l_graph)
    assert (p_eager - p_graph).abs().max().item() <= 1e-5 * p_eager.abs().max().item()
    assert l_eager[-1] < l_eager[0]


@pytest.mark.gpu
@pytest.mark.parametrize("encoder,hidden,attn", [("ggnn", 64, "nie"), ("relgcn", 64, "nie"), ("ggnn", 40, None)])
def test_side_stream_step_is_bit_identical_to_the_one_stream_step(encoder, hidden, attn):
    """The weight-gradient launches of the planned backward go to a low-priority side stream (bmp/plan.py SideStream; fused
    step / layer kernels at hidden 64, the bmp_gru_bwd / bmp_msg_bwd / bmp_readout_bwd stream_w form at hidden 40): same
    launches, same order inside every gradient buffer -> the same bits as with everything in line on one stream."""
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(60, seed=5, n_lo=4, n_hi=50, n_mean=18)
    ms = packed.MolStore(store)
    i1, i2 = np.arange(0, 24), np.arange(24, 48)
    pb = packed.pack_from_store(ms, [i1, i2], device=dev)
    t = (torch.arange(24, device=dev) % 2).int().view(-1, 1)
    torch.manual_seed(2)
    model = build_pair_predictor(hidden_dim=hidden, out_dim=hidden, n_layers=3, attn=attn, head=4, encoder=encoder).to(dev)
    opt = FlatAdam(model, alpha=0.0)
    outs = []
    for side_on in (True, False, True):
        y = opt.functional_forward(pb)
        assert opt.plan is not None and opt.plan.side is not None
        if not side_on:
            saved, opt.plan.side = opt.plan.side, None
            y = opt.functional_forward(pb)          # prepare() again: the step's state must not carry the stream
        model.loss(y, t).backward()
        used = bool(opt.plan.state.get("side_used"))
        opt.collect_grads()
        if not side_on:
            opt.plan.side = saved
        assert used == side_on
        outs.append((y.detach().clone(), opt.grad.clone()))
    torch.cuda.synchronize()
    for y, g in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(g, outs[0][1])
    assert outs[0][1].abs().max().item() > 0


@pytest.mark.gpu
def test_readout_taken_off_the_chain_still_gives_the_molecule_vectors():
    """A fine co-attention never reads g_1 / g_2 (nie_coattention.py:335-370): the pair predictor tells a planned encoder,
    which then computes its readout beside the chain.  After the join the vectors are the eager encoder's; differentiating
    them is an error, not a silent zero."""
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(40, seed=7, n_lo=4, n_hi=40, n_mean=14)
    ms = packed.MolStore(store)
    pb = packed.pack_from_store(ms, [np.arange(0, 16), np.arange(16, 32)], device=dev)
    torch.manual_seed(3)
    model = build_pair_predictor(hidden_dim=64, out_dim=64, n_layers=2, attn="nie", head=4).to(dev)
    with torch.no_grad():
        g_eager = model.graph_conv(pb).clone()
    opt = FlatAdam(model, alpha=0.0)
    seen = {}
    enc_forward = model.graph_conv.forward

    def spy(*a, **k):
        seen["g"] = enc_forward(*a, **k)
        seen["off"] = model.graph_conv._readout_off_chain
        return seen["g"]
    model.graph_conv.forward = spy
    try:
        y = opt.functional_forward(pb)
    finally:
        model.graph_conv.forward = enc_forward
    # (beside the chain = on the forward's second-chain stream, idle once the encoder is through; BMP_READOUT_STREAM=side: on the
    #  weight-gradient stream)
    assert seen["off"] is True and (opt.plan.state.get("split_open") or opt.plan.state.get("side_used"))
    opt.plan.split.join()
    opt.plan.side.join()
    torch.cuda.synchronize()
    assert torch.equal(seen["g"].detach(), g_eager)
    with pytest.raises(RuntimeError, match="off_chain"):
        seen["g"].sum().backward(retain_graph=True)
    y.sum().backward()                      # the chain itself is unaffected
    opt.collect_grads()
    assert torch.isfinite(opt.grad).all()


@pytest.mark.gpu
def test_coattention_backward_clears_what_its_pair_kernels_do_not_write():
    """bmp_coattn_nie_bwd with the packed batches' row -> molecule maps clears only the dead rows and the padding columns of
    its work arrays (instead of filling all of them): with the allocator's free blocks poisoned with NaN beforehand, the
    planned gradients are still the eager ones."""
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(60, seed=9, n_lo=4, n_hi=60, n_mean=20)
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, dev)
    i1, i2 = np.arange(0, 30), np.arange(30, 60)
    pb = packed.pack_from_store_device(ds, [i1, i2])
    assert pb.row_mol is not None and int((pb.row_mol < 0).sum()) > 0          # there are dead rows to clear
    t = (torch.arange(30, device=dev) % 2).int().view(-1, 1)
    torch.manual_seed(4)
    model = build_pair_predictor(hidden_dim=64, out_dim=64, n_layers=2, attn="nie", head=4).to(dev)
    y = model(pb)
    model.loss(y, t).backward()
    eager = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()]).clone()
    opt = FlatAdam(model, alpha=0.0)
    for _ in range(2):
        poison = [torch.full((n,), float("nan"), device=dev) for n in (1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14)]
        del poison                                                   # back to the allocator, NaN inside
        y2 = opt.functional_forward(pb)
        model.loss(y2, t).backward()
        opt.collect_grads()
        assert torch.isfinite(opt.grad).all()
        assert (opt.grad - eager).abs().max().item() <= 1e-5 * eager.abs().max().item()

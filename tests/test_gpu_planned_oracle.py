"""The path bench.py times -- FlatAdam.functional_forward (layout plan, prepared weights, PStepFn / PNieFn /
PRelLayerFn), collect_grads, bmp_adam_step -- against the dense float64 oracle, at the widths of BASELINE.json's configs:
C2 (GGNN 4-step d=128 tied + Nie, head 8) and C3 as composed (RelGCN 3x128, scale_adj, atoms tap + Nie).  Logits, loss, the
flat gradient (every parameter) and the parameters after one Adam step (oracle.chainer_adam_step) within 1e-4 of the
tensor's max-abs (the north star's fp32 tolerance).  Parity unpinned: the oracle is a restatement, SURVEY.md 8(c)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = 1e-4


from parity_util import close as _close      # asserts AND logs the achieved relative error


def _oracle_step(p, batch, encoder, n_layers, attn, alpha, **fwd_kw):
    from oracle import ref_cpu as O
    a1, j1, a2, j2, lab = batch
    p = {k: v.clone().requires_grad_() for k, v in p.items()}
    y, _, _ = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), encoder=encoder, n_layers=n_layers, attn=attn, **fwd_kw)
    loss = O.sigmoid_cross_entropy(y, T(lab))
    names = sorted(p)
    grads = torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(p[n])) for n, g in zip(names, grads)}
    newp = {n: p[n].detach().clone() for n in names}
    state = [dict(m=torch.zeros_like(newp[n]), v=torch.zeros_like(newp[n])) for n in names]
    O.chainer_adam_step([newp[n] for n in names], [grads[n] for n in names], state, 1, alpha=alpha)
    return y.detach(), loss.detach(), grads, newp


@pytest.mark.parametrize("encoder,n_layers", [("ggnn", 4), ("relgcn", 3)])
def test_planned_training_step_matches_oracle_at_d128(encoder, n_layers):
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(48, seed=11, n_lo=2, n_hi=44, n_mean=16)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(5)
    B = 14
    i1, i2 = rs.randint(0, 48, B), rs.randint(0, 48, B)
    lab = rs.randint(0, 2, (B, 1)).astype(np.int32)
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    p = O.make_pair_params(777, encoder=encoder, hidden_dim=128, out_dim=128, n_layers=n_layers, attn="nie", head=8,
                           dtype=torch.float64, bias_scale=0.05)
    alpha = 1e-2
    y_o, loss_o, g_o, p_o = _oracle_step(p, (a1, j1, a2, j2, lab), encoder, n_layers, "nie", alpha)

    model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=n_layers, attn="nie", head=8, encoder=encoder).to(dev)
    load_param_dict(model, p)
    opt = FlatAdam(model, alpha=alpha)
    ds = packed.DeviceMolStore(ms, dev)
    pb, t = packed.pack_from_store_device(ds, [i1, i2], labels=lab)          # the bench's collate
    y = opt.functional_forward(pb)
    assert opt.plan is not None and {"graph_conv.", "attn."} <= set(opt.plan.P)        # the planned path really ran
    loss = model.loss(y, t)
    loss.backward()
    opt.collect_grads()
    _close(y, y_o, "logits"); _close(loss, loss_o, "loss")
    off = 0
    for name, shp in zip(opt.names, opt.shapes):
        n = int(np.prod(shp))
        _close(opt.grad[off:off + n].view(shp), g_o[name.replace(".", "/")], f"grad {name}")
        off += n
    opt.step()
    off = 0
    for name, shp in zip(opt.names, opt.shapes):
        n = int(np.prod(shp))
        key = name.replace(".", "/")
        # the update is alpha_t * m / (sqrt(v) + eps) ~ alpha * sign(g): compare the UPDATE, not the parameter
        upd_o = p_o[key] - p[key]
        upd = opt.flat[off:off + n].view(shp).double().cpu() - p[key].float().double()
        big = g_o[key].abs() > 1e-3 * g_o[key].abs().max().clamp(min=1e-30)           # sign(g) is ill-conditioned at g ~ 0
        if big.any():
            assert (upd - upd_o)[big].abs().max().item() <= 1e-3 * alpha, name
        off += n


@pytest.mark.parametrize("layout", ["instance", "encoder"])
@pytest.mark.parametrize("sim_method", ["ntn", "hole"])
def test_planned_reference_headline_model_matches_oracle(sim_method, layout):
    """The model every figure of the reference was trained with (DDI.md:6, RECORD.txt:246-251; train_binary.py:165-187,226-227):
    GGNN hidden 32, 8 propagation steps, weight_tying=False, fp_out_dim 16 (the script's default) + NieFineCoattention(32, 16,
    head=8, tanh) + NTN / HolE with --net-hidden-dims= (no hidden layer), batch 32 -- through the PLANNED path
    (functional_loss: the d = 32 fused step kernels of bmp_fused_small.hip, the pair kernels, the link predictor's kernels),
    per-instance batches and the encoder layout (tile table: blocks of 1..4 live 32-row blocks), against the dense float64
    oracle: logits, loss, every gradient at 2e-5 of the tensor's max-abs."""
    from bmp import enclayout, packed, synth
    from bmp import functional as Fn
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(60, seed=23, n_lo=3, n_hi=70, n_mean=22)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(8)
    B = 32
    i1, i2 = rs.randint(0, 60, B), rs.randint(0, 60, B)
    lab = rs.randint(0, 2, (B, 1)).astype(np.int32)
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    kw = dict(hidden_dim=32, out_dim=16, n_layers=8, weight_tying=False, attn="nie", head=8, sim_method=sim_method, mlp_hidden=())
    p = O.make_pair_params(777, encoder="ggnn", dtype=torch.float64, bias_scale=0.05, **kw)
    y_o, loss_o, g_o, _ = _oracle_step(p, (a1, j1, a2, j2, lab), "ggnn", 8, "nie", 1e-3, weight_tying=False, sim_method=sim_method,
                                       mlp_hidden=0)
    model = build_pair_predictor(encoder="ggnn", **kw).to(dev)
    load_param_dict(model, p)
    opt = FlatAdam(model, alpha=1e-3)
    ds = packed.DeviceMolStore(ms, dev)
    if layout == "encoder":
        pb, t = enclayout.encode_from_store_device(ds, [i1, i2], labels=lab)
    else:
        pb, t = packed.pack_from_store_device(ds, [i1, i2], labels=lab)
    assert Fn.step_supported(32)
    for rep in range(2):
        loss = opt.functional_loss(pb, t=t)
        assert opt.plan is not None and {"graph_conv.", "attn."} <= set(opt.plan.P) and model.graph_conv._plan_fused()
        loss.backward()
        opt.collect_grads()
        _close(model.y, y_o, f"logits ({sim_method}, {layout}, rep {rep})", tol=2e-5)
        _close(loss, loss_o, f"loss ({sim_method}, {layout}, rep {rep})", tol=2e-5)
        off = 0
        for name, shp in zip(opt.names, opt.shapes):
            n = int(np.prod(shp))
            _close(opt.grad[off:off + n].view(shp), g_o[name.replace(".", "/")], f"grad {name} ({sim_method}, {layout}, rep {rep})",
                   tol=2e-5)
            off += n


def test_planned_c4_step_matches_oracle_at_d256():
    """Config C4's model (GGNN d = 256, no co-attention, 37 multi-hot classes, train_ggnn_hole_multi_class_x37.py:71-91) on the
    planned path of the UNFUSED operators (the widths the fused step kernels do not cover: PMsgFn / PGRUFn on prepared
    weights): logits, loss and the flat gradient against the oracle."""
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(40, seed=4, n_lo=3, n_hi=36, n_mean=14)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(2)
    B, C = 10, 37
    i1, i2 = rs.randint(0, 40, B), rs.randint(0, 40, B)
    lab = (rs.uniform(size=(B, C)) < 0.08).astype(np.int32)
    lab[1, 5] = -1
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    p = O.make_pair_params(777, hidden_dim=256, out_dim=256, n_layers=4, attn=None, class_num=C, dtype=torch.float64, bias_scale=0.05)
    y_o, loss_o, g_o, _ = _oracle_step(p, (a1, j1, a2, j2, lab), "ggnn", 4, None, 1e-3)
    model = build_pair_predictor(hidden_dim=256, out_dim=256, n_layers=4, attn=None, class_num=C).to(dev)
    load_param_dict(model, p)
    opt = FlatAdam(model, alpha=1e-3)
    pb, t = packed.pack_from_store_device(packed.DeviceMolStore(ms, dev), [i1, i2], labels=lab)
    for _ in range(2):                                   # twice: nothing stale may survive from the step before
        y = opt.functional_forward(pb)
        assert opt.plan is not None and "graph_conv." in opt.plan.P and "msg0.WT" in opt.plan.P["graph_conv."]
        loss = model.loss(y, t)
        loss.backward()
        opt.collect_grads()
        _close(y, y_o, "logits"); _close(loss, loss_o, "loss")
        off = 0
        for name, shp in zip(opt.names, opt.shapes):
            n = int(np.prod(shp))
            _close(opt.grad[off:off + n].view(shp), g_o[name.replace(".", "/")], f"grad {name}")
            off += n


def test_c3_composed_matches_dense_oracle_eager_at_d128():
    """build_pair_predictor(encoder='relgcn', hidden_dim=128, n_layers=3, attn='nie') (models/relgcn.py:61-73 +
    nie_coattention.py:335-370), module call (eager path): logits, loss and every parameter gradient."""
    from bmp import packed, synth
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import grad_dict, load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(40, seed=13, n_lo=1, n_hi=50, n_mean=18)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(6)
    B = 10
    i1, i2 = rs.randint(0, 40, B), rs.randint(0, 40, B)
    lab = rs.randint(0, 2, (B, 1)).astype(np.int32)
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    p = O.make_pair_params(777, encoder="relgcn", hidden_dim=128, out_dim=128, n_layers=3, attn="nie", head=8,
                           dtype=torch.float64, bias_scale=0.05)
    y_o, loss_o, g_o, _ = _oracle_step(p, (a1, j1, a2, j2, lab), "relgcn", 3, "nie", 1e-3)
    model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=3, attn="nie", head=8, encoder="relgcn").to(dev)
    load_param_dict(model, p)
    for form in ("packed", "four arrays"):
        model.zero_grad()
        if form == "packed":
            y = model(packed.pack_from_store(ms, [i1, i2], device=dev))
        else:                                                       # the reference's call form, train_binary.py:84-96
            y = model(a1, j1, a2, j2)
        loss = model.loss(y, T(lab).to(dev))
        loss.backward()
        _close(y, y_o, f"logits ({form})"); _close(loss, loss_o, f"loss ({form})")
        for name, gr in grad_dict(model).items():
            _close(gr, g_o[name], f"grad {name} ({form})")


# ---- C3 at BASELINE.json's full batch size: properties that need no dense oracle of that size ----
B_FULL = 1024


@pytest.fixture(scope="module")
def c3():
    from bmp import packed, synth
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store()
    ms = packed.MolStore(store)
    i1, i2, lab = synth.make_pairs()
    p = O.make_pair_params(777, encoder="relgcn", hidden_dim=128, out_dim=128, n_layers=3, attn="nie", head=8,
                           dtype=torch.float32, bias_scale=0.05)
    model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=3, attn="nie", head=8, encoder="relgcn").to(dev)
    load_param_dict(model, p)
    return dict(dev=dev, store=store, ms=ms, i1=i1[:B_FULL], i2=i2[:B_FULL], lab=lab[:B_FULL], p=p, model=model)


def test_c3_full_size_pair_order_is_irrelevant(c3):
    from bmp import packed
    w = c3
    with torch.no_grad():
        y = w["model"](packed.pack_from_store(w["ms"], [w["i1"], w["i2"]], device=w["dev"]))
        perm = np.random.RandomState(1).permutation(B_FULL)
        yp = w["model"](packed.pack_from_store(w["ms"], [w["i1"][perm], w["i2"][perm]], device=w["dev"]))
    ref = y[T(perm).to(y.device)]
    assert (yp - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1.0)
    assert torch.isfinite(y).all() and y.std().item() > 1e-4


def test_c3_full_size_gradient_is_the_mean_over_shards(c3):
    from bmp import packed
    w = c3
    n = w["ms"].n_atoms
    pad = [int(n[w["i1"]].max()), int(n[w["i2"]].max())]
    model = w["model"]
    t = T(w["lab"].reshape(-1, 1)).to(w["dev"])

    def grad(sl):
        for q in model.parameters():
            q.grad = None
        pb = packed.pack_from_store(w["ms"], [w["i1"][sl], w["i2"][sl]], device=w["dev"], pad_to=pad)
        model.loss(model(pb), t[sl]).backward()
        return torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1) for q in model.parameters()])

    whole = grad(slice(0, B_FULL))
    halves = 0.5 * (grad(slice(0, B_FULL // 2)) + grad(slice(B_FULL // 2, B_FULL)))
    assert (whole - halves).abs().max().item() <= 2e-5 * whole.abs().max().item()


def test_c3_full_size_oracle_spot_check(c3):
    """Six pairs of the 1024 through the dense oracle with the full batch's padding (RelGCN's readout sums over padded
    positions too, models/relgcn.py:72-73) against their logits in the full-batch GPU run."""
    from bmp import packed
    from oracle import ref_cpu as O
    w = c3
    with torch.no_grad():
        y = w["model"](packed.pack_from_store(w["ms"], [w["i1"], w["i2"]], device=w["dev"]))
    n = w["ms"].n_atoms
    A1, A2 = int(n[w["i1"]].max()), int(n[w["i2"]].max())
    pick = np.random.RandomState(3).choice(B_FULL, 6, replace=False)

    def dense(idx, A):
        atoms = np.zeros((len(idx), A), np.int32); adj = np.zeros((len(idx), 4, A, A), np.float32)
        for b, k in enumerate(idx):
            m = w["store"][k]
            atoms[b, :m.n] = m.atoms; adj[b, :, :m.n, :m.n] = m.dense_adj()
        return T(atoms), T(adj)

    a1, j1 = dense(w["i1"][pick], A1)
    a2, j2 = dense(w["i2"][pick], A2)
    p64 = {k: v.double() for k, v in w["p"].items()}
    yo, _, _ = O.pair_forward(p64, a1, j1.double(), a2, j2.double(), encoder="relgcn", n_layers=3, attn="nie")
    got = y[T(pick).to(y.device)].double().cpu()
    assert (got - yo).abs().max().item() <= 1e-4 * max(yo.abs().max().item(), 1.0)

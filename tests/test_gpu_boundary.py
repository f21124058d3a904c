"""The reference's own call forms on the GPU path, against the dense oracle:
dense (mb, N, hid) atom arrays into the co-attention modules (nie_coattention.py:335-341; every position counts, the
reference masks nothing), float atom features through the encoder (models/ggnn.py:600-605), the lazily sized MLP
(train_ddi_modify.py:136), dropout in eval mode, retained-graph double backward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


from parity_util import close as _close      # asserts AND logs the achieved relative error


@pytest.mark.parametrize("attn", ["nie", "pool", "parallel", "alternating", "global", "neural"])
def test_coattention_takes_dense_atom_arrays(attn):
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    d, o, mb, N1, N2 = 32, 16, 7, 19, 11
    head = 1 if attn == "parallel" else 4
    p = O.make_pair_params(777, hidden_dim=d, out_dim=o, n_layers=2, attn=attn, head=head, dtype=torch.float64, bias_scale=0.1)
    model = build_pair_predictor(hidden_dim=d, out_dim=o, n_layers=2, attn=attn, head=head).to(dev)
    load_param_dict(model, p)
    rs = np.random.RandomState(0)
    a1 = T(rs.normal(size=(mb, N1, d))).requires_grad_(); a2 = T(rs.normal(size=(mb, N2, d))).requires_grad_()
    g1 = T(rs.normal(size=(mb, o))).requires_grad_(); g2 = T(rs.normal(size=(mb, o))).requires_grad_()
    fn = {"nie": lambda: O.nie_coattention(p, a1, a2, "tanh", prefix="attn/"),
          "pool": lambda: O.pooling_coattention(p, a1, a2, "tanh", prefix="attn/"),
          "parallel": lambda: O.parallel_coattention(p, a1, g1, a2, g2, "tanh", prefix="attn/"),
          "alternating": lambda: O.alternating_coattention(p, a1, g1, a2, g2, prefix="attn/"),
          "global": lambda: O.global_coattention(p, a1, a2, prefix="attn/"),
          "neural": lambda: O.neural_coattention(p, a1, a2, "tanh", prefix="attn/")}[attn]
    c1o, c2o = fn()
    wv = T(rs.normal(size=(mb, o)))
    ((c1o * wv).sum() + (c2o * wv.flip(0)).sum()).backward()
    x1 = a1.detach().float().to(dev).requires_grad_(); x2 = a2.detach().float().to(dev).requires_grad_()
    h1 = g1.detach().float().to(dev).requires_grad_(); h2 = g2.detach().float().to(dev).requires_grad_()
    c1, c2 = model.attn(x1, h1, x2, h2)                       # exactly the reference's call, train_binary.py:96
    w = wv.float().to(dev)
    ((c1 * w).sum() + (c2 * w.flip(0)).sum()).backward()
    _close(c1, c1o, "compact_1"); _close(c2, c2o, "compact_2")
    _close(x1.grad, a1.grad, "d atoms_1"); _close(x2.grad, a2.grad, "d atoms_2")
    if g1.grad is not None:
        _close(h1.grad, g1.grad, "d g_1"); _close(h2.grad, g2.grad, "d g_2")


def test_float_atom_features_bypass_the_embedding():
    from bmp import synth
    from bmp.ggnn import GGNN
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(9, seed=3, n_lo=3, n_hi=20, n_mean=9)
    a, j = synth.concat_mols(store)
    mb, A = a.shape
    d = 64
    p = O.make_pair_params(777, hidden_dim=d, out_dim=32, n_layers=3, attn=None, dtype=torch.float64)
    sub = {k[len("graph_conv/"):]: v for k, v in p.items() if k.startswith("graph_conv/")}
    rs = np.random.RandomState(1)
    x = T(rs.normal(size=(mb, A, d)) * (a[:, :, None] != 0)).requires_grad_()        # zero features at padded positions
    g_o, at_o = O.ggnn_forward(sub, x, T(j).double(), 3, True, prefix="")
    wv = T(rs.normal(size=tuple(g_o.shape)))
    (g_o * wv).sum().backward()
    enc = GGNN(out_dim=32, hidden_dim=d, n_layers=3).to(dev)
    load_param_dict(enc, sub)
    xd = x.detach().float().to(dev).requires_grad_()
    for adj in (j, T(j).to(dev)):                                  # host and device adjacency
        xd.grad = None
        g = enc(xd, adj)                                           # models/ggnn.py:600-605: a float array in the atoms slot
        (g * wv.float().to(dev)).sum().backward()
        _close(g, g_o, "g"); _close(enc.get_atom_array().dense(), at_o, "atoms"); _close(xd.grad, x.grad, "d features")
    with pytest.raises(ValueError):
        enc(torch.zeros(mb, A, d + 8, device=dev), j)


def test_relgcn_float_input_type():
    """RelGCN(input_type='float') (models/relgcn.py:42-43,61-66): the embedding is a lazily sized GraphLinear on float atom
    features, applied to every position (padded ones get its bias)."""
    from bmp import synth
    from bmp.relgcn import RelGCN
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(8, seed=5, n_lo=3, n_hi=18, n_mean=8)
    a, j = synth.concat_mols(store)
    mb, A = a.shape
    f, chs = 12, [32, 64, 64]
    dr = O._Draw(5, torch.float64, 0.1)
    O.init_relgcn(dr, "", 16, chs)
    p = dict(dr.p)
    rs = np.random.RandomState(2)
    p["embed/W"] = T(rs.normal(size=(chs[0], f)) / np.sqrt(f)); p["embed/b"] = T(rs.normal(size=chs[0]) * 0.1)
    p = {k: v.requires_grad_() for k, v in p.items()}
    x = T(rs.normal(size=(mb, A, f)) * (a[:, :, None] != 0)).requires_grad_()
    # the oracle's relgcn_forward embeds ids; the float form replaces that line by the GraphLinear (models/relgcn.py:67)
    h = O.linear(x, p["embed/W"], p["embed/b"])
    adj = O.rescale_adj(T(j).double())
    for i in range(2):
        pre = f"rgcn_convs/{i}"
        h = torch.tanh(O.relgcn_update(h, adj, p[f"{pre}/graph_linear_self/W"], p[f"{pre}/graph_linear_self/b"],
                                       p[f"{pre}/graph_linear_edge/W"], p[f"{pre}/graph_linear_edge/b"]))
    g_o = O.ggnn_readout_block(p, "rgcn_readout", h, None, True, "tanh", "identity")
    wv = T(rs.normal(size=tuple(g_o.shape)))
    (g_o * wv).sum().backward()
    enc = RelGCN(out_channels=16, ch_list=chs, input_type='float', scale_adj=True).to(dev)
    xd = x.detach().float().to(dev).requires_grad_()
    g = enc(xd, j)                                   # materialises the embedding at this first call
    load_param_dict(enc, {k: v.detach() for k, v in p.items()})
    enc.zero_grad()
    g = enc(xd, j)
    (g * wv.float().to(dev)).sum().backward()
    _close(g, g_o, "g"); _close(xd.grad, x.grad, "d features")
    _close(enc.embed.W.grad, p["embed/W"].grad, "d embed W"); _close(enc.embed.b.grad, p["embed/b"].grad, "d embed b")


def test_reference_construction_forms_train_on_the_gpu():
    """set_up_predictor of train_ddi_modify.py:134-150, verbatim argument forms, then one optimizer step."""
    from bmp import synth
    from bmp.dp import FlatAdam
    from bmp.ggnn import GGNN
    from bmp.mlp import MLP
    from bmp.predictor import GraphConvPredictorForPair
    dev = torch.device("cuda:0")
    mlp = MLP(out_dim=1, hidden_dims=(32, 16))
    ggnn = GGNN(out_dim=16, hidden_dim=64, n_layers=2, concat_hidden=False, dropout_rate=0.0)
    predictor = GraphConvPredictorForPair(ggnn, mlp).to(dev)
    store = synth.make_store(12, seed=7, n_lo=3, n_hi=20, n_mean=9)
    a1, j1 = synth.concat_mols(store[:6]); a2, j2 = synth.concat_mols(store[6:])
    t = (torch.arange(6, device=dev) % 2).int().view(-1, 1)
    y = predictor(a1, j1, a2, j2)                                 # eager call, exactly the reference's (train_ddi_modify.py:66)
    predictor.loss(y, t).backward()
    eager = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1) for q in predictor.parameters()]).clone()
    opt = FlatAdam(predictor, alpha=1e-3)
    losses = []
    for k in range(6):
        y = opt.functional_forward(a1, j1, a2, j2)
        loss = predictor.loss(y, t)
        loss.backward()
        opt.collect_grads()
        if k == 0:
            assert (opt.grad - eager).abs().max().item() <= 1e-5 * eager.abs().max().item()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    prob = predictor.predict(a1, j1, a2, j2)
    assert prob.shape == (6, 1) and ((prob > 0) & (prob < 1)).all()
    h, (g1, g2) = predictor.predict_eval(a1, j1, a2, j2)                # eval_coattention.py:103-124: logits + the two vectors
    assert torch.allclose(torch.sigmoid(h), prob) and g1.shape == (6, 16) and g2.shape == (6, 16) and not h.requires_grad


def test_dropout_is_identity_in_eval_mode():
    from bmp import synth
    from bmp.ggnn import GGNN
    dev = torch.device("cuda:0")
    store = synth.make_store(6, seed=9, n_lo=3, n_hi=12, n_mean=7)
    a, j = synth.concat_mols(store)
    torch.manual_seed(0)
    enc0 = GGNN(out_dim=16, hidden_dim=64, n_layers=2).to(dev)
    enc1 = GGNN(out_dim=16, hidden_dim=64, n_layers=2, dropout_rate=0.3).to(dev)
    enc1.load_state_dict(enc0.state_dict())
    enc1.eval()
    with torch.no_grad():
        assert torch.equal(enc0(a, j), enc1(a, j))


@pytest.mark.parametrize("d,n_layers", [(64, 3), (128, 4)])
def test_dropout_in_training_keeps_the_gru_state_undropped(d, n_layers):
    """models/ggnn.py:626-627: F.dropout acts on the step output; the stateful GRU (models/ggnn.py:132) keeps its own
    un-dropped state.  With the SAME masks on both sides (the padded positions of a molecule share a mask: they are one
    packed row) the separate-state GRU path equals the oracle: readout, atoms, every parameter gradient."""
    from bmp import packed, synth
    from bmp.ggnn import GGNN
    from bmp.snapshot import grad_dict, load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(10, seed=21, n_lo=2, n_hi=24, n_mean=9)
    a, j = synth.concat_mols(store)
    mb, A = a.shape
    p_drop = 0.25
    p = O.make_pair_params(777, hidden_dim=d, out_dim=32, n_layers=n_layers, attn=None, dtype=torch.float64, bias_scale=0.05)
    sub = {k[len("graph_conv/"):]: v.requires_grad_() for k, v in p.items() if k.startswith("graph_conv/")}
    pb = packed.pack_from_dense([a], [j], device=dev)
    rs = np.random.RandomState(3)
    masks_rows = [T((rs.uniform(size=(pb.n_rows, d)) >= p_drop).astype(np.float64) / (1.0 - p_drop)) for _ in range(n_layers)]
    dm = pb.dense_maps[0].cpu()
    masks_dense = [mk[dm] for mk in masks_rows]                      # (mb, A, d): what F.dropout would have drawn
    g_o, at_o = O.ggnn_forward(sub, T(a), T(j).double(), n_layers, True, prefix="", dropout_masks=masks_dense)
    wv = T(rs.normal(size=tuple(g_o.shape)))
    (g_o * wv).sum().backward()
    enc = GGNN(out_dim=32, hidden_dim=d, n_layers=n_layers, dropout_rate=p_drop).to(dev)
    load_param_dict(enc, {k: v.detach() for k, v in sub.items()})
    enc.train()
    enc._dropout_masks = [mk.float().to(dev) for mk in masks_rows]
    g = enc(pb)
    (g * wv.float().to(dev)).sum().backward()
    _close(g, g_o, "g"); _close(enc.get_atom_array().dense(), at_o, "atoms")
    for name, gr in grad_dict(enc).items():
        _close(gr, sub[name].grad, f"grad {name}")
    # and with its own random masks it runs and differs from the eval-mode output
    enc._dropout_masks = None
    g_rand = enc(pb)
    enc.eval()
    with torch.no_grad():
        g_eval = enc(pb)
    assert torch.isfinite(g_rand).all() and not torch.allclose(g_rand, g_eval)


def test_second_backward_over_a_retained_graph_gives_the_same_weight_gradients():
    """tied steps accumulate their weight gradients inside the kernels, driven by host-side counters; a second
    backward over the same graph must start a fresh accumulation, not add to the first one's sums."""
    from bmp import packed, synth
    from bmp.ggnn import GGNN
    dev = torch.device("cuda:0")
    store = synth.make_store(10, seed=5, n_lo=3, n_hi=25, n_mean=10)
    ms = packed.MolStore(store)
    pb = packed.pack_from_store(ms, [np.arange(10)], device=dev)
    torch.manual_seed(0)
    enc = GGNN(out_dim=16, hidden_dim=64, n_layers=3).to(dev)
    g = enc(pb)
    loss = (g * g).sum()
    params = list(enc.parameters())
    first = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
    second = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
    for a, b, (name, _) in zip(first, second, enc.named_parameters()):
        if a is not None:
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), name

"""GPU parity tests of the drug-pair path: co-attention kernels and the whole pair predictor
(encoder x2 -> co-attention -> MLP -> loss), forward and backward, against the float64 oracle.
Tolerance 1e-4 (max-abs error relative to the max-abs reference, per tensor)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O            # noqa: E402
from bmp import synth, packed              # noqa: E402
import packed_ref as PR                    # noqa: E402
from test_gpu_ops import close, dev, to_dev, T       # noqa: E402


@pytest.fixture(scope="module")
def pairs():
    store = synth.make_store(48, seed=5, n_lo=2, n_hi=40, n_mean=12)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(11)
    i1, i2 = rs.randint(0, 48, 19), rs.randint(0, 48, 19)
    pb = packed.pack_from_store(ms, [i1, i2], device="cpu", with_dense_map=True)
    return store, i1, i2, pb


@pytest.mark.parametrize("d,o,act", [(16, 16, "tanh"), (128, 128, "tanh"), (24, 12, "identity")])
def test_nie_coattention_op(pairs, d, o, act):
    """The co-attention operator alone, on random atom states, vs the packed float64 restatement."""
    from bmp.coattention import NieFineCoattention
    from bmp.ggnn import PackedAtoms
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    pbd = to_dev(pb)
    B = len(i1)
    dr = O._Draw(d, torch.float64, 0.2)
    O.init_nie(dr, "", d, o, 8)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    g = torch.Generator().manual_seed(d)
    X = (torch.randn(pb.n_rows, d, generator=g, dtype=torch.float64) * (0.3 if act == "identity" else 1.0))
    Xr = X.clone().requires_grad_()
    c1, c2 = PR.nie_coattention(p, pb, Xr, np.arange(B), B + np.arange(B), activation=act)
    w1 = torch.randn(B, o, generator=g, dtype=torch.float64); w2 = torch.randn(B, o, generator=g, dtype=torch.float64)
    ((c1 * w1).sum() + (c2 * w2).sum()).backward()

    att = NieFineCoattention(d, o, 8, activation=act).to(dev())
    load_param_dict(att, p)
    Xd = X.float().to(dev()).requires_grad_()
    at = PackedAtoms(Xd, pbd)
    o1, o2 = att(at, None, at, None)
    close(o1, c1, "compact_1"); close(o2, c2, "compact_2")
    ((o1 * w1.float().to(dev())).sum() + (o2 * w2.float().to(dev())).sum()).backward()
    # rows that belong to no molecule get no gradient in either implementation
    close(Xd.grad, Xr.grad, "dX")
    for name, gr in grad_dict(att).items():
        # with the identity activation the softmaxes are shift invariant, so d/d(energy bias) is
        # analytically 0: compare it on the scale of the other energy-layer gradients
        floor = p["energy_layer/V1"].grad.abs().max().item() if name == "energy_layer/b" else 1e-6
        close(gr, p[name].grad, f"grad {name}", floor=floor)


@pytest.mark.parametrize("d,o", [(16, 16), (64, 32)])
def test_fourier_nie_coattention_op(pairs, d, o):
    """FourierFineCoattention: the product folds the DFT into the energy operands; the restatement transforms the
    atom states and calls the bilinear form twice (nie_coattention.py:460-505)."""
    from bmp.coattention import FourierFineCoattention
    from bmp.ggnn import PackedAtoms
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    pbd = to_dev(pb)
    B = len(i1)
    dr = O._Draw(d + 5, torch.float64, 0.2)
    O.init_nie(dr, "", d, o, 8)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    g = torch.Generator().manual_seed(d)
    X = torch.randn(pb.n_rows, d, generator=g, dtype=torch.float64) * (0.5 / d ** 0.5)    # |DFT| ~ sqrt(d) |x|
    Xr = X.clone().requires_grad_()
    c1, c2 = PR.nie_coattention(p, pb, Xr, np.arange(B), B + np.arange(B), activation="tanh", fourier=True)
    w1 = torch.randn(B, o, generator=g, dtype=torch.float64); w2 = torch.randn(B, o, generator=g, dtype=torch.float64)
    ((c1 * w1).sum() + (c2 * w2).sum()).backward()
    att = FourierFineCoattention(d, o, 8, activation="tanh").to(dev())
    load_param_dict(att, p)
    Xd = X.float().to(dev()).requires_grad_()
    at = PackedAtoms(Xd, pbd)
    o1, o2 = att(at, None, at, None)
    close(o1, c1, "compact_1"); close(o2, c2, "compact_2")
    ((o1 * w1.float().to(dev())).sum() + (o2 * w2.float().to(dev())).sum()).backward()
    close(Xd.grad, Xr.grad, "dX")
    for name, gr in grad_dict(att).items():
        close(gr, p[name].grad, f"grad {name}")


@pytest.mark.parametrize("n_lt,d,o", [(1, 16, 16), (2, 64, 32), (3, 128, 128)])
def test_deep_nie_coattention_op(pairs, n_lt, d, o):
    """Deep / VeryDeep / ExtremeDeep NieFineCoattention: the product folds each side's affine chain into the
    projection operands; the restatement applies the layers one by one (nie_coattention.py:54-59, :155-163)."""
    from bmp import coattention as C
    from bmp.ggnn import PackedAtoms
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    pbd = to_dev(pb)
    B = len(i1)
    dr = O._Draw(d + n_lt, torch.float64, 0.2)
    O.init_nie(dr, "", d, o, 8, n_lt=n_lt)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    g = torch.Generator().manual_seed(d)
    X = torch.randn(pb.n_rows, d, generator=g, dtype=torch.float64)
    Xr = X.clone().requires_grad_()
    c1, c2 = PR.nie_coattention(p, pb, Xr, np.arange(B), B + np.arange(B), activation="tanh", n_lt=n_lt)
    w1 = torch.randn(B, o, generator=g, dtype=torch.float64); w2 = torch.randn(B, o, generator=g, dtype=torch.float64)
    ((c1 * w1).sum() + (c2 * w2).sum()).backward()
    cls = {1: C.DeepNieFineCoattention, 2: C.VeryDeepNieFineCoattention, 3: C.ExtremeDeepNieFineCoattention}[n_lt]
    att = cls(d, o, 8, activation="tanh").to(dev())
    load_param_dict(att, p)
    Xd = X.float().to(dev()).requires_grad_()
    at = PackedAtoms(Xd, pbd)
    o1, o2 = att(at, None, at, None)
    close(o1, c1, "compact_1"); close(o2, c2, "compact_2")
    ((o1 * w1.float().to(dev())).sum() + (o2 * w2.float().to(dev())).sum()).backward()
    close(Xd.grad, Xr.grad, "dX")
    gd = grad_dict(att)
    assert set(gd) == set(p)
    for name, gr in gd.items():
        close(gr, p[name].grad, f"grad {name}")


@pytest.mark.parametrize("d,nl,tying", [(16, 2, True), (128, 4, True), (32, 3, False)])
def test_pair_predictor_matches_dense_oracle(pairs, d, nl, tying):
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    pbd = to_dev(pb)
    B = len(i1)
    p = O.make_pair_params(777, hidden_dim=d, out_dim=d, n_layers=nl, weight_tying=tying, attn="nie",
                           dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    label = torch.from_numpy((np.random.RandomState(1).uniform(size=(B, 1)) < 0.4).astype(np.int32))
    label[3, 0] = -1                                   # ignored label (sigmoid_cross_entropy)
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=nl, weight_tying=tying)
    loss = O.sigmoid_cross_entropy(y, label)
    loss.backward()

    model = build_pair_predictor(hidden_dim=d, out_dim=d, n_layers=nl, weight_tying=tying, attn="nie").to(dev())
    load_param_dict(model, p)
    yd = model(pbd)
    close(yd, y, "logits")
    close(model.g1, g1, "g1"); close(model.g2, g2, "g2")
    ld = model.loss(yd, label.to(dev()))
    close(ld, loss, "loss")
    ld.backward()
    grads = grad_dict(model)
    for name, gr in grads.items():
        ref = p[name].grad
        if ref is None:                               # readout params: the fine family ignores g_1/g_2
            assert name.startswith(("graph_conv/i_layers", "graph_conv/j_layers")), name
            continue
        close(gr, ref, f"grad {name}")
    # predict(): sigmoid of the logits under no-grad (train_binary.py:120-127)
    close(model.predict(pbd), torch.sigmoid(y), "predict")


def test_pair_golden_dense_four_array_form(golden_dir):
    """The reference's call form: four dense arrays (atoms_1, adjs_1, atoms_2, adjs_2)."""
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    z = np.load(os.path.join(golden_dir, "pair_nie_small.npz"))
    p = {k[6:]: z[k] for k in z.files if k.startswith("param:")}
    model = build_pair_predictor(hidden_dim=8, out_dim=8, n_layers=2, attn="nie").to(dev())
    load_param_dict(model, p)
    y = model(T(z["atoms_1"]), T(z["adj_1"]), T(z["atoms_2"]), T(z["adj_2"]))
    close(y, T(z["y"]), "logits")
    close(model.g1, T(z["g1"]), "g1"); close(model.g2, T(z["g2"]), "g2")
    loss = model.loss(y, T(z["label"]).to(dev()))
    close(loss, T(z["loss"]), "loss")
    loss.backward()
    for name, gr in grad_dict(model).items():
        close(gr, T(z["grad:" + name]), f"grad {name}")


def test_pair_no_attention_form(pairs):
    """train_ddi_modify.py:66-77: encoder x2 -> concat -> MLP, no co-attention."""
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    pbd = to_dev(pb)
    p = O.make_pair_params(5, hidden_dim=16, out_dim=16, n_layers=2, attn=None, dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    y, _, _ = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn=None)
    y.sum().backward()
    model = build_pair_predictor(hidden_dim=16, out_dim=16, n_layers=2, attn=None).to(dev())
    load_param_dict(model, p)
    yd = model(pbd)
    close(yd, y, "logits")
    yd.sum().backward()
    for name, gr in grad_dict(model).items():
        close(gr, p[name].grad, f"grad {name}")


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize("d,act", [(16, "tanh"), (64, "tanh")])
def test_pooling_coattention_pair(pairs, d, act):
    """PoolingFineCoattention (PoolingFineCoattention.py:31-57) through the pair predictor vs the dense oracle."""
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    pbd = to_dev(pb)
    p = O.make_pair_params(31, hidden_dim=d, out_dim=d, n_layers=2, attn="pool", dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn="pool")
    c = torch.randn(y.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(17))
    (y * c).sum().backward()
    model = build_pair_predictor(hidden_dim=d, out_dim=d, n_layers=2, attn="pool").to(dev())
    load_param_dict(model, p)
    yd = model(pbd)
    close(yd, y, "logits"); close(model.g1, g1, "g1"); close(model.g2, g2, "g2")
    (yd * c.float().to(dev())).sum().backward()
    for name, gr in grad_dict(model).items():
        ref = p[name].grad
        if ref is None:
            assert name.startswith(("graph_conv/i_layers", "graph_conv/j_layers")), name
            continue
        # the atom weights are a softmax of means of C: a common shift of C nearly cancels, so the energy bias gradient
        # is a small difference of large sums -- compare it on the scale of the energy layer's other gradients
        floor = p["attn/energy_layer/V1"].grad.abs().max().item() if name == "attn/energy_layer/b" else 1e-6
        close(gr, ref, f"grad {name}", floor=floor)


@pytest.mark.parametrize("attn", ["parallel", "circ", "alternating", "global", "neural", "bimpm"])
@pytest.mark.parametrize("joint", [True, False])
def test_coarse_coattention_pair(pairs, attn, joint):
    """Coarse (atom x molecule-vector) co-attention family through the pair predictor vs the dense oracle,
    on one two-sided packed batch (joint) and in the reference's four-dense-array call form."""
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    d = 16
    # BiMPM is built with head = fp_out_dim (train_binary.py:253-256), the others with the builder's head = 8
    p = O.make_pair_params(41, hidden_dim=d, out_dim=d, n_layers=2, attn=attn, dtype=torch.float64, head=(d if attn == "bimpm" else 8))
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn=attn)
    c = torch.randn(y.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(17))
    (y * c).sum().backward()
    model = build_pair_predictor(hidden_dim=d, out_dim=d, n_layers=2, attn=attn).to(dev())
    load_param_dict(model, p)
    yd = model(to_dev(pb)) if joint else model(T(a1), T(j1), T(a2), T(j2))
    close(model.g1, g1, "g1"); close(model.g2, g2, "g2"); close(yd, y, "logits")
    (yd * c.float().to(dev())).sum().backward()
    for name, gr in grad_dict(model).items():
        ref = p[name].grad
        if ref is None:
            assert float(gr.abs().max()) == 0.0, name
            continue
        # a bias added right before a softmax has an analytically zero gradient (shift invariance):
        # compare it on the scale of the weight gradient of the same layer
        floor = p["attn/energy_layers_2/0/W"].grad.abs().max().item() if name == "attn/energy_layers_2/0/b" else 1e-6
        if attn == "circ":       # gate and value both come from j_layer: gradients are sums of o-term products
            floor = 1e-4 * float(ref.abs().max())
        # BiMPM with 16 perspectives on these molecules has near-ties in its max selections: the ORACLE ITSELF evaluated in
        # float32 differs from its float64 gradients by up to 1.3e-3 here (W of the GRU; 7e-4 on embed.W), while its logits
        # agree to 2e-6 -- an argmax flips, the gradient takes the other branch.  The HIP path lands at 2.4e-4.
        close(gr, ref, f"grad {name}", floor=floor, **(dict(tol=2e-3) if attn == "bimpm" else {}))


@pytest.mark.parametrize("attn", ["deep", "extreme-deep", "fourier"])
def test_folded_fine_variants_through_the_pair_predictor(pairs, attn):
    """Deep* / Fourier co-attention inside the whole pair model (encoder gradients included) vs the dense oracle."""
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    d = 16
    p = O.make_pair_params(43, hidden_dim=d, out_dim=d, n_layers=2, attn=attn, dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn=attn)
    c = torch.randn(y.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(17))
    (y * c).sum().backward()
    model = build_pair_predictor(hidden_dim=d, out_dim=d, n_layers=2, attn=attn).to(dev())
    load_param_dict(model, p)
    yd = model(to_dev(pb))
    close(model.g1, g1, "g1"); close(model.g2, g2, "g2"); close(yd, y, "logits")
    (yd * c.float().to(dev())).sum().backward()
    for name, gr in grad_dict(model).items():
        ref = p[name].grad
        if ref is None:
            assert gr is None or float(gr.abs().max()) == 0.0, name
            continue
        close(gr, ref, f"grad {name}")


@pytest.mark.parametrize("attn", ["deep", "extreme-deep", "fourier", "circ", "pool", "parallel", "alternating", "global", "neural", "bimpm"])
def test_pair_golden_other_coattention(golden_dir, attn):
    """Committed float64 vectors of every co-attention family (tests/golden/make_golden.py), reference call form
    (four dense arrays)."""
    import os
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    z = np.load(os.path.join(golden_dir, f"pair_attn_{attn.replace('-', '_')}.npz"))
    p = {k[6:]: z[k] for k in z.files if k.startswith("param:")}
    model = build_pair_predictor(hidden_dim=8, out_dim=8, n_layers=2, attn=attn, head=8 if attn != "parallel" else 1).to(dev())
    load_param_dict(model, p)
    y = model(T(z["atoms_1"]), T(z["adj_1"]), T(z["atoms_2"]), T(z["adj_2"]))
    close(y, T(z["y"]), "logits")
    loss = model.loss(y, T(z["label"]).to(dev()))
    close(loss, T(z["loss"]), "loss")
    loss.backward()
    for name, gr in grad_dict(model).items():
        ref = T(z["grad:" + name])
        if gr is None:
            assert float(ref.abs().max()) == 0.0, name
            continue
        close(gr, ref, f"grad {name}", floor=1e-4 * float(ref.abs().max()) + 1e-7)

"""BiMPM (models/coattention/bimpm.py:45-199) on the HIP path against the dense float64 oracle restatement: module call
on PackedAtoms (two-sided batch and two one-sided batches), on the reference's dense (mb, N, hid) arrays, composed in the
pair predictor; values, input gradients and the three perspective matrices' gradients.  Parity unpinned."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


from parity_util import close as _close      # asserts AND logs the achieved relative error


@pytest.mark.parametrize("d,H,mb,N1,N2", [(32, 8, 5, 9, 13), (64, 16, 3, 33, 20), (128, 128, 2, 12, 7)])
def test_bimpm_dense_arrays_match_oracle(d, H, mb, N1, N2):
    from bmp.bimpm import BiMPM
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(d + H)
    p = {f"attn/{n}": T(rs.normal(0, np.sqrt(2.0 / d), size=(H, d))).requires_grad_() for n in ("max_pooling_W", "att_mean_W", "att_max_W")}
    a1 = T(rs.normal(size=(mb, N1, d))).requires_grad_(); a2 = T(rs.normal(size=(mb, N2, d))).requires_grad_()
    m1o, m2o = O.bimpm_coattention(p, a1, a2, prefix="attn/")
    wv1, wv2 = T(rs.normal(size=tuple(m1o.shape))), T(rs.normal(size=tuple(m2o.shape)))
    ((m1o * wv1).sum() + (m2o * wv2).sum()).backward()
    mod = BiMPM(hidden_dim=d, out_dim=H, head=H).to(dev)
    with torch.no_grad():
        for n in ("max_pooling_W", "att_mean_W", "att_max_W"):
            getattr(mod, n).copy_(p[f"attn/{n}"].float())
    x1 = a1.detach().float().to(dev).requires_grad_(); x2 = a2.detach().float().to(dev).requires_grad_()
    m1, m2 = mod(x1, None, x2, None)                                   # the reference's call form (train_binary.py:96)
    assert m1.shape == (mb, 3 * H)
    ((m1 * wv1.float().to(dev)).sum() + (m2 * wv2.float().to(dev)).sum()).backward()
    _close(m1, m1o, "mol_1"); _close(m2, m2o, "mol_2")
    _close(x1.grad, a1.grad, "d atoms_1"); _close(x2.grad, a2.grad, "d atoms_2")
    for n in ("max_pooling_W", "att_mean_W", "att_max_W"):
        _close(getattr(mod, n).grad, p[f"attn/{n}"].grad, f"d {n}")


@pytest.mark.parametrize("form", ["two-sided", "four arrays"])
def test_bimpm_pair_predictor_matches_oracle(form):
    """GGNN encoder + BiMPM + lazily sized MLP as train_binary.py:253-277 composes them; padded positions included
    (the virtual pad row carries their multiplicity in the sums and takes part in the maxima once)."""
    from bmp import packed, synth
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import grad_dict, load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(30, seed=17, n_lo=2, n_hi=30, n_mean=11)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(4)
    B, d, H = 9, 64, 16
    i1, i2 = rs.randint(0, 30, B), rs.randint(0, 30, B)
    lab = rs.randint(0, 2, (B, 1)).astype(np.int32)
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    p = O.make_pair_params(777, hidden_dim=d, out_dim=H, n_layers=2, attn="bimpm", head=H, dtype=torch.float64, bias_scale=0.05)
    p = {k: v.requires_grad_() for k, v in p.items()}
    y_o, _, _ = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn="bimpm")
    loss_o = O.sigmoid_cross_entropy(y_o, T(lab))
    loss_o.backward()
    model = build_pair_predictor(hidden_dim=d, out_dim=H, n_layers=2, attn="bimpm").to(dev)      # builder default head = 8 ...
    assert model.attn.head == H                                       # ... but BiMPM(head=fp_out_dim), train_binary.py:253-256
    assert model.mlp.layers[0].W.shape == (32, 2 * 3 * H)
    load_param_dict(model, p)
    if form == "two-sided":
        y = model(packed.pack_from_store(ms, [i1, i2], device=dev))
    else:
        y = model(a1, j1, a2, j2)
    loss = model.loss(y, T(lab).to(dev))
    loss.backward()
    _close(y, y_o, "logits"); _close(loss, loss_o, "loss")
    for name, gr in grad_dict(model).items():
        if p[name].grad is not None:
            _close(gr, p[name].grad, f"grad {name}", tol=2e-4)

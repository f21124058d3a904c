"""The other BASELINE.json configurations as parity cases.

C1 (CPU, no GPU): 256 drug pairs, GGNN 2-step d=16 + MLP, batch 32, the oracle's own fwd+bwd+Adam loop -- the
"repo CPU path" plumbing the CPU baseline of bench.py times.
C4 (GPU): 37-class multi-label DDI, GGNN d=256, no co-attention (train_ggnn_hole_multi_class_x37.py:71-91),
multi-hot labels -- the HIP path against the dense oracle, logits, loss and every gradient.
"""
import numpy as np
import pytest
import torch

from bmp import synth
from oracle import ref_cpu as O

T = torch.from_numpy


def test_c1_cpu_path_trains():
    store = synth.make_store()
    i1, i2, lab = synth.make_pairs()
    p = O.make_pair_params(777, hidden_dim=16, out_dim=16, n_layers=2, attn=None, dtype=torch.float32, bias_scale=0.0)
    names = sorted(p)
    params = [p[n].requires_grad_() for n in names]
    state = [dict(m=torch.zeros_like(x), v=torch.zeros_like(x)) for x in params]
    losses = []
    for epoch in range(3):
        tot = 0.0
        for step in range(8):                                     # 8 x 32 = the first 256 pairs of the permutation
            sl = slice(step * 32, (step + 1) * 32)
            a1, j1 = synth.concat_mols([store[k] for k in i1[sl]]); a2, j2 = synth.concat_mols([store[k] for k in i2[sl]])
            y, _, _ = O.pair_forward(p, T(a1), T(j1), T(a2), T(j2), n_layers=2, attn=None)
            loss = O.sigmoid_cross_entropy(y, T(lab[sl].reshape(-1, 1)))
            grads = torch.autograd.grad(loss, params, allow_unused=True)
            grads = [g if g is not None else torch.zeros_like(x) for g, x in zip(grads, params)]
            with torch.no_grad():
                O.chainer_adam_step(params, grads, state, epoch * 8 + step + 1, alpha=1e-2)
            tot += float(loss)
        losses.append(tot / 8)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.gpu
def test_c4_multilabel_d256_matches_oracle():
    from bmp import packed
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import grad_dict, load_param_dict
    dev = torch.device("cuda:0")
    store = synth.make_store(60, seed=4, n_lo=4, n_hi=40, n_mean=16)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(2)
    B, C = 12, 37
    i1, i2 = rs.randint(0, 60, B), rs.randint(0, 60, B)
    lab = (rs.uniform(size=(B, C)) < 0.08).astype(np.int32)
    lab[1, 5] = -1; lab[7, 0] = -1                               # ignored entries
    p = O.make_pair_params(777, hidden_dim=256, out_dim=256, n_layers=4, attn=None, class_num=C, dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    y, _, _ = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=4, attn=None)
    loss = O.sigmoid_cross_entropy(y, T(lab))
    loss.backward()

    model = build_pair_predictor(hidden_dim=256, out_dim=256, n_layers=4, attn=None, class_num=C).to(dev)
    load_param_dict(model, p)
    pb = packed.pack_from_store(ms, [i1, i2], device=dev)
    yd = model(pb)
    ld = model.loss(yd, T(lab).to(dev))
    ld.backward()

    from parity_util import close
    close(yd, y, "logits"); close(ld, loss, "loss")
    for name, gr in grad_dict(model).items():
        close(gr, p[name].grad, f"grad {name}")

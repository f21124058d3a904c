"""Link-predictor tail on the device (bmp_mlp_* / bmp_sce_*) against the plain fp32 torch ops of the same
functions (models/mlp.py:20-45; chainer sigmoid_cross_entropy, train_ddi_modify.py:285)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_mlp(mlp, x):
    h = x
    for l in mlp.layers:
        h = torch.relu(torch.nn.functional.linear(h, l.W, l.b))
    return torch.nn.functional.linear(h, mlp.l_out.W, mlp.l_out.b)


def _ref_sce(y, t):
    mask = t != -1
    loss = torch.nn.functional.softplus(y) - t.to(y.dtype) * y
    return torch.where(mask, loss, torch.zeros_like(loss)).sum() / mask.sum().clamp(min=1)


@pytest.mark.parametrize("B,o,hidden,C", [(1024, 128, (32, 16), 1), (37, 20, (32, 16), 37), (5, 8, (), 3), (130, 64, (64, 64, 16), 2)])
def test_mlp_and_loss_match_torch(B, o, hidden, C):
    from bmp.mlp import MLP, sigmoid_cross_entropy
    dev = torch.device("cuda:0")
    torch.manual_seed(B + C)
    mlp = MLP(C, hidden, in_dim=2 * o).to(dev)
    with torch.no_grad():
        for p in mlp.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    g1 = torch.randn(B, o, device=dev, requires_grad=True)
    g2 = torch.randn(B, o, device=dev, requires_grad=True)
    t = torch.randint(-1, 2, (B, C), device=dev, dtype=torch.int32)
    y = mlp(g1, g2)
    loss = sigmoid_cross_entropy(y, t)
    loss.backward()
    got = [y.detach().clone(), loss.detach().clone(), g1.grad.clone(), g2.grad.clone()] + [p.grad.clone() for p in mlp.parameters()]
    for p in list(mlp.parameters()) + [g1, g2]:
        p.grad = None
    yr = _ref_mlp(mlp, torch.cat((g1, g2), dim=1))
    lr = _ref_sce(yr, t)
    lr.backward()
    want = [yr.detach(), lr.detach(), g1.grad, g2.grad] + [p.grad for p in mlp.parameters()]
    for a, b in zip(got, want):
        scale = max(b.abs().max().item(), 1e-6)
        assert (a - b).abs().max().item() <= 1e-5 * scale + 1e-7, ((a - b).abs().max().item(), scale)


def test_loss_ignores_minus_one_and_empty_mask():
    from bmp.mlp import sigmoid_cross_entropy
    dev = torch.device("cuda:0")
    y = torch.tensor([[0.5], [-2.0], [3.0]], device=dev, requires_grad=True)
    t = torch.tensor([[1], [-1], [0]], device=dev, dtype=torch.int32)
    loss = sigmoid_cross_entropy(y, t)
    loss.backward()
    ref = (torch.nn.functional.softplus(torch.tensor(0.5)) - 0.5 + torch.nn.functional.softplus(torch.tensor(3.0))) / 2
    assert abs(loss.item() - ref.item()) < 1e-6
    assert y.grad[1].item() == 0.0
    y2 = torch.zeros(2, 1, device=dev, requires_grad=True)
    l2 = sigmoid_cross_entropy(y2, torch.full((2, 1), -1, device=dev, dtype=torch.int32))
    l2.backward()
    assert l2.item() == 0.0 and float(y2.grad.abs().sum()) == 0.0


@pytest.mark.parametrize("B,o,hidden,C,gscale", [(1024, 128, (32, 16), 1, 1.0), (37, 20, (32, 16), 37, 1.0), (5, 8, (), 3, -2.5),
                                                  (130, 64, (64, 64, 16), 2, 0.125), (1024, 256, (32, 16), 37, 1.0)])
def test_link_predictor_and_loss_in_one_launch_equal_the_separate_launches(B, o, hidden, C, gscale):
    """MLP.forward_loss (bmp_mlp_sce_fwdbwd + bmp_mlp_bwd_w: the reference's Classifier around the link predictor as one launch
    each way) against MLP.forward + sigmoid_cross_entropy: the same logits, input and parameter gradients bit for bit (the same
    sums in the same order; the scale arrives as a power of two or one), the loss to rounding (its numerators meet in another
    order); and both against the plain torch ops."""
    from bmp.mlp import MLP, sigmoid_cross_entropy
    dev = torch.device("cuda:0")
    torch.manual_seed(B + C)
    mlp = MLP(C, hidden, in_dim=2 * o).to(dev)
    with torch.no_grad():
        for p in mlp.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    g1 = torch.randn(B, o, device=dev, requires_grad=True)
    g2 = torch.randn(B, o, device=dev, requires_grad=True)
    t = torch.randint(-1, 2, (B, C), device=dev, dtype=torch.int32)
    res = []
    for fused in (True, False):
        for p in list(mlp.parameters()) + [g1, g2]:
            p.grad = None
        if fused:
            loss, y = mlp.forward_loss(g1, g2, t)
            assert not y.requires_grad
        else:
            y = mlp(g1, g2)
            loss = sigmoid_cross_entropy(y, t)
        (loss * gscale).backward()
        res.append([y.detach().clone(), loss.detach().clone(), g1.grad.clone(), g2.grad.clone()] + [p.grad.clone() for p in mlp.parameters()])
    exact = gscale in (1.0, 0.125)
    for k, (a, b) in enumerate(zip(*res)):
        scale = max(b.abs().max().item(), 1e-6)
        if k == 1 or not exact:
            assert (a - b).abs().max().item() <= 2e-6 * scale, (k, (a - b).abs().max().item(), scale)
        else:
            assert torch.equal(a, b), (k, (a - b).abs().max().item())
    for p in list(mlp.parameters()) + [g1, g2]:
        p.grad = None
    yr = _ref_mlp(mlp, torch.cat((g1, g2), dim=1))
    lr = _ref_sce(yr, t)
    (lr * gscale).backward()
    want = [yr.detach(), lr.detach(), g1.grad, g2.grad] + [p.grad for p in mlp.parameters()]
    for a, b in zip(res[0], want):
        scale = max(b.abs().max().item(), 1e-6)
        assert (a - b).abs().max().item() <= 1e-5 * scale + 1e-7, ((a - b).abs().max().item(), scale)
    # a second pass right behind the first: the launch leaves its ticket word at zero
    loss2, y2 = mlp.forward_loss(g1, g2, t)
    assert torch.equal(loss2, res[0][1]) and torch.equal(y2, res[0][0])


def test_loss_of_a_batch_without_counted_labels_is_zero():
    from bmp.mlp import MLP
    dev = torch.device("cuda:0")
    mlp = MLP(2, (32, 16), in_dim=16).to(dev)
    g1 = torch.randn(9, 8, device=dev, requires_grad=True)
    g2 = torch.randn(9, 8, device=dev, requires_grad=True)
    loss, _y = mlp.forward_loss(g1, g2, torch.full((9, 2), -1, device=dev, dtype=torch.int32))
    loss.backward()
    assert loss.item() == 0.0 and float(g1.grad.abs().sum()) == 0.0 and all(float(p.grad.abs().sum()) == 0.0 for p in mlp.parameters())


def test_classifier_form_of_the_pair_predictor_equals_forward_plus_loss():
    """GraphConvPredictorForPair.forward_loss / FlatAdam.functional_loss (the reference's Classifier call, train_ddi_modify.py:
    284-286) against functional_forward + model.loss on the same batch: the flat gradient bit for bit, the loss to rounding;
    with and without a co-attention, single- and multi-label."""
    import numpy as np
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(40, seed=3, n_lo=4, n_hi=40, n_mean=16)
    ds = packed.DeviceMolStore(packed.MolStore(store), dev)
    rs = np.random.RandomState(1)
    i1, i2 = rs.randint(0, 40, 96), rs.randint(0, 40, 96)
    for attn, C in (("nie", 1), (None, 5)):
        lab = rs.randint(-1, 2, (96, C)).astype(np.int32)
        pb, t = packed.pack_from_store_device(ds, [i1, i2], labels=lab)
        torch.manual_seed(0)
        model = build_pair_predictor(hidden_dim=64, out_dim=64, n_layers=2, attn=attn, head=4, class_num=C).to(dev)
        opt = FlatAdam(model, alpha=1e-3)
        y = opt.functional_forward(pb)
        l0 = model.loss(y, t)
        l0.backward(); opt.collect_grads()
        g0 = opt.grad.clone()
        l1 = opt.functional_loss(pb, t=t)
        l1.backward(); opt.collect_grads()
        assert torch.equal(model.y, y)
        assert abs(l0.item() - l1.item()) <= 2e-6 * abs(l0.item())
        assert torch.equal(opt.grad, g0), (attn, (opt.grad - g0).abs().max().item())
        # a factor on the loss (it reaches the pair kernels as a device scalar), and something else added to the molecule
        # vectors' gradients on the way back (the factor then goes on the head's share alone)
        for extra in (False, True):
            outs = []
            for fused in (False, True):
                if fused:
                    l = opt.functional_loss(pb, t=t)
                else:
                    l = model.loss(opt.functional_forward(pb), t)
                tot = l * 0.3 + (0.01 * (model.g1.sum() + model.g2.square().sum()) if extra else 0.0)
                tot.backward(); opt.collect_grads()
                outs.append(opt.grad.clone())
            scale = outs[0].abs().max().item()
            assert (outs[0] - outs[1]).abs().max().item() <= 3e-6 * scale, (attn, extra, (outs[0] - outs[1]).abs().max().item(), scale)

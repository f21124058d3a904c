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

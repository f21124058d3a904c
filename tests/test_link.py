"""Link predictors NTN / DistMult / SymMLP / HolE (models/mlp.py:48-193).

CPU: known answers of the oracle restatement (the fft route of HolE.circular_correlation against the direct
circular sum; BilinearDiag against the explicit diagonal Bilinear; tiny hand-computed cases).
GPU: the HIP pair-feature kernels + MLP tail against the oracle, values and all gradients."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O


def test_oracle_circular_correlation_is_the_direct_sum():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(3, 7, generator=g, dtype=torch.float64)
    b = torch.randn(3, 7, generator=g, dtype=torch.float64)
    c = O.circular_correlation(a, b)
    want = torch.zeros_like(c)
    for k in range(7):
        for i in range(7):
            want[:, k] += a[:, i] * b[:, (i + k) % 7]
    assert torch.allclose(c, want, atol=1e-12)
    # known answer: correlating with a one-hot shifts
    e = torch.zeros(1, 7, dtype=torch.float64); e[0, 2] = 1.0
    assert torch.allclose(O.circular_correlation(e, b[:1]), torch.roll(b[:1], -2, dims=1), atol=1e-12)


def test_oracle_distmult_is_weighted_elementwise_product():
    dr = O._Draw(3, torch.float64, 0.1)
    O.init_link(dr, "mlp/", "distmult", 6, 2, (5,), feat_dim=4)
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(4, 6, generator=g, dtype=torch.float64); x2 = torch.randn(4, 6, generator=g, dtype=torch.float64)
    feat = (x1 * x2) @ dr.p["mlp/dm_layer/W"].t()
    want = O.linear(torch.relu(O.linear(feat, dr.p["mlp/mlp_layers/0/W"], dr.p["mlp/mlp_layers/0/b"])),
                    dr.p["mlp/l_out/W"], dr.p["mlp/l_out/b"])
    assert torch.allclose(O.distmult_forward(dr.p, x1, x2, 1), want, atol=1e-12)


def test_oracle_symmlp_swaps():
    dr = O._Draw(4, torch.float64, 0.1)
    O.init_link(dr, "mlp/", "symmlp", 5, 3, (8, 4))
    g = torch.Generator().manual_seed(2)
    x1 = torch.randn(2, 5, generator=g, dtype=torch.float64); x2 = torch.randn(2, 5, generator=g, dtype=torch.float64)
    assert torch.allclose(O.symmlp_forward(dr.p, x1, x2, 2), O.symmlp_forward(dr.p, x2, x1, 2), atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,fp,C,hidden", [("ntn", 32, 1, (16,)), ("ntn", 128, 3, (32, 16)), ("distmult", 64, 1, (16,)),
                                              ("symmlp", 48, 2, (32, 16)), ("hole", 128, 1, (32, 16)), ("hole", 20, 37, (16,))])
def test_link_predictors_match_oracle(kind, fp, C, hidden):
    from bmp.predictor import build_link_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    dev = torch.device("cuda:0")
    dr = O._Draw(11, torch.float64, 0.1)
    O.init_link(dr, "", kind, fp, C, hidden)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    B = 70
    g = torch.Generator().manual_seed(5)
    x1 = torch.randn(B, fp, generator=g, dtype=torch.float64, requires_grad=True)
    x2 = torch.randn(B, fp, generator=g, dtype=torch.float64, requires_grad=True)
    fwd = dict(ntn=O.ntn_forward, distmult=O.distmult_forward, symmlp=O.symmlp_forward, hole=O.hole_forward)[kind]
    y_ref = fwd(p, x1, x2, len(hidden), prefix="")
    cy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    (y_ref * cy).sum().backward()

    lp = build_link_predictor({"distmult": "dist-mult"}.get(kind, kind), fp, C, hidden).to(dev)
    load_param_dict(lp, p)
    a = x1.detach().float().to(dev).requires_grad_()
    b = x2.detach().float().to(dev).requires_grad_()
    y = lp(a, b)
    (y * cy.float().to(dev)).sum().backward()

    from parity_util import close
    close(y, y_ref, "y")
    close(a.grad, x1.grad, "dx1")
    close(b.grad, x2.grad, "dx2")
    for name, gr in grad_dict(lp).items():
        close(gr, p[name].grad, f"grad {name}")


def test_classifier_form_of_the_mlp_on_host_tensors_is_forward_plus_loss():
    """MLP.forward_loss (the reference's Classifier around the link predictor, train_ddi_modify.py:284-286) on host tensors
    takes the plain ops: the same loss and logits as forward + sigmoid_cross_entropy, labels of -1 left out."""
    import torch
    from bmp.mlp import MLP, sigmoid_cross_entropy
    torch.manual_seed(0)
    mlp = MLP(3, (32, 16), in_dim=24)
    g1, g2 = torch.randn(7, 12, requires_grad=True), torch.randn(7, 12, requires_grad=True)
    t = torch.randint(-1, 2, (7, 3), dtype=torch.int32)
    loss, y = mlp.forward_loss(g1, g2, t)
    y0 = mlp(g1, g2)
    assert torch.equal(y, y0) and torch.equal(loss, sigmoid_cross_entropy(y0, t))
    loss.backward()
    assert g1.grad is not None and torch.isfinite(g1.grad).all()

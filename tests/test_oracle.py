"""CPU tests of the oracle: golden vectors + the known-answer tests of SURVEY.md section 7.

The reference has no tests or fixtures for this path, so
the oracle is pinned by algebraic invariants that do not depend on any
implementation, and by the committed float64 vectors in tests/golden/.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from bmp import synth

T = torch.from_numpy


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    p = {k[len("param:"):]: T(z[k]).requires_grad_() for k in z.files if k.startswith("param:")}
    g = {k[len("grad:"):]: T(z[k]) for k in z.files if k.startswith("grad:")}
    return z, p, g


def _check_grads(loss, p, gref):
    names = sorted(p)
    gs = torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True)
    for n, g in zip(names, gs):
        g = torch.zeros_like(p[n]) if g is None else g
        assert torch.allclose(g, gref[n], rtol=1e-10, atol=1e-12), n


def test_golden_ggnn_small(golden_dir):
    z, p, gref = _load(golden_dir, "ggnn_small.npz")
    g, at = O.ggnn_forward(p, T(z["atoms"]), T(z["adj"]).double(), 3)
    assert torch.allclose(g, T(z["g"]), rtol=1e-12, atol=1e-13)
    assert torch.allclose(at, T(z["atom_out"]), rtol=1e-12, atol=1e-13)
    loss = (g * T(z["cg"])).sum() + 0.1 * (at * T(z["ca"])).sum()
    _check_grads(loss, p, gref)


def test_golden_ggnn_untied(golden_dir):
    z, p, gref = _load(golden_dir, "ggnn_untied_pad10.npz")
    g, at = O.ggnn_forward(p, T(z["atoms"]), T(z["adj"]).double(), 2, weight_tying=False)
    assert torch.allclose(g, T(z["g"]), rtol=1e-12, atol=1e-13)
    _check_grads((g * T(z["cg"])).sum(), p, gref)


@pytest.mark.parametrize("name,enc,nl", [("pair_nie_small.npz", "ggnn", 2), ("pair_relgcn_small.npz", "relgcn", 3)])
def test_golden_pair(golden_dir, name, enc, nl):
    z, p, gref = _load(golden_dir, name)
    y, g1, g2 = O.pair_forward(p, T(z["atoms_1"]), T(z["adj_1"]).double(), T(z["atoms_2"]), T(z["adj_2"]).double(),
                               encoder=enc, n_layers=nl, attn="nie")
    assert torch.allclose(y, T(z["y"]), rtol=1e-12, atol=1e-13)
    assert torch.allclose(g1, T(z["g1"]), rtol=1e-12, atol=1e-13)
    loss = O.sigmoid_cross_entropy(y, T(z["label"]))
    assert torch.allclose(loss, T(z["loss"]), rtol=1e-12)
    _check_grads(loss, p, gref)


ATTN_GOLDEN = ["deep", "extreme-deep", "fourier", "circ", "pool", "parallel", "alternating", "global", "neural", "bimpm"]


@pytest.mark.parametrize("attn", ATTN_GOLDEN)
def test_golden_pair_other_coattention(golden_dir, attn):
    z, p, gref = _load(golden_dir, f"pair_attn_{attn.replace('-', '_')}.npz")
    y, g1, g2 = O.pair_forward(p, T(z["atoms_1"]), T(z["adj_1"]).double(), T(z["atoms_2"]), T(z["adj_2"]).double(),
                               n_layers=2, attn=attn)
    assert torch.allclose(y, T(z["y"]), rtol=1e-12, atol=1e-13)
    assert torch.allclose(g1, T(z["g1"]), rtol=1e-12, atol=1e-13) and torch.allclose(g2, T(z["g2"]), rtol=1e-12, atol=1e-13)
    loss = O.sigmoid_cross_entropy(y, T(z["label"]))
    assert torch.allclose(loss, T(z["loss"]), rtol=1e-12)
    _check_grads(loss, p, gref)


# ------------------------------------------------------------------------------------------ KATs
def _ggnn_params(d=8, o=8, n_layers=3, seed=3, **kw):
    dr = O._Draw(seed, torch.float64, 0.2)
    O.init_ggnn(dr, "", o, d, n_layers, **kw)
    return dr.p


def _mols(n=6, seed=4):
    return synth.make_store(n, seed=seed, n_lo=3, n_hi=9, n_mean=6)


def test_kat_zero_bond_molecule():
    """(i) no bonds => m == 0 and h1 = z * h_bar from biases + embedding only."""
    p = _ggnn_params(n_layers=1)
    atoms = T(np.array([[6, 8, 7]], np.int32))
    adj = torch.zeros(1, 4, 3, 3, dtype=torch.float64)
    _, at = O.ggnn_forward(p, atoms, adj, 1)
    h = p["embed/W"][atoms.long()][0]
    x = torch.cat((h, torch.zeros_like(h)), 1)
    z = torch.sigmoid(x @ p["update_layer/W_z/W"].t() + p["update_layer/W_z/b"])
    c = torch.tanh(x @ p["update_layer/W/W"].t() + p["update_layer/W/b"])
    assert torch.allclose(at[0], z * c, atol=1e-14)


def test_kat_padding_affine_law():
    """(ii) g(A+1) - g(A) is the same vector for every molecule: the pad-atom readout term."""
    p = _ggnn_params()
    mols = _mols()
    a, adj = synth.concat_mols(mols)
    A = a.shape[1]
    gs = []
    for extra in (0, 1, 2):
        ap = np.zeros((len(mols), A + extra), np.int32); ap[:, :A] = a
        jp = np.zeros((len(mols), 4, A + extra, A + extra), np.float32); jp[:, :, :A, :A] = adj
        gs.append(O.ggnn_forward(p, T(ap), T(jp).double(), 3)[0])
    d1 = gs[1] - gs[0]
    assert torch.allclose(d1, d1[0:1].expand_as(d1), atol=1e-12)
    assert torch.allclose(gs[2] - gs[1], d1, atol=1e-12)
    assert d1.abs().max() > 1e-3          # padding really changes the result (no mask)


def test_kat_permutation():
    """(iii) atom permutation: atoms equivariant, g invariant."""
    p = _ggnn_params()
    m = _mols(1, seed=9)[0]
    a, adj = synth.concat_mols([m])
    perm = np.random.RandomState(0).permutation(m.n)
    g0, at0 = O.ggnn_forward(p, T(a), T(adj).double(), 3)
    g1, at1 = O.ggnn_forward(p, T(a[:, perm]), T(adj[:, :, perm][:, :, :, perm]).double(), 3)
    assert torch.allclose(g0, g1, atol=1e-12)
    assert torch.allclose(at0[:, perm], at1, atol=1e-12)


def test_kat_edge_type_layout():
    """(iv) message output feature k = 4*c + e: with W = 0 and a one-hot bias on feature
    4*c0 + e0, only bonds of type e0 deliver, into channel c0, one unit per bond."""
    d = 4
    h = torch.randn(1, 3, d, dtype=torch.float64)
    for e0 in range(4):
        for c0 in range(d):
            W = torch.zeros(4 * d, d, dtype=torch.float64)
            b = torch.zeros(4 * d, dtype=torch.float64); b[4 * c0 + e0] = 1.0
            adj = torch.zeros(1, 4, 3, 3, dtype=torch.float64)
            adj[0, e0, 0, 1] = adj[0, e0, 1, 0] = 1.0
            adj[0, (e0 + 1) % 4, 1, 2] = adj[0, (e0 + 1) % 4, 2, 1] = 1.0
            m = O.ggnn_message(h, adj, W, b)
            exp = torch.zeros(1, 3, d, dtype=torch.float64); exp[0, 0, c0] = 1; exp[0, 1, c0] = 1
            assert torch.equal(m, exp)


def test_kat_first_step_gru_has_no_U_terms():
    """(v) the first GRU call after reset ignores U_r/U_z/U and their biases."""
    p = _ggnn_params(n_layers=1)
    a, adj = synth.concat_mols(_mols(2))
    g0, _ = O.ggnn_forward(p, T(a), T(adj).double(), 1)
    q = dict(p)
    for n in ("U_r", "U_z", "U"):
        q[f"update_layer/{n}/W"] = torch.randn_like(p[f"update_layer/{n}/W"])
        q[f"update_layer/{n}/b"] = torch.randn_like(p[f"update_layer/{n}/b"])
    q["update_layer/W_r/W"] = torch.randn_like(p["update_layer/W_r/W"])
    g1, _ = O.ggnn_forward(q, T(a), T(adj).double(), 1)
    assert torch.equal(g0, g1)
    g2, _ = O.ggnn_forward(q, T(a), T(adj).double(), 2)       # second step does see them
    g3, _ = O.ggnn_forward(p, T(a), T(adj).double(), 2)
    assert not torch.allclose(g2, g3)


def _nie_params(d=6, o=5, head=8, seed=5):
    dr = O._Draw(seed, torch.float64, 0.2)
    O.init_nie(dr, "", d, o, head)
    return dr.p


def test_kat_coattention_orientation():
    """(vi) C is (mb, N2, N1): C[b,i,j] = act(a1[j]^T W a2[i] + a1[j].V1 + a2[i].V2 + c);
    L_2 normalises over i, L_1 over j."""
    p = _nie_params()
    a1 = torch.randn(2, 3, 6, dtype=torch.float64); a2 = torch.randn(2, 4, 6, dtype=torch.float64)
    C = O.fine_energy(p, "", a1, a2, "tanh")
    assert C.shape == (2, 4, 3)
    W = p["energy_layer/W"][:, :, 0]
    for b in range(2):
        for i in range(4):
            for j in range(3):
                e = a1[b, j] @ W @ a2[b, i] + a1[b, j] @ p["energy_layer/V1"][:, 0] \
                    + a2[b, i] @ p["energy_layer/V2"][:, 0] + p["energy_layer/b"][0]
                assert abs(C[b, i, j] - torch.tanh(e)) < 1e-13
    assert torch.allclose(torch.softmax(C, dim=1).sum(dim=1), torch.ones(2, 3, dtype=torch.float64))


def test_kat_coattention_swap():
    """(vii) with side-symmetric weights, swapping the sides swaps the outputs."""
    p = _nie_params()
    W = p["energy_layer/W"][:, :, 0]
    p["energy_layer/W"] = ((W + W.t()) / 2)[:, :, None]
    p["energy_layer/V2"] = p["energy_layer/V1"].clone()
    p["lt_layer_2/W"] = p["lt_layer_1/W"].clone()
    p["attention_layer_2/W"] = p["attention_layer_1/W"].clone()
    a1 = torch.randn(2, 3, 6, dtype=torch.float64); a2 = torch.randn(2, 5, 6, dtype=torch.float64)
    c1, c2 = O.nie_coattention(p, a1, a2)
    d1, d2 = O.nie_coattention(p, a2, a1)
    assert torch.allclose(c1, d2, atol=1e-13) and torch.allclose(c2, d1, atol=1e-13)


def test_rescale_adj_column_degree():
    a, adj = synth.concat_mols(_mols(3))
    adj = T(adj).double()
    r = O.rescale_adj(adj)
    deg = adj.sum(dim=(1, 2))
    col = r.sum(dim=(1, 2))
    assert torch.allclose(col[deg > 0], torch.ones_like(col[deg > 0]))
    assert torch.equal(col[deg == 0], torch.zeros_like(col[deg == 0]))


def test_sigmoid_cross_entropy_ignores_minus_one():
    y = torch.tensor([[0.3], [-1.2], [2.0]], dtype=torch.float64)
    t = torch.tensor([[1], [-1], [0]])
    ref = (torch.nn.functional.softplus(y[0]) - y[0] + torch.nn.functional.softplus(y[2])) / 2
    assert torch.allclose(O.sigmoid_cross_entropy(y, t), ref[0])


def test_modular_ggnn_untied_takes_first_branch():
    """models/models/ggnn.py: each GGNNUpdate owns a GRU, so untied layers never see U."""
    dr = O._Draw(1, torch.float64, 0.2)
    O.init_ggnn_modular(dr, 6, 6, 2, weight_tying=False)
    p = dr.p
    a, adj = synth.concat_mols(_mols(2))
    g0, _ = O.ggnn_modular_forward(p, T(a), T(adj).double(), 2, weight_tying=False)
    q = dict(p)
    q["update_layers/1/update_layer/U/W"] = torch.randn_like(p["update_layers/1/update_layer/U/W"])
    g1, _ = O.ggnn_modular_forward(q, T(a), T(adj).double(), 2, weight_tying=False)
    assert torch.equal(g0, g1)


def test_circular_parallel_against_explicit_rotation_sum():
    """CircularParallelCoattention (parallel_coattention.py:87-187): the fft form of the oracle against the defining
    sum gate[k] = tanh(sum_t j[t] * g[(t + k) mod o])."""
    torch.manual_seed(3)
    mb, n1, n2, d, o = 3, 5, 4, 6, 8
    p = {"j_layer/W": torch.randn(o, d, dtype=torch.float64), "j_layer/b": torch.randn(o, dtype=torch.float64)}
    a1, a2 = torch.randn(mb, n1, d, dtype=torch.float64), torch.randn(mb, n2, d, dtype=torch.float64)
    g1, g2 = torch.randn(mb, o, dtype=torch.float64), torch.randn(mb, o, dtype=torch.float64)
    c1, c2 = O.circular_parallel_coattention(p, a1, g1, a2, g2)
    for atoms, g, c in ((a1, g2, c1), (a2, g1, c2)):
        J = atoms @ p["j_layer/W"].t() + p["j_layer/b"]
        ref = torch.zeros(mb, o, dtype=torch.float64)
        for k in range(o):
            gate = torch.tanh((J * torch.roll(g, -k, dims=1)[:, None, :]).sum(-1))     # (mb, n)
            ref[:, k] = (gate * J[:, :, k]).sum(1)
        assert torch.allclose(c, ref, atol=1e-12)


@pytest.mark.parametrize("attn", ["deep", "very-deep", "extreme-deep", "fourier", "circ"])
def test_pair_forward_with_the_added_coattention_variants(attn):
    from bmp import synth
    store = synth.make_store(6, seed=2, n_lo=3, n_hi=9, n_mean=6)
    a1, j1 = synth.concat_mols(store[:3]); a2, j2 = synth.concat_mols(store[3:])
    p = O.make_pair_params(5, hidden_dim=8, out_dim=8, n_layers=2, attn=attn, dtype=torch.float64)
    y, g1, g2 = O.pair_forward(p, torch.from_numpy(a1), torch.from_numpy(j1).double(), torch.from_numpy(a2),
                               torch.from_numpy(j2).double(), n_layers=2, attn=attn)
    assert y.shape == (3, 1) and g1.shape == (3, 8) and torch.isfinite(y).all()


# ---- BiMPM (models/coattention/bimpm.py): known answers of the restatement ----
def _bimpm_params(H, d, seed=0):
    rs = np.random.RandomState(seed)
    return {f"{n}": T(rs.normal(size=(H, d))) for n in ("max_pooling_W", "att_mean_W", "att_max_W")}


def test_bimpm_single_atom_pair_by_hand():
    """One atom per molecule: every max / mean is over one element, so all three matchings are closed forms."""
    H, d = 3, 4
    p = _bimpm_params(H, d)
    rs = np.random.RandomState(1)
    x, y = rs.normal(size=d), rs.normal(size=d)
    m1, m2 = O.bimpm_coattention(p, T(x).view(1, 1, d), T(y).view(1, 1, d))
    eps = 1e-5
    m = lambda u, v: float(u @ v / ((np.linalg.norm(u) + eps) * (np.linalg.norm(v) + eps)))
    P, Q, R = (p[n].numpy() for n in ("max_pooling_W", "att_mean_W", "att_max_W"))
    att = m(x, y)
    mean2 = att * y / max(att, 1e-4); mean1 = att * x / max(att, 1e-4)          # div_with_small_value, bimpm.py:122-124
    want1 = [m(P[k] * x, P[k] * y) for k in range(H)] + [m(Q[k] * x, Q[0] * mean2) for k in range(H)] + \
            [m(R[k] * x, R[0] * (att * y)) for k in range(H)]                  # perspective 0 on the attended vector (:76-78)
    want2 = [m(P[k] * y, P[k] * x) for k in range(H)] + [m(Q[k] * y, Q[0] * mean1) for k in range(H)] + \
            [m(R[k] * y, R[0] * (att * x)) for k in range(H)]
    assert np.allclose(m1.numpy().ravel(), want1, rtol=1e-12) and np.allclose(m2.numpy().ravel(), want2, rtol=1e-12)


def test_bimpm_is_invariant_to_atom_order_and_swaps_with_the_sides():
    H, d = 4, 6
    p = _bimpm_params(H, d, 2)
    rs = np.random.RandomState(3)
    a1, a2 = T(rs.normal(size=(2, 5, d))), T(rs.normal(size=(2, 7, d)))
    m1, m2 = O.bimpm_coattention(p, a1, a2)
    q1, q2 = O.bimpm_coattention(p, a1[:, rs.permutation(5)], a2[:, rs.permutation(7)])
    assert torch.allclose(m1, q1, rtol=1e-12, atol=1e-13) and torch.allclose(m2, q2, rtol=1e-12, atol=1e-13)
    s2, s1 = O.bimpm_coattention(p, a2, a1)                                   # the module is symmetric in its two sides
    assert torch.allclose(m1, s1, rtol=1e-12, atol=1e-13) and torch.allclose(m2, s2, rtol=1e-12, atol=1e-13)
    assert m1.shape == (2, 3 * H)

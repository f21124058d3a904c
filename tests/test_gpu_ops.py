"""GPU parity tests of the HIP operators, called through the C ABI (bmp.functional).

Reference values: float64 torch on the CPU -- the packed-form restatement (tests/packed_ref.py)
for single operators on exactly the tensors the kernels see, and the dense oracle
(oracle/ref_cpu.py) for whole encoders.  Tolerance: max-abs error <= 1e-4 x max-abs reference
per tensor (BASELINE.json north_star: "within 1e-4 relative fp32"); index work is bit-exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O            # noqa: E402
from bmp import synth, packed              # noqa: E402
import packed_ref as PR                    # noqa: E402

TOL = 1e-4
T = torch.from_numpy


def dev():
    assert torch.cuda.is_available(), "GPU test selected without a GPU"
    return torch.device("cuda:0")


def close(got, ref, name="", tol=TOL, floor=1e-6):
    """max|got - ref| <= tol * max(max|ref|, floor).  ``floor`` is for quantities that are
    analytically zero (their natural scale is not in the reference value).  Logs the achieved error (parity_util)."""
    from parity_util import close as _c
    return _c(got, ref, name, tol, floor)


@pytest.fixture(scope="module")
def fn():
    from bmp import functional
    from bmp import _lib
    assert _lib.lib().bmp_tile_rows() == 128
    return functional


@pytest.fixture(scope="module")
def batch():
    store = synth.make_store(48, seed=5, n_lo=2, n_hi=40, n_mean=12)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(7)
    i1, i2 = rs.randint(0, 48, 21), rs.randint(0, 48, 21)
    pb = packed.pack_from_store(ms, [i1, i2], device="cpu", with_dense_map=True)
    return store, i1, i2, pb


def to_dev(pb):
    import dataclasses
    kw = {}
    for f in dataclasses.fields(pb):
        v = getattr(pb, f.name)
        if isinstance(v, torch.Tensor):
            kw[f.name] = v.to(dev())
        elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            kw[f.name] = [x.to(dev()) for x in v]
    return dataclasses.replace(pb, **kw, _cache={})


# --------------------------------------------------------------------------------- row GEMM
@pytest.mark.parametrize("K,Nout", [(8, 8), (16, 32), (64, 40), (72, 64), (128, 128), (136, 136), (512, 128),
                                    (128, 384), (24, 1)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_rows_fwd_bwd(fn, K, Nout, act):
    g = torch.Generator().manual_seed(K * 1000 + Nout)
    N = 384
    X = torch.randn(N, K, generator=g, dtype=torch.float64)
    W = torch.randn(K, Nout, generator=g, dtype=torch.float64) / np.sqrt(K)
    b = torch.randn(Nout, generator=g, dtype=torch.float64)
    c = torch.randn(N, Nout, generator=g, dtype=torch.float64)
    Xr, Wr, br = (t.clone().requires_grad_() for t in (X, W, b))
    pre = Xr @ Wr + br
    Yr = [pre, torch.sigmoid(pre), torch.tanh(pre)][act]
    (Yr * c).sum().backward()
    Xd, Wd, bd = (t.float().to(dev()).requires_grad_() for t in (X, W, b))
    Y = fn.LinearRowsFn.apply(Xd, Wd, bd, act)
    close(Y, Yr, "Y")
    (Y * c.float().to(dev())).sum().backward()
    close(Xd.grad, Xr.grad, "dX")
    close(Wd.grad, Wr.grad, "dW")
    close(bd.grad, br.grad, "db")


def test_linear_rows_asymmetric_identity(fn):
    """A = I with an asymmetric B: catches a transposed C-write (cdna guide section 3)."""
    N, K = 128, 128
    X = torch.eye(N, K)
    W = (torch.arange(K)[:, None] * 1000 + torch.arange(K)[None, :]).float()
    Y = fn.LinearRowsFn.apply(X.to(dev()), W.to(dev()), None, 0)
    assert torch.equal(Y.cpu(), W)


def test_linear_rejects_bad_shapes(fn):
    with pytest.raises(ValueError):
        fn.LinearRowsFn.apply(torch.zeros(100, 8, device=dev()), torch.zeros(8, 8, device=dev()), None, 0)
    with pytest.raises(ValueError):
        fn.LinearRowsFn.apply(torch.zeros(128, 12, device=dev()), torch.zeros(12, 8, device=dev()), None, 0)
    with pytest.raises(ValueError):
        fn.LinearRowsFn.apply(torch.zeros(128, 8), torch.zeros(8, 8), None, 0)       # CPU tensor


# --------------------------------------------------------------------------------- embedding
def test_embed_fwd_bwd(fn, batch):
    _, _, _, pb = batch
    pbd = to_dev(pb)
    d = 24
    W = torch.randn(117, d, dtype=torch.float64)
    ids = pb.atom_id.long()
    Wr = W.clone().requires_grad_()
    c = torch.randn(pb.n_rows, d, dtype=torch.float64)
    (Wr[ids] * c).sum().backward()
    Wd = W.float().to(dev()).requires_grad_()
    out = fn.EmbedFn.apply(Wd, pbd.atom_id)
    assert torch.equal(out.cpu(), W.float()[ids])                  # pure gather: bit-exact
    (out * c.float().to(dev())).sum().backward()
    close(Wd.grad, Wr.grad, "dW_emb")


# --------------------------------------------------------------------------------- message
@pytest.mark.parametrize("d_in,d_out,self_conn", [(16, 16, False), (128, 128, False), (16, 24, True), (40, 8, True)])
def test_msg_fwd_bwd(fn, batch, d_in, d_out, self_conn):
    _, _, _, pb = batch
    pbd = to_dev(pb)
    g = torch.Generator().manual_seed(d_in + d_out)
    N = pb.n_rows
    x = torch.randn(N, d_in, generator=g, dtype=torch.float64)
    W = torch.randn(4 * d_out, d_in, generator=g, dtype=torch.float64) / np.sqrt(d_in)
    b = torch.randn(4 * d_out, generator=g, dtype=torch.float64)
    Ws = torch.randn(d_out, d_in, generator=g, dtype=torch.float64) / np.sqrt(d_in)
    bs = torch.randn(d_out, generator=g, dtype=torch.float64)
    c = torch.randn(N, d_out, generator=g, dtype=torch.float64)
    xr, Wr, br, Wsr, bsr = (t.clone().requires_grad_() for t in (x, W, b, Ws, bs))
    ref = PR.message(pb, xr, Wr, br)
    if self_conn:
        ref = torch.tanh(ref + xr @ Wsr.t() + bsr)
    (ref * c).sum().backward()

    from bmp.ggnn import Linear, message_kernel_weights
    lin = Linear(d_in, 4 * d_out).to(dev())
    with torch.no_grad():
        lin.W.copy_(W.float()); lin.b.copy_(b.float())
    WT, bE = message_kernel_weights(lin)
    xd = x.float().to(dev()).requires_grad_()
    Wsd = Ws.float().to(dev()).requires_grad_()
    bsd = bs.float().to(dev()).requires_grad_()
    out = fn.MsgFn.apply(xd, WT, bE, Wsd.t() if self_conn else None, bsd if self_conn else None, pbd,
                         2 if self_conn else 0)
    close(out, ref, "msg out")
    (out * c.float().to(dev())).sum().backward()
    close(xd.grad, xr.grad, "dx")
    close(lin.W.grad, Wr.grad, "dW_msg")
    close(lin.b.grad, br.grad, "db_msg")
    if self_conn:
        close(Wsd.grad, Wsr.grad, "dW_self")
        close(bsd.grad, bsr.grad, "db_self")


def test_gather_is_exact_on_integers(fn, batch):
    """Integer-valued features: the neighbour sum must be bit-exact (index work)."""
    _, _, _, pb = batch
    pbd = to_dev(pb)
    d = 8
    x = torch.randint(-8, 9, (pb.n_rows, d)).double()
    agg, wdeg = PR.gather_agg(pb, x)
    WT = torch.zeros(4 * d, 4 * d); WT[torch.arange(4 * d), torch.arange(4 * d)] = 1.0      # identity: out = agg
    out = fn.MsgFn.apply(x.float().to(dev()), WT.to(dev()), torch.zeros(4, 4 * d, device=dev()), None, None, pbd, 0)
    assert torch.equal(out.cpu().double(), agg)


# --------------------------------------------------------------------------------- GRU
@pytest.mark.parametrize("d", [16, 128])
@pytest.mark.parametrize("first", [True, False])
def test_gru_fwd_bwd(fn, batch, d, first):
    _, _, _, pb = batch
    pbd = to_dev(pb)
    dr = O._Draw(d + int(first), torch.float64, 0.3)
    O.init_ggnn(dr, "", d, d, 1)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    g = torch.Generator().manual_seed(d)
    N = pb.n_rows
    h = torch.randn(N, d, generator=g, dtype=torch.float64)
    m = torch.randn(N, d, generator=g, dtype=torch.float64)
    c = torch.randn(N, d, generator=g, dtype=torch.float64)
    hr, mr = h.clone().requires_grad_(), m.clone().requires_grad_()
    ref, _ = PR.gru(p, "update_layer", hr, mr, first)
    (ref * c).sum().backward()

    from bmp.ggnn import GRU
    from bmp.snapshot import load_param_dict
    gru = GRU(2 * d, d).to(dev())
    load_param_dict(gru, {k: v for k, v in p.items() if k.startswith("update_layer/")}, prefix="update_layer/")
    AT, UcT, b = gru.kernel_weights(first)
    hd, md = h.float().to(dev()).requires_grad_(), m.float().to(dev()).requires_grad_()
    out = fn.GRUFn.apply(hd, md, AT, UcT, b, pbd, first)
    close(out, ref, "gru out")
    (out * c.float().to(dev())).sum().backward()
    close(hd.grad, hr.grad, "dh")
    close(md.grad, mr.grad, "dm")
    for n in ("W_r", "W_z", "W", "U_r", "U_z", "U"):
        for wb in ("W", "b"):
            ref_g = p[f"update_layer/{n}/{wb}"].grad
            got = getattr(getattr(gru, n), wb).grad
            if ref_g is None:                                   # first call: U_* and W_r unused
                assert got is None or float(got.abs().max()) == 0.0, (n, wb)
            else:
                close(got, ref_g, f"d{n}.{wb}")


# --------------------------------------------------------------------------------- whole encoder
def _encoder_case(d, o, n_layers, tying, batch, seed=777):
    store, i1, i2, pb = batch
    dr = O._Draw(seed, torch.float64, 0.1)
    O.init_ggnn(dr, "", o, d, n_layers, weight_tying=tying)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    outs = []
    for idx in (i1, i2):
        a, j = synth.concat_mols([store[k] for k in idx])
        outs.append(O.ggnn_forward(p, T(a), T(j).double(), n_layers, weight_tying=tying))
    return p, outs


@pytest.mark.parametrize("d,o,n_layers,tying", [(16, 16, 2, True), (128, 128, 4, True), (32, 24, 3, False)])
def test_ggnn_encoder_matches_dense_oracle(fn, batch, d, o, n_layers, tying):
    store, i1, i2, pb = batch
    pbd = to_dev(pb)
    p, ((g1, at1), (g2, at2)) = _encoder_case(d, o, n_layers, tying, batch)
    g_ref = torch.cat((g1, g2))
    cg = torch.randn(g_ref.shape, dtype=torch.float64)
    ca1 = torch.randn(at1.shape, dtype=torch.float64); ca2 = torch.randn(at2.shape, dtype=torch.float64)
    ((g_ref * cg).sum() + 0.1 * (at1 * ca1).sum() + 0.1 * (at2 * ca2).sum()).backward()

    from bmp.ggnn import GGNN
    from bmp.snapshot import load_param_dict, grad_dict
    enc = GGNN(out_dim=o, hidden_dim=d, n_layers=n_layers, weight_tying=tying).to(dev())
    load_param_dict(enc, p)
    g = enc(pbd)
    atoms = enc.get_atom_array()
    close(g, g_ref, "g")
    close(atoms.dense(0), at1, "atoms side 1")
    close(atoms.dense(1), at2, "atoms side 2")
    loss = (g * cg.float().to(dev())).sum() + 0.1 * (atoms.dense(0) * ca1.float().to(dev())).sum() \
        + 0.1 * (atoms.dense(1) * ca2.float().to(dev())).sum()
    loss.backward()
    for name, gr in grad_dict(enc).items():
        close(gr, p[name].grad, f"grad {name}")


def test_ggnn_dense_input_golden(fn, golden_dir):
    """Drop-in form: dense (atoms, adj) arrays in, committed golden vectors out."""
    import os
    from bmp.ggnn import GGNN
    from bmp.snapshot import load_param_dict, grad_dict
    z = np.load(os.path.join(golden_dir, "ggnn_small.npz"))
    p = {k[6:]: z[k] for k in z.files if k.startswith("param:")}
    enc = GGNN(out_dim=8, hidden_dim=8, n_layers=3).to(dev())
    load_param_dict(enc, p)
    g = enc(torch.from_numpy(z["atoms"]), torch.from_numpy(z["adj"]))
    at = enc.get_atom_array().dense()
    close(g, T(z["g"]), "g")
    close(at, T(z["atom_out"]), "atoms")
    loss = (g * T(z["cg"]).float().to(dev())).sum() + 0.1 * (at * T(z["ca"]).float().to(dev())).sum()
    close(loss, T(z["loss"]), "loss")
    loss.backward()
    for name, gr in grad_dict(enc).items():
        close(gr, T(z["grad:" + name]), f"grad {name}")


def test_ggnn_untied_pad10_golden(fn, golden_dir):
    import os
    from bmp.ggnn import GGNN
    from bmp.snapshot import load_param_dict, grad_dict
    z = np.load(os.path.join(golden_dir, "ggnn_untied_pad10.npz"))
    p = {k[6:]: z[k] for k in z.files if k.startswith("param:")}
    enc = GGNN(out_dim=8, hidden_dim=8, n_layers=2, weight_tying=False).to(dev())
    load_param_dict(enc, p)
    g = enc(torch.from_numpy(z["atoms"]), torch.from_numpy(z["adj"]))
    close(g, T(z["g"]), "g")
    close(enc.get_atom_array().dense(), T(z["atom_out"]), "atoms")
    (g * T(z["cg"]).float().to(dev())).sum().backward()
    for name, gr in grad_dict(enc).items():
        close(gr, T(z["grad:" + name]), f"grad {name}")


def test_unsupported_options_raise():
    from bmp.ggnn import GGNN
    with pytest.raises(NotImplementedError):
        GGNN(16, 16, use_attention=True)
    with pytest.raises(ValueError):
        GGNN(16, 16, message_function="nope")


# --------------------------------------------------------------------------------- fused step kernel
@pytest.mark.parametrize("d", [32, 64, 128])
@pytest.mark.parametrize("first", [True, False])
def test_fused_step_fwd_bwd(fn, batch, d, first):
    """bmp_ggnn_step_* (message + GRU in one kernel per tile) vs the packed float64 restatement.  d = 32: the one-wave-per-block
    kernels of bmp_fused_small.hip (the width of the reference's published models, DDI.md:6)."""
    assert fn.step_supported(d)
    _, _, _, pb = batch
    pbd = to_dev(pb)
    dr = O._Draw(d + 7 * int(first), torch.float64, 0.3)
    O.init_ggnn(dr, "", d, d, 1)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    g = torch.Generator().manual_seed(d + 1)
    N = pb.n_rows
    h = torch.randn(N, d, generator=g, dtype=torch.float64)
    c = torch.randn(N, d, generator=g, dtype=torch.float64)
    hr = h.clone().requires_grad_()
    m = PR.message(pb, hr, p["message_layers/0/W"], p["message_layers/0/b"])
    ref, _ = PR.gru(p, "update_layer", hr, m, first)
    (ref * c).sum().backward()

    from bmp.ggnn import GGNN
    from bmp.snapshot import load_param_dict, grad_dict
    from bmp.ggnn import message_kernel_weights
    enc = GGNN(out_dim=d, hidden_dim=d, n_layers=1).to(dev())
    load_param_dict(enc, p)
    WT, bE = message_kernel_weights(enc.message_layers[0])
    AT, UcT, b = enc.update_layer.kernel_weights(first)
    hd = h.float().to(dev()).requires_grad_()
    out = fn.GGNNStepFn.apply(hd, WT, bE, AT, UcT, b, pbd, first)
    close(out, ref, "step out")
    (out * c.float().to(dev())).sum().backward()
    close(hd.grad, hr.grad, "dh")
    grads = grad_dict(enc)
    for name in ("message_layers/0/W", "message_layers/0/b"):
        close(grads[name], p[name].grad, f"grad {name}")
    for n in ("W_r", "W_z", "W", "U_r", "U_z", "U"):
        for wb in ("W", "b"):
            ref_g = p[f"update_layer/{n}/{wb}"].grad
            got = grads[f"update_layer/{n}/{wb}"]
            if ref_g is None:
                assert float(got.abs().max()) == 0.0, (n, wb)
            else:
                close(got, ref_g, f"grad {n}.{wb}")


def test_fused_and_unfused_encoders_agree(fn, batch):
    """Same encoder through the fused step kernel and through msg + GRU launches."""
    from bmp.ggnn import GGNN
    _, _, _, pb = batch
    pbd = to_dev(pb)
    torch.manual_seed(3)
    enc = GGNN(out_dim=128, hidden_dim=128, n_layers=3).to(dev())
    g1 = enc(pbd); a1 = enc.get_atom_array().rows
    enc.fused = False
    g2 = enc(pbd); a2 = enc.get_atom_array().rows
    close(g1, g2, "g fused vs unfused", tol=2e-5)
    close(a1, a2, "atoms fused vs unfused", tol=2e-5)


@pytest.mark.parametrize("d,o,n_layers,tying,fused", [(64, 32, 3, True, True), (128, 128, 2, True, False),
                                                      (64, 64, 2, False, True), (32, 16, 8, False, True), (32, 32, 3, True, True)])
def test_ggnn_encoder_fused_variants(fn, batch, d, o, n_layers, tying, fused):
    store, i1, i2, pb = batch
    pbd = to_dev(pb)
    p, ((g1, at1), (g2, at2)) = _encoder_case(d, o, n_layers, tying, batch, seed=31)
    g_ref = torch.cat((g1, g2))
    cg = torch.randn(g_ref.shape, dtype=torch.float64)
    (g_ref * cg).sum().backward()
    from bmp.ggnn import GGNN
    from bmp.snapshot import load_param_dict, grad_dict
    enc = GGNN(out_dim=o, hidden_dim=d, n_layers=n_layers, weight_tying=tying).to(dev())
    enc.fused = fused
    load_param_dict(enc, p)
    g = enc(pbd)
    close(g, g_ref, "g")
    close(enc.get_atom_array().dense(0), at1, "atoms")
    (g * cg.float().to(dev())).sum().backward()
    for name, gr in grad_dict(enc).items():
        close(gr, p[name].grad, f"grad {name}")


@pytest.mark.parametrize("d,with_h0,act", [(128, True, "identity"), (64, True, "tanh"), (128, False, "tanh")])
def test_readout_tile_kernel_equals_gemm_plus_segment_sum(fn, batch, d, with_h0, act):
    """bmp_readout_tile_fwd (one kernel per tile, sums taken in LDS) against bmp_readout_fwd (row GEMM + segment sum)
    on the same inputs, and the gradients that flow through the saved ij of either."""
    import dataclasses
    store, _i1, _i2, pb = batch
    pbd = to_dev(pb)
    assert pbd.row_mol is not None
    g = torch.Generator().manual_seed(d + with_h0)
    mk = lambda *s_: (torch.randn(*s_, generator=g) * 0.3).to(dev())
    h0_, x_ = mk(pb.n_rows, d), mk(pb.n_rows, d)
    WT_, b_ = mk(d * (2 if with_h0 else 1), 2 * d), mk(2 * d)
    cw = mk(pb.n_mols, d)
    res = []
    for tile in (True, False):
        pbx = pbd if tile else dataclasses.replace(pbd, row_mol=None, _cache={})       # no map -> the two-kernel path
        x, h0, WT, b = (t.clone().requires_grad_() for t in (x_, h0_, WT_, b_))
        y = fn.ReadoutFn.apply(x, h0 if with_h0 else None, WT, b, pbx, fn.ACT[act])
        (y * cw).sum().backward()
        res.append((y, x.grad, h0.grad if with_h0 else None, WT.grad, b.grad))
    rows = torch.zeros(pb.n_rows, dtype=torch.bool)
    for r0, nr in zip(pb.mol_row0.tolist(), pb.mol_nrows.tolist()):
        rows[r0:r0 + nr] = True
    rows = rows.to(dev())
    for name, a, b2 in zip(("g", "dh", "dh0", "dWT", "db"), res[0], res[1]):
        if a is None:
            continue
        if name in ("dh", "dh0"):
            a, b2 = a[rows], b2[rows]
        close(a, b2, name)


def test_gru_with_separate_state_fwd_bwd():
    """bmp_gru_state_fwd / _bwd (the later-call GRU with its state apart from its input: dropout on the step output,
    models/ggnn.py:626-627) against the update rule written out in float64 (SURVEY.md A.2 with x = [hd, m], state s)."""
    from bmp import functional as Fn, packed, synth
    store = synth.make_store(10, seed=21, n_lo=2, n_hi=24, n_mean=9)
    pb = packed.pack_from_store(packed.MolStore(store), [np.arange(10)], device=dev())
    d, N = 64, pb.n_rows
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 0.3
    ref = [mk(N, d), mk(N, d), mk(N, d), mk(2 * d, 3 * d), mk(d, 2 * d), mk(d, d), mk(3 * d)]
    cw = mk(N, d)
    hd, m, s, WT, UrzT, UcT, b = [t.requires_grad_() for t in ref]
    pre = torch.cat((hd, m), 1) @ WT
    r = torch.sigmoid(pre[:, :d] + s @ UrzT[:, :d] + b[:d])
    z = torch.sigmoid(pre[:, d:2 * d] + s @ UrzT[:, d:] + b[d:2 * d])
    c = torch.tanh(pre[:, 2 * d:] + (r * s) @ UcT + b[2 * d:])
    out = z * c + (1 - z) * s
    (out * cw).sum().backward()
    got = [t.detach().float().to(dev()).requires_grad_() for t in ref]
    o = Fn.GRUStateFn.apply(*got, pb)
    (o * cw.float().to(dev())).sum().backward()
    rel = lambda a, w: ((a.double().cpu() - w).abs().max() / w.abs().max()).item()
    assert rel(o.detach(), out.detach()) < 1e-5
    for name, a, w in zip("hd m s WT UrzT UcT b".split(), got, ref):
        assert rel(a.grad, w.grad) < 1e-5, name


def test_step_and_layer_forward_over_tile_ranges_equal_the_whole_launch(fn, batch):
    """bmp_ggnn_step_fwd / bmp_relgcn_layer_fwd over two tile ranges of the whole arrays (tile0 argument: the two chains of the
    planned encoder's forward) write exactly what one launch over all tiles writes -- a step is tile-local."""
    from bmp import _lib
    from bmp._lib import check, ptr, stream
    from bmp.functional import pack_k4
    L = _lib.lib()
    _, _, _, pb = batch
    pbd = to_dev(pb)
    d, N, T = 128, pb.n_rows, pb.n_tiles
    assert T >= 3
    g = torch.Generator().manual_seed(5)
    r = lambda *s: (0.2 * torch.randn(*s, generator=g)).to(dev())
    h = r(N, d)
    WTp, bE, ATp, UcTp, b = pack_k4(r(4 * d, d)), r(4, d), pack_k4(r(2 * d, 3 * d)), pack_k4(r(d, d)), r(3 * d)
    WsTp, bs = pack_k4(r(d, d)), r(d)
    f = lambda *s: torch.full(s, float("nan"), device=dev())

    def step(ranges):
        m, rz, c, hout = f(N, d), f(N, 2 * d), f(N, d), f(N, d)
        for t0, nt in ranges:
            check(L.bmp_ggnn_step_fwd(ptr(h), t0, nt, d, 0, ptr(pbd.csr_ptr), ptr(pbd.csr_col), ptr(pbd.csr_val), ptr(WTp), ptr(bE),
                                      ptr(ATp), ptr(UcTp), ptr(b), ptr(m), ptr(rz), ptr(c), ptr(hout), None, None, 0, 0, stream()), "step")
        return m, rz, c, hout

    def layer(ranges):
        out, wdeg = f(N, d), f(N, 4)
        for t0, nt in ranges:
            check(L.bmp_relgcn_layer_fwd(ptr(h), t0, nt, d, ptr(pbd.csr_ptr), ptr(pbd.csr_col), ptr(pbd.csr_val), ptr(WTp), ptr(bE),
                                         ptr(WsTp), ptr(bs), 2, ptr(out), ptr(wdeg), None, None, 0, stream()), "layer")
        return out, wdeg

    for run in (step, layer):
        whole = run([(0, T)])
        parts = run([(T // 3, T - T // 3), (0, T // 3)])
        for a, bb in zip(whole, parts):
            assert torch.isfinite(a).all() and torch.equal(a, bb)


def test_low_priority_stream_entry_points():
    """bmp_stream_create_low / bmp_stream_destroy: a usable stream handle of the device's lowest priority."""
    import ctypes
    from bmp import _lib
    from bmp._lib import check
    L = _lib.lib()
    hnd = ctypes.c_void_p()
    check(L.bmp_stream_create_low(ctypes.byref(hnd)), "bmp_stream_create_low")
    assert hnd.value
    s = torch.cuda.ExternalStream(hnd.value, device=dev())
    with torch.cuda.stream(s):
        x = torch.ones(1024, device=dev()) * 3
    s.synchronize()
    assert float(x.sum()) == 3072.0
    check(L.bmp_stream_destroy(hnd), "bmp_stream_destroy")
    assert L.bmp_stream_create_low(None) != 0

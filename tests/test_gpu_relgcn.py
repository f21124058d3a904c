"""GPU parity tests: RelGCN encoder (models/relgcn.py), modular GGNN (models/models/ggnn.py) and the
RelGCN + Nie pair predictor (BASELINE.json config 3), against the float64 dense oracle; tolerance 1e-4."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O            # noqa: E402
from bmp import synth, packed              # noqa: E402
from test_gpu_ops import close, dev, to_dev, T       # noqa: E402


@pytest.fixture(scope="module")
def pairs():
    store = synth.make_store(40, seed=9, n_lo=2, n_hi=30, n_mean=10)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(2)
    i1, i2 = rs.randint(0, 40, 13), rs.randint(0, 40, 13)
    return store, i1, i2, packed.pack_from_store(ms, [i1, i2], device="cpu", with_dense_map=True)


@pytest.mark.parametrize("ch_list,out,scale", [([16, 128, 64], 64, True), ([8, 8, 8, 8], 8, True), ([16, 24], 12, False),
                                               ([64, 64, 64], 32, True), ([128, 128, 128, 128], 128, True),    # fused layers
                                               ([16, 64, 64, 32], 16, False)])                                   # mixed
def test_relgcn_matches_dense_oracle(pairs, ch_list, out, scale):
    from bmp.relgcn import RelGCN
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    dr = O._Draw(3, torch.float64, 0.2)
    O.init_relgcn(dr, "", out, ch_list)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    g1, at1 = O.relgcn_forward(p, T(a1), T(j1).double(), len(ch_list) - 1, scale)
    g2, at2 = O.relgcn_forward(p, T(a2), T(j2).double(), len(ch_list) - 1, scale)
    g_ref = torch.cat((g1, g2))
    gen = torch.Generator().manual_seed(5)
    cg = torch.randn(g_ref.shape, dtype=torch.float64, generator=gen); ca = torch.randn(at1.shape, dtype=torch.float64, generator=gen)
    ((g_ref * cg).sum() + 0.1 * (at1 * ca).sum()).backward()
    enc = RelGCN(out_channels=out, ch_list=ch_list, scale_adj=scale).to(dev())
    load_param_dict(enc, p)
    g = enc(to_dev(pb))
    atoms = enc.get_atom_array()
    close(g, g_ref, "g"); close(atoms.dense(0), at1, "atoms 1"); close(atoms.dense(1), at2, "atoms 2")
    ((g * cg.float().to(dev())).sum() + 0.1 * (atoms.dense(0) * ca.float().to(dev())).sum()).backward()
    for name, gr in grad_dict(enc).items():
        close(gr, p[name].grad, f"grad {name}")


def test_rescale_adj_is_exact_index_work(pairs):
    """1/deg per source atom, bit-identical to the dense rescale_adj on the same inputs."""
    from bmp.relgcn import rescale_adj
    store, i1, i2, pb = pairs
    pbs = rescale_adj(to_dev(pb))
    import dataclasses
    host = dataclasses.replace(pb, csr_val=pbs.csr_val.cpu(), csrT_val=pbs.csrT_val.cpu(), _cache={})
    for s, idx in enumerate((i1, i2)):
        a, j = synth.concat_mols([store[k] for k in idx])
        ref = O.rescale_adj(T(j)).numpy()
        _, got = packed.unpack_to_dense(host, s)
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("tying,concat,act", [(True, False, "identity"), (False, True, "tanh")])
def test_modular_ggnn_matches_dense_oracle(pairs, tying, concat, act):
    from bmp.relgcn import GGNNModular
    from bmp.snapshot import load_param_dict, grad_dict
    store, i1, i2, pb = pairs
    d, o, nl = 16, 12, 3
    dr = O._Draw(4, torch.float64, 0.2)
    O.init_ggnn_modular(dr, o, d, nl, weight_tying=tying, concat_hidden=concat)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    g1, _ = O.ggnn_modular_forward(p, T(a1), T(j1).double(), nl, tying, concat, act)
    g2, _ = O.ggnn_modular_forward(p, T(a2), T(j2).double(), nl, tying, concat, act)
    g_ref = torch.cat((g1, g2))
    cg = torch.randn(g_ref.shape, dtype=torch.float64)
    (g_ref * cg).sum().backward()
    enc = GGNNModular(o, d, nl, concat_hidden=concat, weight_tying=tying, activation=act).to(dev())
    load_param_dict(enc, p)
    g = enc(to_dev(pb))
    close(g, g_ref, "g")
    (g * cg.float().to(dev())).sum().backward()
    for name, gr in grad_dict(enc).items():
        ref = p[name].grad
        if ref is None:          # untied layers never reach the U terms (first-call branch)
            assert gr is None or float(gr.abs().max()) == 0.0, name
            continue
        close(gr, ref, f"grad {name}")


def test_modular_ggnn_is_real_node_mask():
    from bmp.relgcn import GGNNModular
    from bmp.snapshot import load_param_dict
    store = synth.make_store(6, seed=1, n_lo=3, n_hi=9, n_mean=6)
    a, j = synth.concat_mols(store)
    mask = (a != 0).astype(np.float32)
    dr = O._Draw(4, torch.float64, 0.2)
    O.init_ggnn_modular(dr, 8, 8, 2)
    g_ref, _ = O.ggnn_modular_forward(dr.p, T(a), T(j).double(), 2, is_real_node=T(mask).double())
    enc = GGNNModular(8, 8, 2).to(dev())
    load_param_dict(enc, dr.p)
    close(enc(T(a), T(j), is_real_node=mask), g_ref, "masked g")


def test_pair_relgcn_golden(golden_dir):
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict, grad_dict
    z = np.load(os.path.join(golden_dir, "pair_relgcn_small.npz"))
    p = {k[6:]: z[k] for k in z.files if k.startswith("param:")}
    model = build_pair_predictor(hidden_dim=8, out_dim=8, n_layers=3, attn="nie", encoder="relgcn").to(dev())
    load_param_dict(model, p)
    y = model(T(z["atoms_1"]), T(z["adj_1"]), T(z["atoms_2"]), T(z["adj_2"]))
    close(y, T(z["y"]), "logits")
    loss = model.loss(y, T(z["label"]).to(dev()))
    close(loss, T(z["loss"]), "loss")
    loss.backward()
    for name, gr in grad_dict(model).items():
        close(gr, T(z["grad:" + name]), f"grad {name}")


@pytest.mark.parametrize("d,act", [(64, "tanh"), (128, "tanh"), (128, "identity")])
def test_fused_relgcn_layer_equals_the_unfused_path(pairs, d, act):
    """bmp_relgcn_layer_* (one kernel per tile) against bmp_msg_* (gather + row GEMMs) on the same inputs: outputs,
    input gradient and all four parameter gradients."""
    from bmp import functional as Fn
    from bmp.relgcn import rescale_adj
    store, i1, i2, pb = pairs
    pbd = rescale_adj(to_dev(pb))
    g = torch.Generator().manual_seed(d)
    mk = lambda *s_: (torch.randn(*s_, generator=g) * 0.2).to(dev())
    x0 = mk(pb.n_rows, d) * 3
    WT0, bE0, WsT0, bs0 = mk(4 * d, d), mk(4, d), mk(d, d), mk(d)
    cw = mk(pb.n_rows, d)
    res = []
    for fn in (Fn.RelLayerFn, Fn.MsgFn):
        x, WT, bE, WsT, bs = (t.clone().requires_grad_() for t in (x0, WT0, bE0, WsT0, bs0))
        y = fn.apply(x, WT, bE, WsT, bs, pbd, Fn.ACT[act])
        (y * cw).sum().backward()
        res.append((y, x.grad, WT.grad, bE.grad, WsT.grad, bs.grad))
    rows = torch.zeros(pb.n_rows, dtype=torch.bool)
    for r0, nr in zip(pb.mol_row0.tolist(), pb.mol_nrows.tolist()):
        rows[r0:r0 + nr] = True                                  # rows of no molecule carry unspecified values
    rows = rows.to(dev())
    for name, a, b in zip(("out", "dx", "dWT", "dbE", "dWsT", "dbs"), res[0], res[1]):
        if name in ("out", "dx"):
            a, b = a[rows], b[rows]
        close(a, b, name)

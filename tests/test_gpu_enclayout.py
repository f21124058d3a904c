"""The encoder layout on the GPU (bmp/enclayout.py): real atoms + one pad row per tile, tiles of 1..4 live 32-row blocks,
optional de-duplication (SURVEY.md 8(d) caveat).  The device collate against its numpy statement (bitwise), the fused tile
kernels on short tiles against the dense oracle (logits and every gradient, all co-attention kinds incl. none), and the
planned step at the headline size against the per-instance planned step."""
import numpy as np
import pytest
import torch

from parity_util import close

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.mark.parametrize("dedup", [False, True])
def test_device_collate_equals_numpy_statement(dedup):
    from bmp import enclayout, packed, synth
    dev = torch.device("cuda:0")
    store = synth.make_store(120, seed=5, n_lo=1, n_hi=127, n_mean=25)
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, dev)
    rs = np.random.RandomState(3)
    for B, n_cu in ((300, 256), (17, 256), (300, 9), (1, 256)):
        sides = [rs.randint(0, 120, B), rs.randint(0, 120, B)]
        eb = enclayout.encode_from_store_device(ds, sides, dedup=dedup, n_cu=n_cu)
        ref = enclayout.encode_from_store(ms, sides, dedup=dedup, n_cu=n_cu)
        torch.cuda.synchronize()
        a, b = eb.pb_enc, ref.pb_enc
        assert (a.n_tiles, a.n_mols, a.n_mtiles, a.n_edges, a.n_real_atoms) == (b.n_tiles, b.n_mols, b.n_mtiles, b.n_edges, b.n_real_atoms)
        for k in ("atom_id", "row_w", "row_mol", "csr_ptr", "csr_col", "csr_val", "csrT_ptr", "csrT_col", "csrT_val", "mol_row0",
                  "mol_nrows", "mt_row0", "mt_nblk"):
            assert torch.equal(getattr(a, k).cpu(), getattr(b, k)), k
        for k in ("uid", "uptr", "uinst", "enc_row0", "enc_n", "enc_pad", "tptr", "tmols"):
            assert torch.equal(getattr(eb, k).cpu(), getattr(ref, k)), k
        assert eb.budget == ref.budget and eb.n_encoded == ref.n_encoded
        # expand / reduce against the host statement (the reduce is the expand's transpose)
        h = torch.randn(a.n_rows, 32, device=dev)
        X = enclayout.EncRowsFn.apply(h.clone().requires_grad_(), eb)
        Xr = enclayout.expand_rows_host(h.cpu(), ref)
        assert torch.equal(X.detach().cpu(), Xr)
        dX = torch.randn(eb.pb.n_rows, 32, device=dev)
        hg = h.clone().requires_grad_()
        enclayout.EncRowsFn.apply(hg, eb).backward(dX)
        hr = h.cpu().double().requires_grad_()
        (enclayout.expand_rows_host(hr, ref) * dX.cpu().double()).sum().backward()
        close(hg.grad, hr.grad, "reduce", tol=1e-6)


@pytest.mark.parametrize("encoder,n_layers,attn,d", [("ggnn", 3, "nie", 128), ("ggnn", 2, "nie", 64), ("relgcn", 2, "nie", 128),
                                                     ("relgcn", 2, "pool", 64), ("ggnn", 2, None, 64), ("ggnn", 2, "global", 64),
                                                     ("ggnn", 2, None, 32)])
@pytest.mark.parametrize("dedup,n_cu", [(False, 256), (True, 256), (False, 5)])
def test_encoder_layout_matches_oracle(encoder, n_layers, attn, d, dedup, n_cu):
    """Small batch with heavy repetition; n_cu = 5 forces tiles of 2..4 blocks shared by several molecules, n_cu = 256 gives
    nearly every molecule a tile of its own height (1..2 blocks): the short-tile paths of the fused kernels at d = 64 and 128
    (d = 32: the row-wise operators on the same layout)."""
    from bmp import enclayout, packed, synth
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import grad_dict, load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(9, seed=31, n_lo=2, n_hi=60, n_mean=16)
    ms = packed.MolStore(store)
    i1 = np.array([0, 1, 2, 0, 3, 3, 8, 1, 0, 5, 5, 2, 6]); i2 = np.array([1, 0, 0, 0, 4, 3, 1, 8, 7, 5, 2, 2, 6])
    B = len(i1)
    lab = (np.arange(B).reshape(-1, 1) % 2).astype(np.int32)
    p = O.make_pair_params(777, encoder=encoder, hidden_dim=d, out_dim=d, n_layers=n_layers, attn=attn, head=8,
                           dtype=torch.float64, bias_scale=0.05)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    yo, g1o, g2o = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), encoder=encoder, n_layers=n_layers, attn=attn)
    O.sigmoid_cross_entropy(yo, T(lab)).backward()
    model = build_pair_predictor(hidden_dim=d, out_dim=d, n_layers=n_layers, attn=attn, head=8, encoder=encoder).to(dev)
    load_param_dict(model, p)
    ds = packed.DeviceMolStore(ms, dev)
    eb, t = enclayout.encode_from_store_device(ds, [i1, i2], labels=lab, dedup=dedup, n_cu=n_cu)
    assert eb.n_encoded == (9 if dedup else 2 * B)
    nb = eb.pb_enc.mt_nblk.cpu().numpy()
    assert nb.min() >= 1 and nb.max() <= 4 and (n_cu != 5 or nb.max() >= 2)
    y = model(eb)
    model.loss(y, t).backward()
    close(y, yo, "logits"); close(model.g1, g1o, "g1"); close(model.g2, g2o, "g2")
    for name, gr in grad_dict(model).items():
        if p[name].grad is not None:
            floor = p["attn/energy_layer/V1"].grad.abs().max().item() if name == "attn/energy_layer/b" else 1e-6
            if name.startswith("attn/energy_layers"):
                floor = 1e-4 * max(v.grad.abs().max().item() for k, v in p.items() if k.startswith("attn/") and v.grad is not None)
            close(gr, p[name].grad, f"grad {name}", floor=floor)


def test_planned_step_in_the_encoder_layout_equals_the_per_instance_step_at_full_size():
    """1024 pairs of the 544-drug store through the planned path three ways -- per-instance layout, encoder layout, encoder
    layout with de-duplication (about 530 distinct molecules among 2048 instances): logits and the flat gradient agree to
    float32 summation order (1e-5 of the tensor's max-abs); each form is bitwise reproducible; the tile table schedules the
    batch as 4 + 3 blocks per CU."""
    from bmp import enclayout, packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store()
    ms = packed.MolStore(store)
    i1, i2, lab = synth.make_pairs()
    i1, i2, lab = i1[:1024], i2[:1024], lab[:1024].reshape(-1, 1)
    p = O.make_pair_params(777, hidden_dim=128, out_dim=128, n_layers=4, attn="nie", head=8, dtype=torch.float32, bias_scale=0.05)
    model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie", head=8).to(dev)
    load_param_dict(model, p)
    opt = FlatAdam(model, alpha=0.0)
    ds = packed.DeviceMolStore(ms, dev)
    pb, t = packed.pack_from_store_device(ds, [i1, i2], labels=lab)
    eb, t1 = enclayout.encode_from_store_device(ds, [i1, i2], labels=lab)
    ed, t2 = enclayout.encode_from_store_device(ds, [i1, i2], labels=lab, dedup=True)
    assert eb.budget == 7 and eb.pb_enc.n_rows <= 57344 + 128 and eb.pb_enc.n_rows < pb.n_rows
    nb = eb.pb_enc.mt_nblk.cpu().numpy()
    assert (np.diff(nb[:-3]) <= 0).all() and set(nb.tolist()) <= {1, 2, 3, 4}           # tallest tiles first
    assert 480 <= ed.n_encoded <= 544 and ed.pb_enc.n_rows < pb.n_rows / 3

    def step(batch, tt):
        y = opt.functional_forward(batch)
        model.loss(y, tt).backward()
        opt.collect_grads()
        torch.cuda.synchronize()
        return y.detach().clone(), opt.grad.clone()

    y_i, g_i = step(pb, t)
    for name, batch, tt in (("encoder layout", eb, t1), ("de-duplicated", ed, t2)):
        y_a, g_a = step(batch, tt)
        y_b, g_b = step(batch, tt)
        assert torch.equal(y_a, y_b) and torch.equal(g_a, g_b), name
        close(y_a, y_i.double(), f"logits {name} vs per-instance", tol=1e-5)
        off = 0
        for pname, shp in zip(opt.names, opt.shapes):
            n = int(np.prod(shp))
            if not pname.startswith(("graph_conv.i_layers", "graph_conv.j_layers")):      # (unused by the fine family: zero both ways)
                close(g_a[off:off + n], g_i[off:off + n].double(), f"grad {pname} {name} vs per-instance", tol=1e-5,
                      floor=1e-3 * g_i.abs().max().item())
            off += n

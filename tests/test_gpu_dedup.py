"""De-duplicated encoding (bmp/dedup.py; SURVEY.md 8(d) "de-duplication caveat"): every distinct molecule of a pair batch
encoded once, the co-attention on the per-instance layout.  Same mathematics as the per-instance path -- checked against the
dense oracle on a small batch with heavy repetition, and against the per-instance planned path at the headline size."""
import numpy as np
import pytest
import torch

from parity_util import close

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.mark.parametrize("encoder,n_layers,attn", [("ggnn", 3, "nie"), ("relgcn", 2, "nie"), ("ggnn", 2, "pool")])
def test_dedup_matches_oracle_on_a_batch_with_repeated_molecules(encoder, n_layers, attn):
    from bmp import packed, synth
    from bmp.dedup import dedup_from_store_device
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import grad_dict, load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    store = synth.make_store(9, seed=31, n_lo=2, n_hi=40, n_mean=14)
    ms = packed.MolStore(store)
    i1 = np.array([0, 1, 2, 0, 3, 3, 8, 1, 0, 5, 5, 2]); i2 = np.array([1, 0, 0, 0, 4, 3, 1, 8, 7, 5, 2, 2])
    B = len(i1)
    lab = (np.arange(B).reshape(-1, 1) % 2).astype(np.int32)
    p = O.make_pair_params(777, encoder=encoder, hidden_dim=64, out_dim=64, n_layers=n_layers, attn=attn, head=8,
                           dtype=torch.float64, bias_scale=0.05)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    yo, g1o, g2o = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), encoder=encoder, n_layers=n_layers, attn=attn)
    O.sigmoid_cross_entropy(yo, T(lab)).backward()
    model = build_pair_predictor(hidden_dim=64, out_dim=64, n_layers=n_layers, attn=attn, head=8, encoder=encoder).to(dev)
    load_param_dict(model, p)
    ds = packed.DeviceMolStore(ms, dev)
    dd, t = dedup_from_store_device(ds, [i1, i2], labels=lab)
    assert dd.n_distinct == 8 and dd.pb_u.n_mols == 8 and dd.pb.n_mols == 2 * B
    y = model(dd)
    model.loss(y, t).backward()
    close(y, yo, "logits"); close(model.g1, g1o, "g1"); close(model.g2, g2o, "g2")
    for name, gr in grad_dict(model).items():
        if p[name].grad is not None:
            floor = p["attn/energy_layer/V1"].grad.abs().max().item() if name == "attn/energy_layer/b" else 1e-6
            close(gr, p[name].grad, f"grad {name}", floor=floor)


def test_dedup_needs_a_fine_coattention():
    from bmp import packed, synth
    from bmp.dedup import dedup_from_store_device
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(6, seed=3, n_lo=2, n_hi=12, n_mean=6)
    ds = packed.DeviceMolStore(packed.MolStore(store), dev)
    dd = dedup_from_store_device(ds, [np.array([0, 1, 2]), np.array([3, 3, 0])])
    for attn in (None, "global"):
        model = build_pair_predictor(hidden_dim=16, out_dim=16, n_layers=2, attn=attn).to(dev)
        with pytest.raises(NotImplementedError):
            model(dd)


def test_dedup_planned_step_equals_the_per_instance_step_at_full_size():
    """1024 pairs of the 544-drug store (about 530 distinct molecules among 2048 instances) through the planned path both
    ways: logits and the flat gradient agree to float32 summation order (1e-5 of the tensor's max-abs), and the de-duplicated
    step is bitwise reproducible (the instance sums run in a fixed order)."""
    from bmp import packed, synth
    from bmp.dedup import dedup_from_store_device
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
    dd, t2 = dedup_from_store_device(ds, [i1, i2], labels=lab)
    assert 480 <= dd.n_distinct <= 544 and dd.pb_u.n_rows < pb.n_rows / 3

    def step(batch, tt):
        y = opt.functional_forward(batch)
        model.loss(y, tt).backward()
        opt.collect_grads()
        torch.cuda.synchronize()
        return y.detach().clone(), opt.grad.clone()

    y_i, g_i = step(pb, t)
    y_d, g_d = step(dd, t2)
    y_d2, g_d2 = step(dd, t2)
    assert torch.equal(y_d, y_d2) and torch.equal(g_d, g_d2)
    close(y_d, y_i.double(), "logits dedup vs per-instance", tol=1e-5)
    off = 0
    for name, shp in zip(opt.names, opt.shapes):
        n = int(np.prod(shp))
        if name.startswith(("graph_conv.i_layers", "graph_conv.j_layers")):
            off += n
            continue                # the readout is unused by the fine family: zero gradient both ways
        close(g_d[off:off + n], g_i[off:off + n].double(), f"grad {name} dedup vs per-instance", tol=1e-5,
              floor=1e-3 * g_i.abs().max().item())
        off += n

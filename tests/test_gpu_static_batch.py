"""The fixed-shape batch (bmp.packed.StaticPairBatch: one molecule per tile, every pair in the pair kernels' 128-row class) and
the training step recorded once as a HIP graph on it (bmp.dp.GraphedTrainStep).  The placement changes where a molecule's rows
sit and in which order a few sums run, nothing else: logits, loss and every gradient against the usual packed batch of the same
pairs at 1e-5, the replayed steps against eager steps of a twin model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bmp import synth, packed              # noqa: E402
from test_gpu_ops import dev               # noqa: E402
from parity_util import close              # noqa: E402


def _model(cfg):
    from bmp.predictor import build_pair_predictor
    torch.manual_seed(5)
    return build_pair_predictor(**cfg).to(dev())


CFGS = {
    "c2": dict(hidden_dim=128, out_dim=128, n_layers=4, attn="nie"),
    "d64": dict(hidden_dim=64, out_dim=64, n_layers=2, attn="nie"),
    "ref_ntn": dict(hidden_dim=32, out_dim=16, n_layers=8, attn="nie", weight_tying=False, sim_method="ntn", mlp_hidden=()),
    "c3": dict(hidden_dim=128, out_dim=128, n_layers=3, attn="nie", encoder="relgcn"),
}


@pytest.fixture(scope="module")
def world():
    store = synth.make_store(120, seed=3, n_lo=3, n_hi=120, n_mean=24)      # rows up to 121: the pair kernels' 128-row class too
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, dev())
    rs = np.random.RandomState(8)
    i1, i2 = rs.randint(0, 120, 512), rs.randint(0, 120, 512)
    lab = (rs.uniform(size=(512, 1)) < 0.35).astype(np.int32)
    lab[::17] = -1
    return ds, i1, i2, lab


@pytest.mark.parametrize("name", ["c2", "d64", "ref_ntn", "c3"])
def test_fixed_shape_batch_gives_the_packed_batch_results(world, name):
    from bmp.dp import FlatAdam
    ds, i1, i2, lab = world
    B = 32
    model = _model(CFGS[name])
    opt = FlatAdam(model, alpha=1e-3)
    sb = packed.StaticPairBatch(ds, B)
    for k in (0, 3):
        sl = slice(k * B, (k + 1) * B)
        pb, t = packed.pack_from_store_device(ds, [i1[sl], i2[sl]], labels=lab[sl])
        loss = opt.functional_loss(pb, t=t); loss.backward(); opt.collect_grads()
        y_ref, g_ref, l_ref = model.y.detach().clone(), opt.grad.clone(), loss.detach().clone()
        sb.load([i1[sl], i2[sl]], lab[sl]); sb.reset_derived(); sb.emit()
        loss = opt.functional_loss(sb.pb, t=sb.t); loss.backward(); opt.collect_grads()
        close(model.y.detach(), y_ref, f"static batch {name} batch {k}: logits", tol=1e-5)
        close(loss.detach().reshape(1), l_ref.reshape(1), f"static batch {name} batch {k}: loss", tol=1e-5)
        close(opt.grad, g_ref, f"static batch {name} batch {k}: flat gradient", tol=1e-5)


@pytest.mark.parametrize("name", ["c2", "ref_ntn"])
def test_one_graph_replays_every_batch(world, name):
    from bmp.dp import FlatAdam, GraphedTrainStep
    ds, i1, i2, lab = world
    B, steps = 32, 12
    eager, graphed = _model(CFGS[name]), _model(CFGS[name])
    oe, og = FlatAdam(eager, alpha=1e-3), FlatAdam(graphed, alpha=1e-3)
    assert torch.equal(oe.flat, og.flat)
    sb = packed.StaticPairBatch(ds, B)
    stepper = GraphedTrainStep(graphed, og)
    le, lg = [], []
    for k in range(steps):
        sl = slice(k * B, (k + 1) * B)
        pb, t = packed.pack_from_store_device(ds, [i1[sl], i2[sl]], labels=lab[sl])
        loss = oe.functional_loss(pb, t=t); loss.backward(); oe.collect_grads(); oe.step()
        le.append(float(loss.detach()))
        sb.load([i1[sl], i2[sl]], lab[sl])
        lg.append(float(stepper(sb).detach()))
    assert len(stepper.graphs) == 1 and og.t == oe.t == steps
    close(torch.tensor(lg), torch.tensor(le), f"graph on the static batch {name}: losses of {steps} steps", tol=1e-4)
    close(og.flat, oe.flat, f"graph on the static batch {name}: parameters after {steps} steps", tol=1e-4)
    assert not np.allclose(le[0], le[-1])          # (different batches: the losses move)


@pytest.mark.parametrize("name", ["c2", "ref_ntn"])
def test_recorded_step_matches_the_oracle(name):
    """What `batch32` of bench.py times -- StaticPairBatch.load + ONE replay of the recorded step (emit, forward, loss, backward,
    weight gradients, Adam) -- against the dense float64 oracle directly: loss, the flat gradient (every parameter) at 2e-5 and
    the update of oracle.chainer_adam_step, for the C2 model and for the reference's published model (NTN), 32 pairs with
    ignored labels.  The replay under test is the SECOND use of the recording (another batch went through the arrays first)."""
    from bmp.dp import FlatAdam, GraphedTrainStep
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    from test_gpu_planned_oracle import _oracle_step
    T = torch.from_numpy
    store = synth.make_store(60, seed=23, n_lo=3, n_hi=100, n_mean=22)
    ds = packed.DeviceMolStore(packed.MolStore(store), dev())
    rs = np.random.RandomState(8)
    B = 32
    first = (rs.randint(0, 60, B), rs.randint(0, 60, B), rs.randint(0, 2, (B, 1)).astype(np.int32))
    i1, i2 = rs.randint(0, 60, B), rs.randint(0, 60, B)
    lab = rs.randint(0, 2, (B, 1)).astype(np.int32)
    lab[::9] = -1
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    if name == "c2":
        kw, okw = dict(hidden_dim=128, out_dim=128, n_layers=4, attn="nie", head=8), {}
    else:
        kw = dict(hidden_dim=32, out_dim=16, n_layers=8, weight_tying=False, attn="nie", head=8, sim_method="ntn", mlp_hidden=())
        okw = dict(weight_tying=False, sim_method="ntn", mlp_hidden=0)
    p = O.make_pair_params(777, encoder="ggnn", dtype=torch.float64, bias_scale=0.05, **kw)
    alpha = 1e-2
    y_o, loss_o, g_o, p_o = _oracle_step(p, (a1, j1, a2, j2, lab), "ggnn", kw["n_layers"], "nie", alpha, **okw)

    model = build_pair_predictor(encoder="ggnn", **kw).to(dev())
    load_param_dict(model, p)
    opt = FlatAdam(model, alpha=alpha)
    flat0 = opt.flat.clone()
    sb = packed.StaticPairBatch(ds, B)
    stepper = GraphedTrainStep(model, opt)
    sb.load([first[0], first[1]], first[2])
    stepper(sb)                                           # records, replays once on another batch
    opt.flat.copy_(flat0); opt.m.zero_(); opt.v.zero_(); opt.t = 0      # back to the oracle's starting point
    sb.load([i1, i2], lab)
    loss = stepper(sb)
    close(loss.detach().reshape(1), loss_o.reshape(1), f"recorded step {name}: loss", tol=2e-5)
    off = 0
    for pname, shp in zip(opt.names, opt.shapes):
        n = int(np.prod(shp))
        key = pname.replace(".", "/")
        ref = g_o[key]
        if float(ref.abs().max()) > 0:                    # (readout parameters: the fine family ignores g_1 / g_2)
            close(opt.grad[off:off + n].view(shp), ref, f"recorded step {name}: grad {pname}", tol=2e-5)
        upd_o = p_o[key] - p[key]
        upd = opt.flat[off:off + n].view(shp).double().cpu() - p[key].float().double()
        big = ref.abs() > 1e-3 * ref.abs().max().clamp(min=1e-30)
        if big.any():
            assert (upd - upd_o)[big].abs().max().item() <= 1e-3 * alpha, pname
        off += n

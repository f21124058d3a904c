"""Oracle check of the FULL-SIZE backward, on what bench.py runs.

`F.sigmoid_cross_entropy` ignores labels equal to -1 (train_ddi_modify.py:284-286): the 1024-pair
batch of each single-GPU configuration goes through the PLANNED path with every stream on (FlatAdam.functional_forward +
collect_grads: layout plan, side stream, two forward chains, device collate), with the labels of all but eight chosen pairs
set to -1.  The loss is then the mean over those eight pairs, and its flat gradient must equal the dense float64 oracle's
gradient on the same eight pairs padded to the full batch's A1 / A2 (the other 1016 pairs contribute exact zeros but run
through every kernel, in all 455 tiles and all pair size classes).  The eight are chosen to cover the <= 32 / <= 64 / <= 96
row classes of the pair kernels and the largest molecule of each side.

C2: GGNN 4-step d=128 tied + Nie (models/ggnn.py:584-654, nie_coattention.py:335-396);  C3: RelGCN 3x128 + Nie
(models/relgcn.py:61-73);  C4: GGNN d=256 + MLP(37), multi-hot labels, the unfused planned operators
(train_ggnn_hole_multi_class_x37.py:71-91).  Tolerance: 2e-5 of the tensor's max-abs (the north star asks 1e-4; achieved: 6e-7 .. 4e-6);
the achieved error is printed and logged (tests/parity_util.py).  Parity unpinned: the oracle is a restatement, SURVEY.md 8(c)."""
import numpy as np
import pytest
import torch

from parity_util import close

pytestmark = pytest.mark.gpu
T = torch.from_numpy
B = 1024
TOL = 2e-5          # achieved in round 3: 1.4e-6 (C2), 4.1e-6 (C3), 6.2e-7 (C4) -- five times tighter than the north star's 1e-4


def _pick_pairs(n1, n2, k=8):
    """Indices of k pairs: at least two of every row class ceil(max(n1, n2) + 1 over 32) present, plus the pairs holding
    the largest molecule of each side."""
    rows = np.maximum(n1, n2) + 1
    cls = (rows + 31) // 32
    rs = np.random.RandomState(17)
    pick = [int(np.argmax(n1)), int(np.argmax(n2))]
    for c in sorted(set(cls.tolist())):
        cand = np.nonzero(cls == c)[0]
        for j in rs.choice(cand, min(2, len(cand)), replace=False):
            if int(j) not in pick:
                pick.append(int(j))
    while len(pick) < k:
        j = int(rs.randint(len(n1)))
        if j not in pick:
            pick.append(j)
    return np.array(sorted(pick[:max(k, len(pick))]))


def _dense(store, idx, A):
    atoms = np.zeros((len(idx), A), np.int32); adj = np.zeros((len(idx), 4, A, A), np.float32)
    for b, k in enumerate(idx):
        m = store[k]
        atoms[b, :m.n] = m.atoms; adj[b, :, :m.n, :m.n] = m.dense_adj()
    return T(atoms), T(adj)


def _run(config, mode="loss"):
    """mode "loss": the step exactly as bench.py's Env.train_step takes it -- ``opt.functional_loss(pb, t=t)``: the one-launch
    head k_mlp_sce, the loss-gradient factor handed to the pair kernels as a device scalar, the deferred weight-gradient
    launches, all streams on.  "forward+loss": ``functional_forward`` + ``model.loss`` (the separate launches; the control).
    "dedup": the bench's de-duplication leg -- ``enclayout.encode_from_store_device(..., dedup=True)`` through
    ``functional_loss`` -- against the ORACLE (not against the per-instance step)."""
    from bmp import enclayout, packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    if config == "c4":
        store = synth.make_store(1704, seed=2018)
        i1, i2, lab = synth.make_multilabel_pairs()
        kw = dict(encoder="ggnn", hidden_dim=256, out_dim=256, n_layers=4, attn=None, class_num=37)
    else:
        store = synth.make_store()
        i1, i2, lab = synth.make_pairs()
        kw = dict(encoder="relgcn" if config == "c3" else "ggnn", hidden_dim=128, out_dim=128, n_layers=3 if config == "c3" else 4,
                  attn="nie", class_num=1)
    i1, i2 = i1[:B], i2[:B]
    lab = lab[:B].reshape(B, -1).astype(np.int32)
    ms = packed.MolStore(store)
    n1, n2 = ms.n_atoms[i1], ms.n_atoms[i2]
    A1, A2 = int(n1.max()), int(n2.max())
    pick = _pick_pairs(n1, n2)
    masked = np.full_like(lab, -1)
    masked[pick] = lab[pick]
    if config == "c4":
        masked[pick[0], 3] = -1                    # an ignored entry inside a live row as well
    okw = dict(encoder=kw["encoder"], hidden_dim=kw["hidden_dim"], out_dim=kw["out_dim"], n_layers=kw["n_layers"], attn=kw["attn"],
               class_num=kw["class_num"], dtype=torch.float64, bias_scale=0.05)
    if kw["attn"]:
        okw["head"] = 8
    p = O.make_pair_params(777, **okw)

    # ---- oracle: the eight pairs alone, padded as the full batch pads them ----
    po = {k: v.clone().requires_grad_() for k, v in p.items()}
    a1, j1 = _dense(store, i1[pick], A1)
    a2, j2 = _dense(store, i2[pick], A2)
    yo, _, _ = O.pair_forward(po, a1, j1.double(), a2, j2.double(), encoder=kw["encoder"], n_layers=kw["n_layers"], attn=kw["attn"])
    lo = O.sigmoid_cross_entropy(yo, T(masked[pick]))
    names = sorted(po)
    go = torch.autograd.grad(lo, [po[n] for n in names], allow_unused=True)
    go = {n: (g if g is not None else torch.zeros_like(po[n])) for n, g in zip(names, go)}

    # ---- the bench's path: device collate, planned forward, streams on ----
    bkw = dict(hidden_dim=kw["hidden_dim"], out_dim=kw["out_dim"], n_layers=kw["n_layers"], attn=kw["attn"], class_num=kw["class_num"],
               encoder=kw["encoder"])
    if kw["attn"]:
        bkw["head"] = 8
    model = build_pair_predictor(**bkw).to(dev)
    load_param_dict(model, p)
    opt = FlatAdam(model, alpha=1e-3)
    ds = packed.DeviceMolStore(ms, dev)
    if mode == "dedup":
        pb, t = enclayout.encode_from_store_device(ds, [i1, i2], labels=masked, dedup=True)
        assert pb.n_encoded < 600 and pb.pb_enc.n_rows < pb.pb.n_rows / 3          # ~530 distinct molecules of 2048 instances
    else:
        pb, t = packed.pack_from_store_device(ds, [i1, i2], labels=masked)
    tag = f"{config}/{mode}"
    for rep in range(2):                                    # twice: nothing stale may survive from the step before
        if mode == "forward+loss":
            y = opt.functional_forward(pb)
            loss = model.loss(y, t)
        else:
            loss = opt.functional_loss(pb, t=t)             # bench.py Env.train_step
            y = model.y
        assert opt.plan is not None and "graph_conv." in opt.plan.P
        assert opt.plan.side is not None and opt.plan.split is not None, "the timed configuration runs with its streams on"
        loss.backward()
        opt.collect_grads()
        torch.cuda.synchronize()
        close(y[T(pick).to(dev)], yo, f"{tag} logits of the live pairs (rep {rep})", tol=TOL)
        close(loss, lo, f"{tag} loss (rep {rep})", tol=TOL)
        off, worst = 0, 0.0
        for name, shp in zip(opt.names, opt.shapes):
            n = int(np.prod(shp))
            worst = max(worst, close(opt.grad[off:off + n].view(shp), go[name.replace(".", "/")], f"{tag} grad {name} (rep {rep})",
                                     tol=TOL))
            off += n
        assert opt.plan.state.get("head_gscale") is None, "the loss-gradient factor was handed over and consumed"
    assert getattr(pb, "pb", pb).n_tiles > 256 and len(pick) >= 8
    return worst


# ---- the step as bench.py times it: opt.functional_loss (one-launch head + gscale into the pair kernels) ----
def test_c2_timed_step_matches_oracle_on_live_pairs():
    _run("c2", "loss")


def test_c3_timed_step_matches_oracle_on_live_pairs():
    _run("c3", "loss")


def test_c4_timed_step_matches_oracle_on_live_pairs():
    _run("c4", "loss")


# ---- control: the separate launches (functional_forward + model.loss), round 3's form of this test ----
def test_c2_forward_plus_loss_matches_oracle_on_live_pairs():
    _run("c2", "forward+loss")


def test_c3_forward_plus_loss_matches_oracle_on_live_pairs():
    _run("c3", "forward+loss")


# ---- the bench's `dedup` leg (every distinct molecule encoded once) against the oracle at 1024 pairs ----
def test_c2_dedup_step_matches_oracle_on_live_pairs():
    _run("c2", "dedup")


def test_c3_dedup_step_matches_oracle_on_live_pairs():
    _run("c3", "dedup")

"""Trainer / evaluator pieces (bmp/trainer.py): metrics against sklearn (the library the reference's evaluators call,
training/extensions/*.py), schedule and stop triggers, pair augmentation; a short fit() on the GPU."""
import numpy as np
import pytest
import torch

from bmp import trainer as T


@pytest.mark.parametrize("seed,ties", [(0, False), (1, True), (2, True)])
def test_metrics_match_sklearn(seed, ties):
    from sklearn import metrics
    rs = np.random.RandomState(seed)
    n, C = 400, 3
    t = (rs.rand(n, C) < 0.3).astype(np.int32)
    prob = rs.rand(n, C)
    if ties:
        prob = np.round(prob, 1 + seed)          # many equal scores: tie handling of the ranks / thresholds
    got = T.classification_metrics(prob, t, ignore_label=None)
    roc = np.mean([metrics.roc_auc_score(t[:, c], prob[:, c]) for c in range(C)])
    prc = []
    for c in range(C):
        p, r, _ = metrics.precision_recall_curve(t[:, c], prob[:, c], pos_label=1)
        prc.append(metrics.auc(r, p))
    acc = np.mean([metrics.accuracy_score(t[:, c], np.round(prob[:, c])) for c in range(C)])
    f1 = np.mean([metrics.f1_score(t[:, c], np.round(prob[:, c]).astype(int), pos_label=1) for c in range(C)])
    assert abs(got["roc_auc"] - roc) < 1e-12
    assert abs(got["prc_auc"] - np.mean(prc)) < 1e-12
    assert abs(got["accuracy"] - acc) < 1e-12 and abs(got["f1"] - f1) < 1e-12


def test_ignore_label_rows_are_left_out():
    t = np.array([[1], [0], [-1], [1], [0]])
    prob = np.array([[0.9], [0.2], [0.99], [0.4], [0.6]])
    a = T.classification_metrics(prob, t, ignore_label=-1)
    b = T.classification_metrics(np.delete(prob, 2, 0), np.delete(t, 2, 0), ignore_label=None)
    assert a == b and a["roc_auc"] == 0.75


def test_shift_and_early_stopping_and_augment():
    class Opt:
        alpha = 1e-3
    o = Opt()
    sh = T.ExponentialShift(o, 0.5, T.SHIFT_SCHEDULES[2])
    lrs = [sh(e) for e in range(1, 12)]
    assert lrs[3] == 1e-3 and lrs[4] == 5e-4 and lrs[9] == 2.5e-4
    st = T.EarlyStopping(patients=2, max_epoch=100)
    vals = [1.0, 0.9, 0.95, 0.91, 0.8]
    assert [st(e + 1, {"validation/main/loss": v}) for e, v in enumerate(vals)] == [False, False, False, True, False]
    i1, i2, lab = T.augment_pairs(np.array([0, 1]), np.array([2, 3]), np.array([1, 0]))
    assert i1.tolist() == [0, 1, 2, 3] and i2.tolist() == [2, 3, 0, 1] and lab.tolist() == [1, 0, 1, 0]


@pytest.mark.parametrize("n,B,world", [(100, 16, 1), (100, 16, 4), (67, 8, 3), (9, 4, 4), (3, 4, 4)])
def test_pair_batches_share_every_global_batch_over_the_ranks(n, B, world):
    i1 = np.arange(n); lab = np.zeros((n, 1), dtype=np.int32)
    its = [T.PairBatches(None, i1, i1, lab, B, shuffle=True, seed=5, rank=r, world=world) for r in range(world)]
    per_rank = [list(it.selections()) for it in its]
    assert len({len(p) for p in per_rank}) == 1 and len(per_rank[0]) == len(its[0])        # every rank steps equally often
    seen = np.concatenate([s for p in per_rank for s in p]) if per_rank[0] else np.zeros(0, dtype=np.int64)
    assert len(set(seen.tolist())) == len(seen) and n - len(seen) < world                  # disjoint, all but < world pairs
    assert all(len(s) >= 1 for p in per_rank for s in p)
    for k in range(len(per_rank[0])):                                                      # a global batch = consecutive order
        got = np.concatenate([per_rank[r][k] for r in range(world)])
        want = np.random.RandomState(5).permutation(n)[k * B * world:(k + 1) * B * world]
        assert got.tolist() == want.tolist()
    second = list(its[0].selections())                                                     # a new order on the next pass
    assert n < 8 or any(a.tolist() != b.tolist() for a, b in zip(per_rank[0], second))
    with pytest.raises(ValueError):
        T.PairBatches(None, i1, i1, lab, B, dedup=True)


@pytest.mark.gpu
def test_fit_from_the_store_in_either_layout_gives_the_same_run():
    """PairBatches in the per-instance layout, in the encoder layout, with each distinct molecule encoded once, and as replays
    of one recorded HIP graph on a fixed-shape batch (layout="static"; the last batch of a pass is short and runs launch by
    launch in between): the same losses and validation metrics epoch by epoch (different summation orders only)."""
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(40, seed=3, n_lo=4, n_hi=40, n_mean=16)
    ds = packed.DeviceMolStore(packed.MolStore(store), dev)
    rs = np.random.RandomState(1)
    i1, i2 = rs.randint(0, 40, 320), rs.randint(0, 40, 320)
    nat = np.array([m.n for m in store])
    lab = ((nat[i1] + nat[i2]) % 2).astype(np.int32).reshape(-1, 1)
    runs = {}
    for name, kw in (("instance", {}), ("encoder", dict(layout="encoder")), ("dedup", dict(layout="encoder", dedup=True)),
                     ("static", dict(layout="static"))):
        torch.manual_seed(0)
        model = build_pair_predictor(hidden_dim=64, out_dim=32, n_layers=2, attn="nie", head=4).to(dev)
        opt = FlatAdam(model, alpha=2e-3)
        tr = T.PairBatches(ds, i1[:256], i2[:256], lab[:256], 48, shuffle=True, seed=2, **kw)
        va = T.PairBatches(ds, i1[256:], i2[256:], lab[256:], 48, **kw)
        assert len(tr) == 6 and len(va) == 2
        runs[name] = T.fit(model, opt, tr, va, epochs=3, eval_train=True)
    for name in ("encoder", "dedup", "static"):
        for a, b in zip(runs["instance"], runs[name]):
            for key in ("main/loss", "validation/main/loss"):
                assert abs(a[key] - b[key]) <= 2e-4 * abs(a[key]), (name, key, a[key], b[key])
            for key in ("val_roc/main/roc_auc", "train_roc/main/roc_auc"):
                assert abs(a[key] - b[key]) <= 5e-3, (name, key, a[key], b[key])
    assert runs["instance"][-1]["main/loss"] < runs["instance"][0]["main/loss"]


@pytest.mark.gpu
def test_fit_lowers_the_loss_and_reports_reference_columns():
    from bmp import packed, synth
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(60, seed=9, n_lo=4, n_hi=30, n_mean=12)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(0)
    i1, i2 = rs.randint(0, 60, 256), rs.randint(0, 60, 256)
    lab = ((np.array([m.n for m in store])[i1] + np.array([m.n for m in store])[i2]) % 2).astype(np.int32)   # learnable
    i1, i2, lab = T.augment_pairs(i1, i2, lab)
    def batches(lo, hi, B=64):
        return [(packed.pack_from_store(ms, [i1[k:k + B], i2[k:k + B]], device=dev),
                 torch.from_numpy(lab[k:k + B].reshape(-1, 1)).to(dev)) for k in range(lo, hi, B)]
    tr, va = batches(0, 384), batches(384, 512)
    torch.manual_seed(0)
    model = build_pair_predictor(hidden_dim=64, out_dim=32, n_layers=2, attn="nie", head=4).to(dev)
    opt = FlatAdam(model, alpha=3e-3)
    logs = T.fit(model, opt, tr, va, epochs=6, shift=T.ExponentialShift(opt, 0.5, (4,)), stopper=T.EarlyStopping(patients=50))
    assert set(logs[0]) >= {"epoch", "main/loss", "validation/main/loss", "val_acc/main/accuracy", "val_roc/main/roc_auc",
                            "val_prc/main/prc_auc", "val_f/main/f1", "lr", "elapsed_time"}
    assert logs[-1]["main/loss"] < logs[0]["main/loss"]
    assert logs[2]["lr"] == 3e-3 and logs[3]["lr"] == 1.5e-3
    assert all(np.isfinite(v) for l in logs for v in l.values())

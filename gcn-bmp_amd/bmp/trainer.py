"""Minimal trainer / evaluator counterpart of what the reference's scripts wire up around the
hot path with chainer.training (train_binary.py:530-665, train_ddi_modify.py:280-380):

* ``evaluate``: BatchEvaluator semantics (training/extensions/batch_evaluator.py:49-100): logits of the whole
  iterator under no-backprop, sigmoid, then ROC-AUC / PRC-AUC / accuracy / F1 on the host with the definitions of
  training/extensions/{roc_auc,prc_auc,acc,f1}_evaluator.py (labels equal to ``ignore_label`` are left out);
* ``ExponentialShift`` of Adam's alpha at manually scheduled epochs (train_binary.py:636-646);
* ``EarlyStopping`` = triggers.EarlyStoppingTrigger(monitor='validation/main/loss', patients=50,
  max_trigger=(500, 'epoch')) (train_binary.py:558);
* ``augment_pairs``: the pair swap of augment_dataset (train_binary.py:285-294) on index arrays;
* ``PairBatches``: SerialIterator + the converter concat_mols (train_binary.py:520-528, 551) over a pair list and a drug store
  resident in HBM: every batch collated on the device, in the per-instance layout or in the encoder layout (optionally with
  every distinct molecule of the batch encoded once), this rank's share of each global batch;
* ``fit``: StandardUpdater + the extensions above as one loop over packed batches, log entries named as the
  reference's PrintReport columns (train_binary.py:650-658).

Metrics are plain numpy (no sklearn at run time); tests/test_trainer.py checks them against sklearn.
"""
from __future__ import annotations

import math
import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

SHIFT_SCHEDULES = {1: (10, 20, 30, 40, 50, 60), 2: (5, 10, 15, 20, 25, 30),
                   3: (5, 10, 15, 20, 25, 30, 40, 50, 60, 70)}          # train_binary.py:638-646


# ---- metrics (one column = one class; averaged over columns like the reference's evaluators) -------------------
def _valid(y: np.ndarray, t: np.ndarray, ignore_label):
    m = np.ones(len(t), dtype=bool) if ignore_label is None else (t != ignore_label)
    return y[m], t[m]


def roc_auc_binary(score: np.ndarray, t: np.ndarray) -> float:
    """Area under the ROC curve = Mann-Whitney statistic with average ranks for ties (sklearn.metrics.roc_auc_score)."""
    pos = t == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(score, kind="mergesort")
    s = score[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):                      # average ranks over runs of equal scores
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty(len(s), dtype=np.float64)
    r[order] = ranks
    return float((r[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def prc_auc_binary(score: np.ndarray, t: np.ndarray) -> float:
    """metrics.auc(recall, precision) over metrics.precision_recall_curve (prc_auc_evaluator.py:111-117): one point
    per distinct score threshold, trapezoid rule, plus the (recall 0, precision 1) end point."""
    n_pos = int((t == 1).sum())
    if n_pos == 0:
        return float("nan")
    order = np.argsort(-score, kind="mergesort")
    s, tt = score[order], (t[order] == 1).astype(np.float64)
    distinct = np.nonzero(np.diff(s))[0]
    idx = np.r_[distinct, len(s) - 1]
    tps = np.cumsum(tt)[idx]
    fps = 1 + idx - tps
    precision = tps / (tps + fps)
    recall = tps / n_pos
    last = int(np.searchsorted(tps, tps[-1]))          # sklearn stops at the first threshold with full recall
    precision = np.r_[precision[:last + 1][::-1], 1.0]
    recall = np.r_[recall[:last + 1][::-1], 0.0]
    return float(-np.trapezoid(precision, recall))


def f1_binary(pred: np.ndarray, t: np.ndarray) -> float:
    tp = float(((pred == 1) & (t == 1)).sum())
    fp = float(((pred == 1) & (t != 1)).sum())
    fn = float(((pred != 1) & (t == 1)).sum())
    return 0.0 if 2 * tp + fp + fn == 0 else 2 * tp / (2 * tp + fp + fn)


def classification_metrics(prob: np.ndarray, t: np.ndarray, ignore_label: Optional[int] = -1) -> Dict[str, float]:
    """prob = sigmoid(logits) (B, C), t (B, C) in {0, 1, ignore_label}.  Means over the classes."""
    prob = np.asarray(prob, dtype=np.float64).reshape(len(prob), -1)
    t = np.asarray(t).reshape(len(t), -1)
    roc, prc, acc, f1 = [], [], [], []
    for c in range(prob.shape[1]):
        y, tc = _valid(prob[:, c], t[:, c], ignore_label)
        pred = np.round(y)                                   # acc_evaluator.py:66
        roc.append(roc_auc_binary(y, tc)); prc.append(prc_auc_binary(y, tc))
        acc.append(float((pred == tc).mean()) if len(tc) else float("nan")); f1.append(f1_binary(pred, tc))
    return {"roc_auc": float(np.mean(roc)), "prc_auc": float(np.mean(prc)), "accuracy": float(np.mean(acc)),
            "f1": float(np.mean(f1))}


# ---- evaluation / schedule / stopping -------------------------------------------------------------------------------
def evaluate(predictor, batches: Iterable, lossfun: Optional[Callable] = None, ignore_label: int = -1, opt=None) -> Dict[str, float]:
    """``batches`` yields (inputs, labels) with inputs a packed batch or the reference's tuple of four arrays.  With ``opt``
    (bmp.dp.FlatAdam) the logits come from ``opt.functional_predict``: the planned forward-only path on the flat buffer."""
    ys, ts, losses, n = [], [], 0.0, 0
    was_training = predictor.training
    predictor.eval()
    with torch.no_grad():
        for inputs, t in batches:
            if callable(getattr(inputs, "emit", None)):          # a fixed-shape batch (PairBatches(layout="static")): write its arrays
                inputs.reset_derived(); inputs.emit()
                inputs = inputs.pb
            if opt is not None:
                y = opt.functional_predict(*inputs)[0] if isinstance(inputs, (tuple, list)) else opt.functional_predict(inputs)[0]
            else:
                y = predictor(*inputs) if isinstance(inputs, (tuple, list)) else predictor(inputs)
            if lossfun is not None:
                losses += float(lossfun(y, t)) * len(t); n += len(t)
            ys.append(torch.sigmoid(y).float().cpu().numpy()); ts.append(torch.as_tensor(t).cpu().numpy())
    predictor.train(was_training)
    out = classification_metrics(np.concatenate(ys), np.concatenate(ts), ignore_label)
    if lossfun is not None:
        out["loss"] = losses / max(n, 1)
    return out


class ExponentialShift:
    """extensions.ExponentialShift('alpha', rate) fired by a ManualScheduleTrigger: alpha *= rate at the listed epochs."""

    def __init__(self, optimizer, rate: float, epochs: Sequence[int], attr: str = "alpha"):
        self.opt, self.rate, self.epochs, self.attr = optimizer, rate, set(epochs), attr

    def __call__(self, epoch: int) -> float:
        if epoch in self.epochs:
            setattr(self.opt, self.attr, getattr(self.opt, self.attr) * self.rate)
        return getattr(self.opt, self.attr)


class EarlyStopping:
    """chainer.training.triggers.EarlyStoppingTrigger in 'min' mode: stop after ``patients`` checks without a new best
    value of the monitored entry, or at ``max_epoch``."""

    def __init__(self, monitor: str = "validation/main/loss", patients: int = 50, max_epoch: int = 500):
        self.monitor, self.patients, self.max_epoch = monitor, patients, max_epoch
        self.best, self.count = math.inf, 0

    def __call__(self, epoch: int, log: Dict[str, float]) -> bool:
        if epoch >= self.max_epoch:
            return True
        v = log.get(self.monitor)
        if v is None:
            return False
        if v < self.best:
            self.best, self.count = v, 0
        else:
            self.count += 1
        return self.count >= self.patients


def augment_pairs(idx1: np.ndarray, idx2: np.ndarray, label: np.ndarray):
    """augment_dataset (train_binary.py:285-294): every pair once in each order, labels repeated."""
    return np.concatenate((idx1, idx2)), np.concatenate((idx2, idx1)), np.concatenate((label, label))


class PairBatches:
    """The reference's SerialIterator(dataset, batchsize, shuffle=...) + converter concat_mols for one rank: an iterable of
    (batch, labels on the device), re-collated from the store in HBM on every pass (a new order per pass when ``shuffle``,
    like SerialIterator's per-epoch permutation; the last batch of a pass is the short remainder, as with repeat=False).

    ``layout`` "instance": bmp.packed.pack_from_store_device.  "encoder": bmp.enclayout.encode_from_store_device (encoder
    tiles balanced over the CUs); with ``dedup`` every distinct molecule of this rank's share is encoded once and its atom
    states are copied to the instances -- the same logits and gradients, fewer encoder rows.  "static": every full batch is
    loaded into ONE fixed-shape batch (bmp.packed.StaticPairBatch: the same object is yielded every time, its contents change),
    on which ``fit`` replays one recorded HIP graph per step -- the way to run small batches such as the reference's default of
    32 pairs (train_ddi_modify.py:196), where a step is ~55 launches of one workgroup round each; the short remainder of a pass
    comes as a usual packed batch.  With ``world`` > 1 a global
    batch is ``batch_size * world`` pairs and this rank takes its contiguous share (a remainder smaller than ``world`` is
    left out, so that no rank steps on an empty batch)."""

    def __init__(self, dstore, idx1: np.ndarray, idx2: np.ndarray, labels: np.ndarray, batch_size: int, shuffle: bool = False,
                 seed: int = 0, layout: str = "instance", dedup: bool = False, rank: int = 0, world: int = 1):
        if layout not in ("instance", "encoder", "static"):
            raise ValueError(f"layout {layout!r}: 'instance', 'encoder' or 'static'")
        if dedup and layout != "encoder":
            raise ValueError("dedup needs layout='encoder'")
        if not (len(idx1) == len(idx2) == len(labels)):
            raise ValueError("idx1, idx2 and labels must have one entry per pair")
        self.dstore, self.i1, self.i2, self.lab = dstore, np.asarray(idx1), np.asarray(idx2), np.asarray(labels)
        self.B, self.shuffle, self.seed, self.layout, self.dedup = int(batch_size), shuffle, seed, layout, dedup
        self.rank, self.world, self.epoch = rank, world, 0

    def __len__(self) -> int:
        g = self.B * self.world
        full, rem = divmod(len(self.lab), g)
        return full + (1 if rem >= self.world else 0)

    def selections(self):
        """This pass's pair indices, one array per batch of this rank (advances the pass counter)."""
        n, g = len(self.lab), self.B * self.world
        order = np.random.RandomState(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        self.epoch += 1
        for lo in range(0, n, g):
            sel = order[lo:lo + g]
            if len(sel) < self.world:
                break
            q, r = divmod(len(sel), self.world)          # ranks 0..r-1 take one pair more: no rank is left without pairs
            a = self.rank * q + min(self.rank, r)
            yield sel[a:a + q + (1 if self.rank < r else 0)]

    def __iter__(self):
        from . import enclayout, packed
        for sel in self.selections():
            sides, lab = [self.i1[sel], self.i2[sel]], self.lab[sel]
            if self.layout == "static" and len(sel) == self.B:
                if getattr(self, "_static", None) is None:
                    self._static = packed.StaticPairBatch(self.dstore, self.B, label_cols=int(np.asarray(lab).reshape(len(sel), -1).shape[1]))
                self._static.load(sides, lab)
                yield self._static, self._static.t
            elif self.layout == "encoder":
                yield enclayout.encode_from_store_device(self.dstore, sides, labels=lab, dedup=self.dedup)
            else:
                yield packed.pack_from_store_device(self.dstore, sides, labels=lab)


def fit(model, opt, train_batches: Sequence, valid_batches: Sequence = (), epochs: int = 1,
        shift: Optional[ExponentialShift] = None, stopper: Optional[EarlyStopping] = None, eval_train: bool = False,
        report: Optional[Callable[[Dict[str, float]], None]] = None) -> List[Dict[str, float]]:
    """One StandardUpdater loop (train_binary.py:551-553) with bmp.dp.FlatAdam ``opt``: per batch forward, loss,
    backward, one gradient all-reduce, Adam; per epoch the evaluators, the alpha shift and the stop trigger."""
    logs: List[Dict[str, float]] = []
    t0 = time.time()
    stepper = None
    for epoch in range(1, epochs + 1):
        tot, n = None, 0
        for pb, t in train_batches:
            if callable(getattr(pb, "emit", None)):                 # a fixed-shape batch: the whole step is one graph replay
                if stepper is None:
                    from .dp import GraphedTrainStep
                    stepper = GraphedTrainStep(model, opt)
                loss = stepper(pb)
                tot = loss.detach() * len(t) if tot is None else tot + loss.detach() * len(t)
                n += len(t)
                continue
            if callable(getattr(model, "forward_loss", None)):      # the reference's Classifier: link predictor + loss together
                loss = opt.functional_loss(*(pb if isinstance(pb, (tuple, list)) else (pb,)), t=t)
            else:
                loss = model.loss(opt.functional_forward(pb), t)
            loss.backward()
            opt.collect_grads()
            opt.all_reduce_grads()
            opt.step()
            tot = loss.detach() * len(t) if tot is None else tot + loss.detach() * len(t)
            n += len(t)
        log = {"epoch": epoch, "main/loss": float(tot) / max(n, 1) if tot is not None else float("nan")}
        if eval_train:
            m = evaluate(model, train_batches, opt=opt)
            log.update({"train_acc/main/accuracy": m["accuracy"], "train_roc/main/roc_auc": m["roc_auc"],
                        "train_prc/main/prc_auc": m["prc_auc"], "train_f/main/f1": m["f1"]})
        if len(valid_batches):
            m = evaluate(model, valid_batches, lossfun=model.loss, opt=opt)
            log.update({"validation/main/loss": m["loss"], "val_acc/main/accuracy": m["accuracy"],
                        "val_roc/main/roc_auc": m["roc_auc"], "val_prc/main/prc_auc": m["prc_auc"], "val_f/main/f1": m["f1"]})
        log["lr"] = shift(epoch) if shift is not None else opt.alpha
        log["elapsed_time"] = time.time() - t0
        logs.append(log)
        if report is not None:
            report(log)
        if stopper is not None and stopper(epoch, log):
            break
    return logs

#!/usr/bin/env python
"""Side measurements of the other BASELINE.json configs (not the bench.py contract line):
  C3  RelGCN 3-layer d=128 + Nie co-attention + MLP, binary DDI set
  C4  GGNN 4-step d=256 + MLP(37 classes), multi-label store (1704 drugs), no co-attention
  C2b the headline model at the reference's default batch of 32 pairs (train_ddi_modify.py:196)
usage: python tools/bench_configs.py [C3] [C4] [C2b]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import numpy as np      # noqa: E402
import torch            # noqa: E402

from bmp import synth, packed                         # noqa: E402
from bmp.predictor import build_pair_predictor        # noqa: E402
from bmp.dp import FlatAdam, GraphedTrainStep         # noqa: E402


def run(name, model, batches, B, steps=20, warmup=5, graphed=False):
    opt = FlatAdam(model, alpha=1e-3)
    stepper = GraphedTrainStep(model, opt) if graphed else None

    def step(i):
        pb, t = batches[i % len(batches)]
        if graphed:                         # one HIP-graph replay per step (recorded at the batch's first use)
            return stepper(pb, t)
        y = opt.functional_forward(pb)
        loss = model.loss(y, t)
        loss.backward()
        opt.collect_grads()
        opt.step()
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": name, "pairs_per_s": round(B * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3),
                      "pairs_per_step": B, "loss": round(float(loss.item()), 5)}))


def main():
    which = sys.argv[1:] or ["C3", "C4", "C2b"]
    dev = torch.device("cuda:0")
    torch.manual_seed(777)
    if "C3" in which or "C2b" in which:
        store = synth.make_store()
        ms = packed.MolStore(store)
        i1, i2, lab = synth.make_pairs()
    if "C3" in which:
        B = 1024
        batches = []
        for k in range(4):
            sl = slice(k * B, (k + 1) * B)
            batches.append((packed.pack_from_store(ms, [i1[sl], i2[sl]], device=dev),
                            torch.from_numpy(lab[sl].reshape(-1, 1)).to(dev)))
        model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=3, attn="nie", encoder="relgcn").to(dev)
        run("C3 RelGCN 3x128 + Nie + MLP, 1024 pairs/step", model, batches, B)
    if "C2b" in which:
        B = 32
        batches = []
        for k in range(16):
            sl = slice(k * B, (k + 1) * B)
            batches.append((packed.pack_from_store(ms, [i1[sl], i2[sl]], device=dev),
                            torch.from_numpy(lab[sl].reshape(-1, 1)).to(dev)))
        model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie").to(dev)
        run("C2 model at the reference's batch of 32 pairs", model, batches, B, steps=50)
        torch.manual_seed(777)
        model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie").to(dev)
        run("C2 model at batch 32, steps replayed as HIP graphs", model, batches, B, steps=200, warmup=len(batches), graphed=True)
    if "C4" in which:
        store = synth.make_store(1704, seed=2018)
        ms = packed.MolStore(store)
        i1, i2, lab = synth.make_multilabel_pairs()
        B = 1024
        batches = []
        for k in range(4):
            sl = slice(k * B, (k + 1) * B)
            batches.append((packed.pack_from_store(ms, [i1[sl], i2[sl]], device=dev), torch.from_numpy(lab[sl]).to(dev)))
        model = build_pair_predictor(hidden_dim=256, out_dim=256, n_layers=4, attn=None, class_num=37).to(dev)
        run("C4 GGNN 4-step d=256 + MLP(37), multi-label, 1024 pairs/step", model, batches, B)


if __name__ == "__main__":
    main()

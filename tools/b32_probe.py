#!/usr/bin/env python
"""The 32-pair training step (the reference's default batch, train_ddi_modify.py:196) under a kernel trace: run as
   rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python tools/b32_probe.py [encoder|instance]
GPU time per step = sum of kernel durations / steps (one dependent chain), against the wall time per step printed here."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gcn-bmp_amd")]
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from bmp import synth, packed, enclayout
from bmp.predictor import build_pair_predictor
from bmp.dp import FlatAdam
layout = sys.argv[1] if len(sys.argv) > 1 else "encoder"
dev = torch.device("cuda:0")
store = synth.make_store(); ms = packed.MolStore(store); ds = packed.DeviceMolStore(ms, dev)
i1, i2, lab = synth.make_pairs(); lab = lab.reshape(-1, 1)
torch.manual_seed(777)
model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie", head=8).to(dev)
opt = FlatAdam(model, alpha=1e-3)
def batch(k):
    sl = slice(32 * k, 32 * k + 32)
    if layout == "encoder":
        return enclayout.encode_from_store_device(ds, [i1[sl], i2[sl]], labels=lab[sl])
    return packed.pack_from_store_device(ds, [i1[sl], i2[sl]], labels=lab[sl])
def step(k):
    pb, t = batch(k)
    y = opt.functional_forward(pb)
    loss = model.loss(y, t)
    loss.backward(); opt.collect_grads(); opt.step()
for k in range(20): step(k)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for k in range(n): step(20 + k)
torch.cuda.synchronize()
print(f"layout {layout}: wall {1e3 * (time.perf_counter() - t0) / n:.3f} ms per step over {n} steps (+20 warm-up)")

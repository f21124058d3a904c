"""Eager vs HIP-graph replay of the same training steps (C2 model, batch 32): the losses of the last steps must agree."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import torch
from bmp import synth, packed
from bmp.predictor import build_pair_predictor
from bmp.dp import FlatAdam, GraphedTrainStep

dev = torch.device("cuda:0")
store = synth.make_store(); ms = packed.MolStore(store); i1, i2, lab = synth.make_pairs()
B = 32
batches = []
for k in range(16):
    sl = slice(k * B, (k + 1) * B)
    batches.append((packed.pack_from_store(ms, [i1[sl], i2[sl]], device=dev), torch.from_numpy(lab[sl].reshape(-1, 1)).to(dev)))
out = {}
for graphed in (False, True):
    torch.manual_seed(777)
    model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie").to(dev)
    opt = FlatAdam(model, alpha=1e-3)
    st = GraphedTrainStep(model, opt) if graphed else None
    losses = []
    for i in range(216):
        pb, t = batches[i % 16]
        if graphed:
            loss = st(pb, t)
        else:
            y = opt.functional_forward(pb); loss = model.loss(y, t); loss.backward(); opt.collect_grads(); opt.step()
        losses.append(float(loss.item()))
    out[graphed] = losses
    print("graphed" if graphed else "eager  ", " ".join(f"{x:.4f}" for x in losses[:3]), "...", " ".join(f"{x:.4f}" for x in losses[-3:]))
print("max |diff| over the run:", max(abs(a - b) for a, b in zip(out[False], out[True])))

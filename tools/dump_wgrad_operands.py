"""Operands of a real weight-gradient launch for tools/wgrad_bf16x3: one training step of config C2 (1024 pairs of the synthetic
binary-DDI set, random-init weights after a few Adam steps), h and gda [N x 7d] of the LAST propagation step's backward.
    python tools/dump_wgrad_operands.py gpurun_out/wgrad_ops.bin       (on the GPU box)
File: int32 N, int32 Nn, float32 X [N x 128], float32 dY [N x Nn]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import __graft_entry__ as ge
ge.build()
from bmp import synth, packed
from bmp import functional as Fn
from bmp.dp import FlatAdam
from bmp.predictor import build_pair_predictor

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "wgrad_ops.bin")
dev = torch.device("cuda:0")
store = synth.make_store(); ms = packed.MolStore(store)
i1, i2, lab = synth.make_pairs()
ds = packed.DeviceMolStore(ms, dev)
torch.manual_seed(777)
model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie", head=8).to(dev)
opt = FlatAdam(model, alpha=1e-3)
grabbed = []
orig = Fn._on_side


def spy(state, keep, fn):
    if len(keep) == 4 and keep[3].dim() == 2 and keep[3].shape[1] == 7 * 128 and grabbed is not None:
        grabbed.append((keep[0].detach().clone(), keep[3].detach().clone()))
    return orig(state, keep, fn)


Fn._on_side = spy
for k in range(6):          # a few steps so that the weights are not exactly the initial draw
    sl = slice(k * 1024, (k + 1) * 1024)
    pb, t = packed.pack_from_store_device(ds, [i1[sl], i2[sl]], labels=lab[sl].reshape(-1, 1))
    grabbed.clear()
    loss = opt.functional_loss(pb, t=t); loss.backward(); opt.collect_grads(); opt.step()
torch.cuda.synchronize()
h, gda = grabbed[0]          # the backward runs the steps last to first: entry 0 is the LAST step (a later GRU call: all 7d columns live)
N, Nn = h.shape[0], gda.shape[1]
with open(out, "wb") as f:
    np.array([N, Nn], np.int32).tofile(f)
    h.cpu().numpy().astype(np.float32).tofile(f)
    gda.cpu().numpy().astype(np.float32).tofile(f)
print(f"wrote {out}: N {N}, Nn {Nn}; |h| max {h.abs().max().item():.3f}, gda abs max {gda.abs().max().item():.3e}, "
      f"median |gda| {gda.abs().median().item():.3e}, zeros {float((gda == 0).float().mean()):.3f}")

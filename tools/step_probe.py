#!/usr/bin/env python
"""Time the fused GGNN step kernels alone (HIP events around every launch, bmp_prof_*):
the GGNN encoder forward + backward of one bench batch, a few repetitions.
usage: python tools/step_probe.py [d] [reps]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import torch            # noqa: E402

from bmp import synth, packed, _lib      # noqa: E402
from bmp.ggnn import GGNN                # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.manual_seed(777)
store = synth.make_store()
ms = packed.MolStore(store)
i1, i2, _ = synth.make_pairs()
pb = packed.pack_from_store(ms, [i1[:1024], i2[:1024]], device=dev)
enc = GGNN(out_dim=d, hidden_dim=d, n_layers=4).to(dev)
L = _lib.lib()


def run():
    g = enc(pb)
    g.sum().backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
out = (ctypes.c_double * 3)()
res = {"tiles": pb.n_tiles, "d": d}
for cls, name in ((5, "fwd"), (6, "bwd")):
    L.bmp_prof_start(cls)
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    n = L.bmp_prof_stop(out)
    res[name + "_us_per_launch"] = round(1e3 * out[0] / max(n, 1), 1)
print(json.dumps(res))

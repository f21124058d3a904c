"""Host cost of one replay of the recorded 32-pair step, with and without the side streams in the recording."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from bmp import synth, packed
from bmp.predictor import build_pair_predictor
from bmp.dp import FlatAdam, GraphedTrainStep
dev = torch.device("cuda:0")
store = synth.make_store(); ms = packed.MolStore(store); ds = packed.DeviceMolStore(ms, dev)
i1, i2, lab = synth.make_pairs(); lab = lab.reshape(-1, 1)
torch.manual_seed(1)
model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, attn="nie").to(dev)
opt = FlatAdam(model, alpha=1e-3)
sb = packed.StaticPairBatch(ds, 32)
st = GraphedTrainStep(model, opt)
def load(i):
    lo = i * 32
    sb.load([i1[lo:lo + 32], i2[lo:lo + 32]], lab[lo:lo + 32])
for i in range(5):
    load(i); st(sb)
torch.cuda.synchronize()
g = st.graphs[next(iter(st.graphs))][0]
n = 300
tl = tr = 0.0
t0 = time.perf_counter()
for i in range(n):
    a = time.perf_counter(); load(i + 5); b = time.perf_counter(); st(sb); c = time.perf_counter()
    tl += b - a; tr += c - b
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ONE_STREAM={os.environ.get('BMP_ONE_STREAM')}: {1e3 * dt / n:.3f} ms per step; host: load {1e6 * tl / n:.0f} us, replay call {1e6 * tr / n:.0f} us")
# GPU time of a replay alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50):
    g.replay(); torch.cuda.synchronize()
print(f"replay + synchronize, one at a time: {1e3 * (time.perf_counter() - t0) / 50:.3f} ms")

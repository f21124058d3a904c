"""Where the host's time goes in one planned training step (cProfile of steps issued into empty queues: nothing blocks).
   python tools/host_profile.py [c2|c3] [pairs]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from bmp import synth, packed
from bmp.predictor import build_pair_predictor
from bmp.dp import FlatAdam
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
store = synth.make_store(); ds = packed.DeviceMolStore(packed.MolStore(store), dev)
i1, i2, lab = synth.make_pairs(); lab = lab.reshape(-1, 1)
torch.manual_seed(1)
kw = dict(hidden_dim=128, out_dim=128, n_layers=4, attn="nie") if cfg == "c2" else dict(hidden_dim=128, out_dim=128, n_layers=3, attn="nie", encoder="relgcn")
model = build_pair_predictor(**kw).to(dev)
opt = FlatAdam(model, alpha=1e-3)
batches = [packed.pack_from_store_device(ds, [i1[k * B:(k + 1) * B], i2[k * B:(k + 1) * B]], labels=lab[k * B:(k + 1) * B]) for k in range(6)]
def step(k):
    pb, t = batches[k % 6]
    loss = opt.functional_loss(pb, t=t); loss.backward(); opt.collect_grads(); opt.step()
for k in range(10): step(k)
torch.cuda.synchronize()
tt = 0.0
for r in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(3): step(k)
    tt += time.perf_counter() - t0
print(f"{cfg} B={B}: host {1e3 * tt / 120:.3f} ms per step")
pr = cProfile.Profile()
for r in range(40):
    torch.cuda.synchronize(); pr.enable()
    for k in range(3): step(k)
    pr.disable()
st = pstats.Stats(pr, stream=sys.stdout); st.sort_stats("tottime").print_stats(32)

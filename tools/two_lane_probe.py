"""Upper bound of running a 1024-pair step as two half-batches on two streams (one lane's latency-bound phases -- co-attention,
readout, MLP, loss -- under the other lane's tile kernels): two independent models, each stepping on 512-pair batches on its
own stream, enqueued alternately by one host thread, against one model on 1024-pair and on 512-pair batches.
   python tools/two_lane_probe.py [c2|c3] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from bmp import synth, packed
from bmp.predictor import build_pair_predictor
from bmp.dp import FlatAdam

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
store = synth.make_store()
ds = packed.DeviceMolStore(packed.MolStore(store), dev)
i1, i2, lab = synth.make_pairs()
lab = lab.reshape(-1, 1)


def batches(B, nb, first=0):
    return [packed.pack_from_store_device(ds, [i1[lo:lo + B], i2[lo:lo + B]], labels=lab[lo:lo + B])
            for lo in range(first, first + B * nb, B)]


def lane():
    torch.manual_seed(777)
    kw = dict(encoder="ggnn", n_layers=4) if cfg == "c2" else dict(encoder="relgcn", n_layers=3)
    m = build_pair_predictor(hidden_dim=128, out_dim=128, attn="nie", head=8, class_num=1, **kw).to(dev)
    return m, FlatAdam(m, alpha=1e-3)


def step(m, o, pb, t):
    y = o.functional_forward(pb)
    loss = m.loss(y, t)
    loss.backward()
    o.collect_grads()
    o.step()
    return loss


def run(name, lanes, bs, B, streams):
    for i in range(6):
        for (m, o), s in zip(lanes, streams):
            with torch.cuda.stream(s):
                step(m, o, *bs[i % len(bs)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for k, ((m, o), s) in enumerate(zip(lanes, streams)):
            with torch.cuda.stream(s):
                step(m, o, *bs[(2 * i + k) % len(bs)])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = steps * len(lanes) * B
    print(f"{name:46s} {n / dt / 1e3:8.1f} k pairs/s   {dt / steps * 1e3:6.3f} ms per round   host {t_host / steps * 1e3:6.3f} ms", flush=True)


b1024 = batches(1024, 16)
b512 = batches(512, 32)
torch.cuda.synchronize()
A, Bm = lane(), lane()
cur = torch.cuda.current_stream()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
run("one lane, 1024 pairs per step", [A], b1024, 1024, [cur])
run("one lane, 512 pairs per step", [A], b512, 512, [cur])
run("two lanes x 512 (caller's stream + one more)", [A, Bm], b512, 512, [cur, sB])
run("two lanes x 512 (two streams of their own)", [A, Bm], b512, 512, [sA, sB])
run("two lanes x 512, both on the caller's stream", [A, Bm], b512, 512, [cur, cur])
run("one lane, 1024 pairs per step (again)", [A], b1024, 1024, [cur])

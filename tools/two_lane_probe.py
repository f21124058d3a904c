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
ONLY = os.environ.get("LANE_ONLY")
if ONLY is None:
  run("one lane, 1024 pairs per step", [A], b1024, 1024, [cur])
  run("one lane, 512 pairs per step", [A], b512, 512, [cur])
  run("two lanes x 512 (caller's stream + one more)", [A, Bm], b512, 512, [cur, sB])
  run("two lanes x 512 (two streams of their own)", [A, Bm], b512, 512, [sA, sB])
  run("two lanes x 512, both on the caller's stream", [A, Bm], b512, 512, [cur, cur])
  run("one lane, 1024 pairs per step (again)", [A], b1024, 1024, [cur])


# ---- the staggered schedule: encoder chains of both halves on the caller's stream, the latency-bound middle (co-attention, MLP,
# loss and their backward) of each half on a second stream, under the other half's encoder kernels ----
from bmp.dp import _Unflatten


def phase1(lane, pb):
    m, o = lane
    leaf = o.flat.detach().requires_grad_()
    o._leaf = leaf
    views = _Unflatten.apply(leaf, o.shapes)
    plan = o._layout_plan()
    plan.prepare(o.flat)
    tape = leaf[:1]
    for prefix, mod in o._plan_sections:
        mod._fast = (plan.P[prefix], plan.G[prefix], plan.state, tape)
    slots = o._param_slots()
    for (reg, key, _orig), k in slots:
        reg[key] = views[k]
    m.graph_conv._readout_off_chain = True
    enc = m._encode(pb, None, None, None)
    m.graph_conv._readout_off_chain = False
    return enc, slots


def phase2(lane, enc, slots, t):
    m, o = lane
    g1, g2, at1, at2, mol0 = enc
    g1, g2 = m.attn(at1, g1, at2, g2, mol0=mol0)
    y = m.mlp(g1, g2)
    for (reg, key, orig), _k in slots:
        reg[key] = orig
    for _p, mod in o._plan_sections:
        mod._fast = None
    return m.loss(y, t)


def staggered(name, lanes, bs, B, s2):
    cur = torch.cuda.current_stream()

    def round_(i):
        (pbA, tA), (pbB, tB) = bs[(2 * i) % len(bs)], bs[(2 * i + 1) % len(bs)]
        eA = phase1(lanes[0], pbA); evA = cur.record_event()
        eB = phase1(lanes[1], pbB); evB = cur.record_event()
        with torch.cuda.stream(s2):
            s2.wait_event(evA)
            lA = phase2(lanes[0], *eA, tA)
            lA.backward()          # (inside the context: the root gradient's fill must not queue behind the other half's encoder)
        with torch.cuda.stream(s2):
            s2.wait_event(evB)
            lB = phase2(lanes[1], *eB, tB)
            lB.backward()
        cur.wait_stream(s2)
        for m, o in lanes:
            o.collect_grads(); o.step()
        return lA, lB
    for i in range(6):
        round_(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        l = round_(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:46s} {steps * 2 * B / dt / 1e3:8.1f} k pairs/s   {dt / steps * 1e3:6.3f} ms per round   host {t_host / steps * 1e3:6.3f} ms"
          f"   losses {float(l[0]):.4f} {float(l[1]):.4f}", flush=True)


staggered("two halves staggered (middle on a 2nd stream)", [A, Bm], b512, 512, sB)
if ONLY is None:
    run("one lane, 1024 pairs per step (again)", [A], b1024, 1024, [cur])

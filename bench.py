#!/usr/bin/env python
"""Headline benchmark: drug-pairs/sec, fwd+bwd(+Adam), binary-DDI GGNN d=128 + Nie co-attention.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one training step over one batch of 1024 drug pairs per GPU (workload C2/C5 of
SURVEY.md 8(d)): both molecules of every pair encoded by the 4-step GGNN, Nie co-attention, MLP,
sigmoid cross entropy, backward, one gradient all-reduce (N > 1), Adam.  Packed batches are
resident in HBM before the timed region (the collate is the reference's host-side
``concat_mols`` step); every molecule INSTANCE is encoded (no per-batch de-duplication).
Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel class
(HIP events, measured in this run) and a CPU baseline (the oracle on the host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402

D, T_STEPS, HEAD, O = 128, 4, 8, 128
PAIRS_PER_GPU = 1024
N_DISTINCT_BATCHES = 8
PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak
PEAK_HBM_GBS = 8000.0


def algorithmic_flops_per_pair(n_atoms_per_pair: float, d=D, T=T_STEPS, o=O, head=HEAD, n1=None, n2=None):
    """fwd per atom-step 26 d^2 (message 8 d^2 + GRU 18 d^2), readout 6 d^2 per atom,
    Nie co-attention, MLP; fwd+bwd = 3 x fwd.  Real atoms only (no pad/dead rows, no folding credit).
    The readout counts ONCE: the fine co-attention ignores g_1 / g_2 (nie_coattention.py:335-370), so the readout
    is computed forward (as the reference does) but no gradient ever reaches it -- there is no readout backward."""
    half = n_atoms_per_pair / 2.0
    ggnn = 26.0 * d * d * n_atoms_per_pair * T
    readout = 6.0 * d * o * n_atoms_per_pair
    co = 2.0 * (half * d * d + half * half * d + n_atoms_per_pair * d * o + 2 * n_atoms_per_pair * head * d)
    mlp = 2.0 * (2 * o * 32 + 32 * 16 + 16)
    return 3.0 * (ggnn + co + mlp) + readout


def algorithmic_bytes_per_pair(n_atoms_per_pair: float, n_edges_per_pair: float, d=D, T=T_STEPS):
    """SURVEY.md 8(d) compulsory-traffic model: 2 n d (20 T + 40) + index bytes."""
    return n_atoms_per_pair * d * (20.0 * T + 40.0) + 16.0 * T * (n_atoms_per_pair / 2 + 1 + n_edges_per_pair / 2) \
        + 4.0 * n_atoms_per_pair


def cpu_baseline(store, idx1, idx2, label, seconds=12.0):
    """The oracle (oracle/ref_cpu.py, dense op-for-op restatement of the reference) timed on the
    host cores: same model, fwd+bwd+Adam, the reference's default batch of 32 pairs
    (train_ddi_modify.py:196), for about `seconds` of wall time."""
    from oracle import ref_cpu as O_
    from bmp import synth
    # the GPU box exposes all host cores but grants a 16-core share per GPU: more threads only thrash
    torch.set_num_threads(min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))
    p = O_.make_pair_params(777, hidden_dim=D, out_dim=O, n_layers=T_STEPS, attn="nie", head=HEAD, dtype=torch.float32,
                            bias_scale=0.0)
    names = sorted(p)
    params = [p[n].requires_grad_() for n in names]
    state = [dict(m=torch.zeros_like(x), v=torch.zeros_like(x)) for x in params]
    B = 32
    done, t_used, step = 0, 0.0, 0
    while True:
        sl = slice(step * B, (step + 1) * B)
        a1, j1 = synth.concat_mols([store[k] for k in idx1[sl]])
        a2, j2 = synth.concat_mols([store[k] for k in idx2[sl]])
        t = torch.from_numpy(label[sl].reshape(-1, 1))
        t0 = time.perf_counter()
        y, _, _ = O_.pair_forward(p, torch.from_numpy(a1), torch.from_numpy(j1), torch.from_numpy(a2),
                                  torch.from_numpy(j2), n_layers=T_STEPS, attn="nie")
        loss = O_.sigmoid_cross_entropy(y, t)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(x) for g, x in zip(grads, params)]
        with torch.no_grad():
            O_.chainer_adam_step(params, grads, state, step + 1)
        dt = time.perf_counter() - t0
        step += 1
        if step > 2:                     # 2 warm-up steps
            done += B
            t_used += dt
        if t_used >= seconds or (step + 1) * B > len(idx1):
            break
    return dict(value=done / t_used, unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{done} pairs ({step - 2} timed steps of batch {B}, dense oracle fwd+bwd+Adam, fp32, "
                       f"{t_used:.1f} s) of the same workload")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-class", type=int, default=0, help="kernel class for the roofline leg (0 = auto)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    # rehearsal switches for a one-GPU box (never set by the driver): all ranks on device 0, collectives over gloo
    one_device = os.environ.get("BMP_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("BMP_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_pg = world == 1 and os.environ.get("BMP_BENCH_FORCE_PG") == "1"      # rehearsal: the RCCL call path on one rank
    if force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from bmp import synth, packed, _lib
    from bmp.predictor import build_pair_predictor
    from bmp.dp import FlatAdam
    L = _lib.lib()

    # ---- workload: the full binary DDI pair list, global batch = 1024 pairs x world -------------------
    store = synth.make_store()
    ms = packed.MolStore(store)
    idx1, idx2, label = synth.make_pairs()
    gb = PAIRS_PER_GPU * world
    batches, n_atoms, n_edges = [], 0, 0
    for k in range(N_DISTINCT_BATCHES):
        lo = k * gb + rank * PAIRS_PER_GPU
        sl = slice(lo, lo + PAIRS_PER_GPU)
        pb = packed.pack_from_store(ms, [idx1[sl], idx2[sl]], device=dev)
        batches.append((pb, torch.from_numpy(label[sl].reshape(-1, 1)).to(dev)))
        n_atoms += pb.n_real_atoms
        n_edges += pb.n_edges
    atoms_per_pair = n_atoms / (N_DISTINCT_BATCHES * PAIRS_PER_GPU)
    edges_per_pair = n_edges / (N_DISTINCT_BATCHES * PAIRS_PER_GPU)

    torch.manual_seed(777)
    model = build_pair_predictor(hidden_dim=D, out_dim=O, n_layers=T_STEPS, attn="nie", head=HEAD).to(dev)
    opt = FlatAdam(model, alpha=1e-3)
    opt.broadcast_parameters(0)
    if force_pg:
        opt.world = 2            # takes the all-reduce branch (a one-rank sum; the folded 1/2 only rescales the updates)

    def step(i, collective=True):
        pb, t = batches[i % N_DISTINCT_BATCHES]
        y = opt.functional_forward(pb)          # parameters = views of the flat buffer, ONE gradient tensor
        loss = model.loss(y, t)
        loss.backward()
        opt.collect_grads()
        if collective:                          # the roofline leg below runs on rank 0 alone: no collective there
            opt.all_reduce_grads()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    value = PAIRS_PER_GPU * world * args.steps / dt

    # ---- roofline leg: HIP events around every launch of the dominant kernel class, same workload ------
    roof = None
    if rank == 0:
        per_class = {}
        classes = [args.prof_class] if args.prof_class else [1, 2, 5, 6]
        n_prof = 3
        out = (ctypes.c_double * 3)()
        for cls in classes:
            L.bmp_prof_start(cls)
            for i in range(n_prof):
                step(i, collective=False)
            torch.cuda.synchronize()
            n = L.bmp_prof_stop(out)
            if n:
                per_class[cls] = dict(launches=n / n_prof, ms=out[0] / n_prof, flops=out[1] / n_prof, bytes=out[2] / n_prof)
        if per_class:
            cls = max(per_class, key=lambda c: per_class[c]["ms"])
            pc = per_class[cls]
            pbs = [b[0] for b in batches[:n_prof]]
            real_frac = sum(p.n_real_atoms for p in pbs) / sum(p.n_rows for p in pbs)
            # executed flops count every row of the packed layout (virtual pad + dead rows);
            # algorithmic flops count real atoms only
            alg_flops = pc["flops"] * real_frac
            achieved = alg_flops / (pc["ms"] * 1e-3) / 1e12
            names = {1: "k_rowgemm (fp32 MFMA row GEMM)", 2: "k_wgrad_lds (fp32 MFMA weight-gradient GEMM)",
                     5: "k_ggnn_step_fwd", 6: "k_ggnn_step_bwd"}
            # HBM bytes per launch of that kernel: PMC counters (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 --pmc
            # passes of this same command, committed under profiles/), not measurable from inside this process
            traffic = None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01h_pmc_hbm_traffic.json")))
                key = {5: "k_ggnn_step_fwd<128, false>", 6: "k_ggnn_step_bwd<128, false>", 2: "k_wgrad_lds<false>",
                       1: "k_rowgemm<1, 4, 1, 0>"}.get(cls)
                if key in pmc:
                    # counters are in KiB; gfx950 tallies a 16-byte-per-lane streaming read at half its bytes
                    # (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE is doubled, WRITE_SIZE taken as is
                    traffic = round((2.0 * pmc[key]["FETCH_SIZE_per_launch"] + pmc[key]["WRITE_SIZE_per_launch"]) * 1024)
            except (OSError, ValueError):
                pass
            roof = dict(bound="mfma", achieved=round(achieved, 3), peak=PEAK_F32_TFLOPS, unit="TFLOP/s",
                        frac=round(achieved / PEAK_F32_TFLOPS, 4), traffic=traffic, kernel=names.get(cls, str(cls)),
                        launches_per_step=pc["launches"], avg_launch_us=round(1e3 * pc["ms"] / pc["launches"], 2),
                        alg_gflop_per_launch=round(alg_flops / pc["launches"] / 1e9, 4),
                        class_ms_per_step={str(k): round(v["ms"], 3) for k, v in per_class.items()},
                        measured="HIP events around every launch of the class, 3 steps after the timed region")
        alg_f = algorithmic_flops_per_pair(atoms_per_pair)
        alg_b = algorithmic_bytes_per_pair(atoms_per_pair, edges_per_pair)
        whole = dict(alg_mflop_per_pair=round(alg_f / 1e6, 1), alg_kb_per_pair=round(alg_b / 1e3, 1),
                     f32_frac=round(value / world * alg_f / (PEAK_F32_TFLOPS * 1e12), 4),
                     hbm_frac=round(value / world * alg_b / (PEAK_HBM_GBS * 1e9), 5))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(store, idx1, idx2, label)

    if rank == 0:
        print(json.dumps({
            "metric": "drug-pairs/sec fwd+bwd, binary-DDI GGNN d=128", "value": round(value, 1), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: full binary DDI set (544 drugs, 147696 pairs), GGNN 4-step d=128 tied + "
                                   "Nie co-attention (head 8, tanh) + MLP(32,16), fwd+bwd+Adam, "
                                   f"{PAIRS_PER_GPU} pairs/GPU/step, every molecule instance encoded",
                       "pairs_per_gpu": PAIRS_PER_GPU, "global_batch": gb, "parallelism": f"dp{world}",
                       "atoms_per_pair": round(atoms_per_pair, 2), "loss": round(float(loss.item()), 5)},
            "roofline": roof, "whole_step": whole, "cpu_baseline": cpu}))
    if world > 1 or force_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark: drug-pairs/sec, fwd+bwd(+Adam), binary-DDI GGNN d=128 + Nie co-attention.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4]
    (N > 1: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or bare: the process then
     starts that launcher itself as a child, before it has imported torch, and exits with its code)

A "step" is one training step over one batch of 1024 drug pairs per GPU: both molecules of every pair encoded, co-attention,
link predictor, sigmoid cross entropy, backward, one gradient all-reduce (N > 1), Adam.  Every molecule INSTANCE is encoded
(no per-batch de-duplication).  Configurations: c2 (default, the metric's configuration; c5 is c2 at N = 8)
GGNN 4-step d=128 tied + Nie; c3 RelGCN 3x128 + Nie; c4 GGNN 4-step d=256 + MLP(37), multi-label store.

Rank 0 prints ONE JSON line:
  value         resident leg: K timed steps over packed batches already in HBM (all batches of one epoch, cycled)
  end_to_end    one epoch with the reference's per-iteration collate inside the timed region (train_ddi_modify.py:280,
                295-296): a fresh permutation, per step the host plan (size arithmetic), one pinned H2D copy and the device
                kernel that writes the packed batch from the HBM-resident drug store
  batch32       c2 at the reference's default batch of 32 pairs (train_ddi_modify.py:196), end to end
  predict       forward only, under no-backprop: the evaluation callers' predict (eval_coattention.py:103-124)
  dedup         every distinct molecule of a step encoded once (SURVEY.md 8(d) caveat): reported beside `value`, never instead
  other_configs the default (c2) run also times c3 and c4 for 20 steps each, with their dominant kernel's roofline fraction
  ref_headline  the model the reference's published figures were trained with (DDI.md:6, RECORD.txt:246-251): GGNN hidden 32, 8
                steps untied + Nie + NTN / HolE, at 1024 pairs (resident) and at the reference's batch of 32; fused d = 32 step
                kernels against the unfused operator chain
  roofline      the dominant kernel and its class: durations from HIP events around every launch, measured in this run;
                `traffic` (HBM bytes per launch) is NOT measured in this run -- PMC counters need rocprofv3 -- but quoted from
                the committed profile named in `traffic_source` (same library version, config and kernel), null without one
  cpu_baseline  the oracle (dense restatement) on the host cores: same model, batch 32; more: C1 and batch 256
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

np = torch = dist = None     # bound by _heavy_imports(): nothing below the self-launch check may run before it


def _heavy_imports():
    """numpy / torch are imported only AFTER main() has decided whether this process is the launcher of `--gpus N`
    (self_launch): the launcher never loads the HIP runtime, let alone touches a device."""
    global np, torch, dist
    import numpy as _np
    import torch as _torch
    import torch.distributed as _dist
    np, torch, dist = _np, _torch, _dist


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` without an outer torchrun (the form of the driver's N = 1 command): start the N ranks as
    CHILD processes -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py
    <same arguments>` -- relay their output (rank 0 prints the JSON line) and return the launcher's exit code.  This
    process has not imported torch and never touches a GPU; nothing is re-exec'ed."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    port = int(env.get("BMP_BENCH_PORT") or _free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)

HEAD = 8
PAIRS_PER_GPU = 1024
PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    "c2": dict(encoder="ggnn", d=128, o=128, layers=4, attn="nie", class_num=1, store="binary",
               workload="C2: full binary DDI set (544 drugs, 147696 pairs), GGNN 4-step d=128 tied + Nie co-attention "
                        "(head 8, tanh) + MLP(32,16), fwd+bwd+Adam"),
    "c3": dict(encoder="relgcn", d=128, o=128, layers=3, attn="nie", class_num=1, store="binary",
               workload="C3: full binary DDI set, RelGCN 3-layer d=128 (scale_adj) + atoms tap + Nie co-attention + MLP(32,16), "
                        "fwd+bwd+Adam"),
    # the model every published figure of the reference was trained with (DDI.md:6, RECORD.txt:246-251): not a BASELINE.json
    # config -- the `ref_headline` leg of the default run
    "ref_ntn": dict(encoder="ggnn", d=32, o=16, layers=8, attn="nie", class_num=1, store="binary", tying=False, sim="ntn", mlp_hidden=(),
                    workload="reference's published model: GGNN hidden 32, 8 steps, weight_tying=False, fp_out 16 + Nie(head 8) + NTN "
                             "(--net-hidden-dims=), binary DDI set, fwd+bwd+Adam"),
    "ref_hole": dict(encoder="ggnn", d=32, o=16, layers=8, attn="nie", class_num=1, store="binary", tying=False, sim="hole", mlp_hidden=(),
                     workload="as ref_ntn with the HolE link predictor (DDI.md:6)"),
    "c4": dict(encoder="ggnn", d=256, o=256, layers=4, attn=None, class_num=37, store="multilabel",
               workload="C4: 37-class multi-label DDI (1704 drugs, 192000 pairs), GGNN 4-step d=256 tied + MLP(32,16) -> 37, "
                        "fwd+bwd+Adam"),
}

KERNEL_NAMES = {16: "k_rowgemm / k_rowgemm_db (fp32 MFMA row GEMM)", 17: "k_rowgemm_multi", 18: "k_readout_tile_fwd",
                32: "k_wgrad_lds<false> (fp32 MFMA weight-gradient GEMM)", 33: "k_wgrad_lds<true>",
                34: "k_wgrad_lds<false> one-hot (embedding gradient)", 35: "k_wgrad (direct)", 36: "k_wgrad_dma_multi<0>",
                37: "k_wgrad_dma_multi<1> (fp32 MFMA weight-gradient GEMM of a fused step: o1 | o2 | dUcT in one launch, stages by LDS-DMA)",
                64: "k_coattn_fwd", 65: "k_coattn_bwd", 80: "k_ggnn_step_fwd<D, false>", 81: "k_ggnn_step_fwd<D, true>",
                82: "k_relgcn_layer_fwd", 83: "k_ggnn_step_fwd<D, .., TS = true> (all T propagation steps of a tile in one launch)",
                96: "k_ggnn_step_bwd<D, false>", 97: "k_ggnn_step_bwd<D, true>",
                98: "k_relgcn_layer_bwd"}
CLASS_NAMES = {1: "row GEMMs", 2: "weight-gradient GEMMs", 3: "gathers", 4: "co-attention pair kernels",
               5: "fused step / layer forward", 6: "fused step / layer backward"}
# rocprofv3 kernel-name PREFIXES of the keys above (profiles/*_pmc_hbm_traffic.json), {D} = hidden width: the template
# arguments behind the prefix differ with the tile layout (whole tiles / tile table) and, for the row GEMM, with its epilogue;
# a key's traffic is the launch-weighted mean over the profile's kernels that carry the prefix
PMC_NAMES = {16: ("k_rowgemm<", "k_rowgemm_db<", "k_rowgemm_lds<"), 32: ("k_wgrad_lds<false>",), 33: ("k_wgrad_lds<true>",),
             36: ("k_wgrad_dma_multi<0>", "k_wgrad_lds_multi<0>"), 37: ("k_wgrad_dma_multi<1>", "k_wgrad_lds_multi<1>"), 80: ("k_ggnn_step_fwd<{D}, false",),
             81: ("k_ggnn_step_fwd<{D}, true",), 96: ("k_ggnn_step_bwd<{D}, false",), 97: ("k_ggnn_step_bwd<{D}, true",),
             82: ("k_relgcn_layer_fwd<{D}",), 83: ("k_ggnn_step_fwd<{D}",), 98: ("k_relgcn_layer_bwd<{D}",), 17: ("k_rowgemm_multi",), 18: ("k_readout_tile_fwd<{D}",)}


def algorithmic_flops_per_pair(cfg, n_pair: float):
    """SURVEY.md 8(d), real atoms only (no pad / dead rows, no folding credit), fwd+bwd = 3 x fwd.
    GGNN: 26 d^2 per atom-step (message 8 d^2 + GRU 18 d^2), readout 6 d o per atom; RelGCN: 10 d^2 per atom-layer,
    readout 4 d o.  With a fine co-attention the readout counts ONCE: it ignores g_1 / g_2 (nie_coattention.py:335-370),
    so the readout runs forward as in the reference but no gradient ever reaches it."""
    d, o, L = cfg["d"], cfg["o"], cfg["layers"]
    half = n_pair / 2.0
    if cfg["encoder"] == "ggnn":
        enc, readout = 26.0 * d * d * n_pair * L, 6.0 * d * o * n_pair
    else:
        enc, readout = 10.0 * d * d * n_pair * L, 4.0 * d * o * n_pair
    mlp = 2.0 * (2 * o * 32 + 32 * 16 + 16 * cfg["class_num"])
    if cfg.get("sim") == "ntn":          # Bilinear(o, o, 8) + V1, V2 (models/mlp.py:52) and the 8 -> class_num output layer
        mlp = 2.0 * (8 * o * o + 2 * 8 * o + 8 * cfg["class_num"])
    elif cfg.get("sim") == "hole":       # circular correlation as the direct sum (o^2 multiply-adds) + o -> class_num
        mlp = 2.0 * (o * o + o * cfg["class_num"])
    if cfg["attn"]:
        co = 2.0 * (half * d * d + half * half * d + n_pair * d * o + 2 * n_pair * HEAD * d)
        return 3.0 * (enc + co + mlp) + readout
    return 3.0 * (enc + readout + mlp)


def algorithmic_bytes_per_pair(cfg, n_pair: float, e_pair: float):
    """SURVEY.md 8(d) compulsory-traffic model: n d (20 L + 28 [+ 12 with co-attention]) + index bytes."""
    d, L = cfg["d"], cfg["layers"]
    return n_pair * d * (20.0 * L + 28.0 + (12.0 if cfg["attn"] else 0.0)) + 16.0 * L * (n_pair / 2 + 1 + e_pair / 2) + 4.0 * n_pair


def oracle_steps(store, idx1, idx2, label, B, seconds, hidden, layers, attn, class_num=1, max_steps=10 ** 9, warm=2):
    """The oracle (oracle/ref_cpu.py, dense op-for-op restatement of the reference) fwd+bwd+Adam on the host cores for
    about `seconds`; returns (pairs/s over the timed steps, timed steps, seconds used)."""
    from oracle import ref_cpu as O_
    from bmp import synth
    p = O_.make_pair_params(777, hidden_dim=hidden, out_dim=hidden, n_layers=layers, attn=attn, head=HEAD, class_num=class_num,
                            dtype=torch.float32, bias_scale=0.0)
    params = [p[n].requires_grad_() for n in sorted(p)]
    state = [dict(m=torch.zeros_like(x), v=torch.zeros_like(x)) for x in params]
    times, step = [], 0
    while True:
        sl = slice(step * B, (step + 1) * B)
        a1, j1 = synth.concat_mols([store[k] for k in idx1[sl]])
        a2, j2 = synth.concat_mols([store[k] for k in idx2[sl]])
        t = torch.from_numpy(label[sl].reshape(len(a1), -1))
        t0 = time.perf_counter()
        y, _, _ = O_.pair_forward(p, torch.from_numpy(a1), torch.from_numpy(j1), torch.from_numpy(a2), torch.from_numpy(j2),
                                  n_layers=layers, attn=attn)
        loss = O_.sigmoid_cross_entropy(y, t)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(x) for g, x in zip(grads, params)]
        with torch.no_grad():
            O_.chainer_adam_step(params, grads, state, step + 1)
        dt = time.perf_counter() - t0
        step += 1
        if step > warm:
            times.append(dt)
        if sum(times) >= seconds or len(times) >= max_steps or (step + 1) * B > len(idx1):
            break
    return B * len(times) / sum(times), len(times), sum(times), B / float(np.median(times))


def cpu_baseline(cfg, store, idx1, idx2, label):
    """`cpu_baseline`: the bench's own model at the reference's default batch of 32 (train_ddi_modify.py:196), ~10 s.
    `more` (binary configs only): C1 (GGNN 2-step d=16 + MLP, batch 32, the first 256 pairs)
    and the C2 model at batch 256."""
    # the GPU box exposes all host cores but grants a 16-core share per GPU: more threads only thrash
    torch.set_num_threads(min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))
    cores = torch.get_num_threads()
    v, n, s, med = oracle_steps(store, idx1, idx2, label, 32, 10.0, cfg["d"], cfg["layers"], cfg["attn"], cfg["class_num"])
    out = dict(value=round(v, 2), unit="pairs/s", cores=cores, kind="port",
               sample=f"{32 * n} pairs ({n} timed steps of batch 32, dense oracle fwd+bwd+Adam, fp32, {s:.1f} s) of the same "
                      f"workload; median step {med:.1f} pairs/s; torch {torch.__version__}")
    more = None
    if cfg["store"] == "binary" and cfg["encoder"] == "ggnn":
        v1, n1, s1, m1 = oracle_steps(store, idx1[:256 + 96], idx2[:256 + 96], label[:256 + 96], 32, 3.0, 16, 2, None, warm=3)
        v2, n2, s2, m2 = oracle_steps(store, idx1, idx2, label, 256, 8.0, cfg["d"], cfg["layers"], cfg["attn"], warm=1, max_steps=10)
        more = {"C1 GGNN 2-step d=16 + MLP, batch 32": dict(value=round(v1, 1), median=round(m1, 1), steps=n1, seconds=round(s1, 2)),
                "C2 model, batch 256": dict(value=round(v2, 2), median=round(m2, 2), steps=n2, seconds=round(s2, 2))}
    return out, more


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the end-to-end, batch-32 and roofline legs")
    ap.add_argument("--host-profile", action="store_true", help="diagnostic: cProfile of the end-to-end epoch's host side to stderr")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # not under a launcher: become one (children do the work)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("BMP_BENCH_RANK_CHECK_ONLY") == "1":        # tests/test_bench_launch.py: the launch path without a GPU
        os.write(1, (json.dumps({"rank_check": True, "rank": rank, "local_rank": local_rank, "world": world,
                                 "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}) + "\n").encode())
        return                  # (one write per rank: the ranks share the launcher's stdout)
    _heavy_imports()
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
    from bmp import synth, packed, _lib, enclayout
    from bmp.predictor import build_pair_predictor
    # Layout the encoder works on, per leg (BMP_BENCH_LAYOUT=encoder|instance forces one everywhere).  Measured on MI355X
    # (DESIGN.md 3a''): the encoder layout's balanced tile heights make the fused kernels 4-8 % shorter; its two index launches
    # and its second host plan cost about as much at 1024 pairs on c2 (352 k vs 354 k pairs/s resident); c3 gains 1.7 %
    # resident but its 1.47 ms step then waits for the host in the end-to-end leg (0.79 of resident instead of 0.97); the
    # 32-pair batch gains 14 % (its tiles spread over four times as many CUs).  So: per-instance batches for the 1024-pair
    # legs, the encoder layout for the 32-pair leg and -- always -- for de-duplication.
    LAYOUT = os.environ.get("BMP_BENCH_LAYOUT", "auto")
    AUTO = {"c2": "instance", "c3": "instance", "c4": "instance", "ref_ntn": "instance", "ref_hole": "instance"}
    from bmp.dp import FlatAdam
    L = _lib.lib()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, body):
        """barrier + synchronize, n_steps x body(i), barrier + synchronize; returns (max over ranks, this rank) seconds."""
        fence()
        t0 = time.perf_counter()
        for i in range(n_steps):
            body(i)
        fence()
        mine = time.perf_counter() - t0
        tt = torch.tensor([mine], device=dev, dtype=torch.float64)
        if world > 1:
            allr = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(allr, tt)
            every = [float(x.item()) for x in allr]
        else:
            every = [mine]
        return max(every), every

    class Env:
        """One configuration's workload and model: the drug store in HBM, the pair permutation, `n_batches` packed batches of
        this rank's shard (None: the whole epoch), the model and its optimizer (layout plan built at the first step)."""

        def __init__(self, name, n_batches=None):
            self.name, self.cfg = name, CONFIGS[name]
            c = self.cfg
            if c["store"] == "binary":
                self.store = synth.make_store()
                self.idx1, self.idx2, label = synth.make_pairs()
            else:
                self.store = synth.make_store(1704, seed=2018)
                self.idx1, self.idx2, label = synth.make_multilabel_pairs()
            self.label = label.reshape(len(self.idx1), -1)
            self.ms = packed.MolStore(self.store)
            self.dstore = packed.DeviceMolStore(self.ms, dev)
            self.gb = PAIRS_PER_GPU * world
            self.steps_per_epoch = len(self.idx1) // self.gb
            nb = self.steps_per_epoch if n_batches is None else min(n_batches, self.steps_per_epoch)
            self.batches = [self.collate(self.idx1, self.idx2, self.label, k) for k in range(nb)]      # resident
            inst = [getattr(b_, "pb", b_) for b_, _ in self.batches]               # the per-instance batches (real atoms, bonds)
            encs = [getattr(b_, "pb_enc", b_) for b_, _ in self.batches]           # what the encoder's kernels walk
            n_atoms = sum(pb.n_real_atoms for pb in inst)
            self.n_rows = sum(pb.n_rows for pb in encs)
            self.n_atoms = n_atoms
            self.atoms_per_pair = n_atoms / (len(self.batches) * PAIRS_PER_GPU)
            self.edges_per_pair = sum(pb.n_edges for pb in inst) / (len(self.batches) * PAIRS_PER_GPU)
            torch.manual_seed(777)
            self.model = build_pair_predictor(hidden_dim=c["d"], out_dim=c["o"], n_layers=c["layers"], attn=c["attn"], head=HEAD,
                                              encoder=c["encoder"], class_num=c["class_num"], weight_tying=c.get("tying", True),
                                              sim_method=c.get("sim", "mlp"), mlp_hidden=c.get("mlp_hidden", (32, 16))).to(dev)
            self.opt = FlatAdam(self.model, alpha=1e-3)
            self.opt.broadcast_parameters(0)
            if force_pg:
                self.opt.world = 2       # takes the all-reduce branch (a one-rank sum; the folded 1/2 only rescales the updates)
            self.alg_f = algorithmic_flops_per_pair(c, self.atoms_per_pair)
            self.alg_b = algorithmic_bytes_per_pair(c, self.atoms_per_pair, self.edges_per_pair)

        def collate(self, i1, i2, lab, k, B=PAIRS_PER_GPU, gbatch=None, dedup=False, layout=None):
            """Global batch k of the pair list (i1, i2, lab): this rank's shard, packed on the device -- the per-instance
            batch and, for the encoder, the encoder layout (bmp/enclayout.py: real atoms + one pad row per tile, tile heights
            balanced over the CUs; every molecule INSTANCE encoded unless ``dedup``).  BMP_BENCH_LAYOUT=instance: the
            per-instance batch alone (round 2's form)."""
            lo = k * (gbatch or B * world) + rank * B
            # (the encoder layout pays through the fused tile kernels, d = 64 / 128; the row-wise operators of other widths
            #  gain nothing from tile heights and would only carry the two extra index launches)
            layout = layout or (AUTO[self.name] if LAYOUT == "auto" else LAYOUT)
            if (layout == "instance" or self.cfg["d"] not in (32, 64, 128)) and not dedup:
                return packed.pack_from_store_device(self.dstore, [i1[lo:lo + B], i2[lo:lo + B]], labels=lab[lo:lo + B])
            return enclayout.encode_from_store_device(self.dstore, [i1[lo:lo + B], i2[lo:lo + B]], labels=lab[lo:lo + B], dedup=dedup)

        def train_step(self, pb, t, collective=True):
            opt = self.opt
            # parameters = views of the flat buffer, ONE gradient tensor; the model's forward_loss = the reference's Classifier
            # (train_ddi_modify.py:284-286): encoder, co-attention, link predictor and the loss against the labels
            loss = opt.functional_loss(pb, t=t)
            loss.backward()
            opt.collect_grads()
            if collective:                          # the roofline leg runs on rank 0 alone: no collective there
                opt.all_reduce_grads()
            opt.step()
            return loss

        def resident(self, steps, warmup):
            """The metric's leg: packed batches already in HBM, cycled.  Returns (pairs/s, seconds, per-rank seconds, loss)."""
            last = {}
            def body(i):
                last["loss"] = self.train_step(*self.batches[i % len(self.batches)])
            for i in range(warmup):
                body(i)
            dt, every = timed(steps, lambda i: body(warmup + i))
            loss = float(last["loss"].item())
            # the host's own time per step: three steps issued into empty queues (nothing to wait for), untimed by the leg
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3):
                body(i)
            self.host_ms_per_step = round(1e3 * (time.perf_counter() - t0) / 3, 3)
            torch.cuda.synchronize()
            return self.gb * steps / dt, dt, every, loss

        def whole(self, value):
            return dict(alg_mflop_per_pair=round(self.alg_f / 1e6, 1), alg_kb_per_pair=round(self.alg_b / 1e3, 1),
                        f32_frac=round(value / world * self.alg_f / (PEAK_F32_TFLOPS * 1e12), 4),
                        hbm_frac=round(value / world * self.alg_b / (PEAK_HBM_GBS * 1e9), 5))

        def roofline(self, n_prof=6, cap=64):
            """HIP events around every launch (all classes), same workload, with every launch whole and in line on one stream.
            In the timed region the weight-gradient launches run on a low-priority side stream beside the backward chain
            (bmp/plan.py SideStream) and the encoder's forward as two chains of tiles (PartStream): launches share the CUs, and
            an event pair around one then spans the sharing, not the kernel."""
            key = (ctypes.c_int * cap)(); cnt = (ctypes.c_int * cap)()
            ms_ = (ctypes.c_double * cap)(); fl = (ctypes.c_double * cap)(); by = (ctypes.c_double * cap)()
            plan = getattr(self.opt, "plan", None)
            side_saved, split_saved = getattr(plan, "side", None), getattr(plan, "split", None)
            if plan is not None:
                torch.cuda.synchronize()
                plan.side = plan.split = None
            L.bmp_prof_start(-1)
            t0p = time.perf_counter()
            for i in range(n_prof):
                self.train_step(*self.batches[i % len(self.batches)], collective=False)
            torch.cuda.synchronize()
            inline_ms = 1e3 * (time.perf_counter() - t0p) / n_prof
            nk = L.bmp_prof_collect(key, cnt, ms_, fl, by, cap)
            if plan is not None:
                plan.side, plan.split = side_saved, split_saved
            enc_of = lambda b_: getattr(b_, "pb_enc", b_)
            rows_prof = sum(enc_of(self.batches[i % len(self.batches)][0]).n_rows for i in range(n_prof))
            real_frac = sum(getattr(self.batches[i % len(self.batches)][0], "pb", self.batches[i % len(self.batches)][0]).n_real_atoms
                            for i in range(n_prof)) / rows_prof
            kern = {key[i]: dict(launches=cnt[i] / n_prof, ms=ms_[i] / n_prof, flops=fl[i] / n_prof) for i in range(nk)}
            if not kern:
                return None

            def entry(ks, name):
                msum = sum(kern[k]["ms"] for k in ks)
                lsum = sum(kern[k]["launches"] for k in ks)
                # the launchers state executed flops over every packed row (virtual pad + dead rows included);
                # algorithmic flops count real atoms only
                alg = sum(kern[k]["flops"] for k in ks) * real_frac
                ach = alg / (msum * 1e-3) / 1e12 if msum > 0 else 0.0
                return dict(kernel=name, launches_per_step=round(lsum, 2), avg_launch_us=round(1e3 * msum / max(lsum, 1e-9), 2),
                            ms_per_step=round(msum, 4), alg_gflop_per_launch=round(alg / max(lsum, 1e-9) / 1e9, 4),
                            achieved=round(ach, 3), frac=round(ach / PEAK_F32_TFLOPS, 4))
            top = max(kern, key=lambda k: kern[k]["ms"])
            cls = top // 16
            k_e = entry([top], KERNEL_NAMES.get(top, str(top)))
            c_e = entry([k for k in kern if k // 16 == cls], CLASS_NAMES.get(cls, str(cls)))
            # HBM bytes per launch of that kernel: PMC counters (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 --pmc passes of
            # this command, committed under profiles/); taken only from a profile of THIS library version, config and kernel
            traffic = traffic_source = None
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):
                try:
                    pmc = json.load(open(path))
                except (OSError, ValueError):
                    continue
                pre = tuple(x.replace("{D}", str(self.cfg["d"])) for x in PMC_NAMES.get(top, ()))
                hits = [v for k_, v in pmc.items() if isinstance(v, dict) and pre and k_.startswith(pre)
                        and "FETCH_SIZE_per_launch" in v and "WRITE_SIZE_per_launch" in v]
                if pmc.get("_bmp_version") == L.bmp_version() and pmc.get("_config", "c2") == self.name and hits:
                    # counters are in KiB; gfx950 tallies a 16-byte-per-lane streaming read at half its bytes
                    # (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE is doubled, WRITE_SIZE taken as is
                    nl = sum(v["launches"] for v in hits)
                    traffic = round(sum((2.0 * v["FETCH_SIZE_per_launch"] + v["WRITE_SIZE_per_launch"]) * v["launches"] for v in hits)
                                    / max(nl, 1) * 1024)
                    traffic_source = dict(file=os.path.relpath(path, ROOT), bmp_version=pmc.get("_bmp_version"),
                                          how="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; "
                                              "2 x FETCH + WRITE per launch; replayed, not measured in this run")
                    break
            if traffic is None:
                print(f"[bench] no profiles/r*_pmc_hbm_traffic.json for bmp_version {L.bmp_version()} / {self.name} / "
                      f"{KERNEL_NAMES.get(top, top)}: roofline.traffic is null", file=sys.stderr)
            return dict(bound="mfma", achieved=k_e["achieved"], peak=PEAK_F32_TFLOPS, unit="TFLOP/s", frac=k_e["frac"],
                        traffic=traffic, traffic_source=traffic_source, **{k: v for k, v in k_e.items() if k not in ("achieved", "frac")}, kernel_class=c_e,
                        per_kernel_ms_per_step={KERNEL_NAMES.get(k, str(k)): round(v["ms"], 4) for k, v in
                                                sorted(kern.items(), key=lambda kv: -kv[1]["ms"])},
                        measured=f"HIP events around every launch of every instrumented kernel, {n_prof} steps after the timed "
                                 "region, each on its launch stream, with the side stream and the two-chain forward switched off for "
                                 "these steps so that no two launches share the CUs (the events' own cost makes such a step "
                                 f"{inline_ms:.2f} ms); achieved = algorithmic flops (real atoms only) / event time",
                        streams="side stream (weight gradients) and two forward chains on in the timed region"
                        if side_saved is not None else "one stream")

    def graphed_batch32(e, epoch, n_steps=300):
        """32-pair steps of Env ``e``, end to end, as replays of one HIP graph on a fixed-shape batch."""
        from bmp.dp import GraphedTrainStep
        sb = packed.StaticPairBatch(e.dstore, 32, label_cols=e.label.shape[1])
        stepper = GraphedTrainStep(e.model, e.opt)
        perm = np.random.RandomState(1000 + epoch).permutation(len(e.idx1))
        p1, p2, pl = e.idx1[perm], e.idx2[perm], e.label[perm]
        host = [0.0]

        def body(i):
            t0 = time.perf_counter()
            lo = (i % (len(p1) // 32)) * 32
            sb.load([p1[lo:lo + 32], p2[lo:lo + 32]], pl[lo:lo + 32])
            stepper(sb)
            host[0] += time.perf_counter() - t0
        for i in range(10):                 # (the first call warms up and records)
            body(i)
        host[0] = 0.0
        sb.wait_seconds = 0.0
        dt_g, _ = timed(n_steps, lambda i: body(i + 10))
        host[0] -= sb.wait_seconds          # (load() blocks when the host is four steps ahead of the GPU: waiting, not work)
        loss_g = float(stepper.graphs[next(iter(stepper.graphs))][1].detach())
        v = 32 * n_steps / dt_g
        return dict(value=round(v, 1), unit="pairs/s", steps=n_steps, ms_per_step=round(1e3 * dt_g / n_steps, 3),
                    host_ms_per_step=round(1e3 * host[0] / n_steps, 3), loss=round(loss_g, 5),
                    f32_frac=round(v * e.alg_f / (PEAK_F32_TFLOPS * 1e12), 5), mode="hip graph on a fixed-shape batch")

    env = Env(args.config)
    atoms_per_pair_main, real_row_fraction_main = env.atoms_per_pair, env.n_atoms / env.n_rows
    env_main_batch0 = env.batches[0][0]
    cfg, store, idx1, idx2, label = env.cfg, env.store, env.idx1, env.idx2, env.label
    gb, steps_per_epoch, batches, dstore, opt = env.gb, env.steps_per_epoch, env.batches, env.dstore, env.opt

    # ---- resident leg (the metric): packed batches already in HBM ---------------------------------------
    value, dt, every, loss_val = env.resident(args.steps, args.warmup)
    env_main_host_ms = env.host_ms_per_step       # what the host needs to issue one step (measured into empty queues, outside the timed region)
    rank_ms = None if world == 1 else dict(min=round(1e3 * min(every) / args.steps, 3), max=round(1e3 * max(every) / args.steps, 3))

    # ---- end-to-end leg: one epoch, fresh permutation, collate inside the timed region --------------------
    e2e = b32 = pred = None
    if not args.no_extras:
        host_ms = []

        def epoch_body(state):
            def body(i):
                if i == 0:          # the reference's SerialIterator shuffles once per epoch
                    perm = np.random.RandomState(1000 + state["epoch"]).permutation(len(idx1))
                    state["p"] = (idx1[perm], idx2[perm], label[perm])
                t0 = time.perf_counter()
                pb, t = env.collate(*state["p"], i, B=state["B"], layout=state.get("layout"))
                host_ms.append(time.perf_counter() - t0)
                env.train_step(pb, t)
                state["host_s"] = state.get("host_s", 0.0) + (time.perf_counter() - t0)
            return body
        st = dict(epoch=0, B=PAIRS_PER_GPU)
        body = epoch_body(st)
        for i in range(3):
            body(i)
        host_ms.clear()
        dstore.plan_seconds, dstore.plan_calls = 0.0, 0
        if args.host_profile:
            import cProfile, pstats
            prof = cProfile.Profile()
            prof.enable()
        dt_e, _ = timed(steps_per_epoch, body)
        if args.host_profile:
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(35)
        e2e = dict(value=round(gb * steps_per_epoch / dt_e, 1), unit="pairs/s", steps=steps_per_epoch,
                   ms_per_step=round(1e3 * dt_e / steps_per_epoch, 3), ratio_to_resident=round(gb * steps_per_epoch / dt_e / value, 4),
                   host_plan_ms_per_batch=round(1e3 * dstore.plan_seconds / max(dstore.plan_calls, 1), 3),
                   collate_call_ms_per_batch=round(1e3 * float(np.mean(host_ms)), 3),
                   what="host_plan = the host's size arithmetic (bmp_collate_plan + pair metadata + labels into the pinned buffer); "
                        "collate_call = the whole call incl. allocations, launches and waiting for a free staging buffer when "
                        "the host runs ahead of the GPU.  One epoch: fresh permutation, per step host plan + pinned H2D (plan table, labels, pair metadata) + "
                        "bmp_collate_emit from the HBM-resident store, then the training step; nothing pre-packed")
        if args.config == "c2" and world == 1:
            graph32 = graphed_batch32(env, epoch=2)
            st = dict(epoch=1, B=32, layout=None if LAYOUT != "auto" else "encoder")
            body = epoch_body(st)
            for i in range(10):
                body(i)
            n32 = 300
            if args.host_profile:
                prof = cProfile.Profile()
                prof.enable()
            st["host_s"] = 0.0
            dt_32, _ = timed(n32, lambda i: body(i + 1))
            if args.host_profile:
                prof.disable()
                pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(45)
            eager32 = dict(value=round(32 * n32 / dt_32, 1), unit="pairs/s", steps=n32, ms_per_step=round(1e3 * dt_32 / n32, 3),
                           host_ms_per_step=round(1e3 * st["host_s"] / n32, 3), layout=st["layout"] or LAYOUT,
                           what="launch by launch through the framework's autograd (collate + ~55 launches per step): "
                                "host_ms_per_step = ms_per_step, the figure follows the box's host")
            b32 = dict(graph32, eager=eager32,
                       what="the same model at the reference's default batch of 32 pairs (train_ddi_modify.py:196), end to end: fresh "
                            "permutation, per step the host plans the batch (numpy on 64 integers + pair metadata), one pinned H2D "
                            "copy, then ONE HIP graph -- bmp_collate_emit, forward, loss, backward, weight gradients, Adam, recorded "
                            "once on a fixed-shape batch (bmp.packed.StaticPairBatch: one molecule per 128-row tile, every pair in "
                            "the pair kernels' 128-row class) -- is replayed (bmp.dp.GraphedTrainStep); host_ms_per_step = time the "
                            "host spends in the step's calls.  `eager`: the same steps launch by launch")
        if world == 1:
            # ---- forward-only leg: the evaluation callers' predict (eval_coattention.py:103-124; the evaluator extensions run
            #      it over the train and validation sets every epoch, training/extensions/batch_evaluator.py:49-100) ----
            n_p = 40
            for i in range(4):
                opt.functional_predict(batches[i % len(batches)][0])
            dt_p, _ = timed(n_p, lambda i: opt.functional_predict(batches[(4 + i) % len(batches)][0]))
            pred = dict(value=round(gb * n_p / dt_p, 1), unit="pairs/s", steps=n_p, ms_per_step=round(1e3 * dt_p / n_p, 3),
                        what="predict under no-backprop on the planned path (FlatAdam.functional_predict): logits + the two molecule "
                             "vectors, 1024 pairs per step, batches resident; the kernels keep nothing for a backward")

    # ---- de-duplication leg (SURVEY.md 8(d) caveat: reported BESIDE the per-instance figure, never instead of it) ----
    dedup = None
    if not args.no_extras and world == 1 and cfg["attn"]:
        nd, n_d = min(24, steps_per_epoch), 40
        dds = [env.collate(idx1, idx2, label, k, dedup=True) for k in range(nd)]
        for i in range(4):
            env.train_step(*dds[i % nd])
        dt_d, _ = timed(n_d, lambda i: env.train_step(*dds[(4 + i) % nd]))
        v_d = gb * n_d / dt_d
        for i in range(4):
            opt.functional_predict(dds[i % nd][0])
        dt_pd, _ = timed(n_d, lambda i: opt.functional_predict(dds[(4 + i) % nd][0]))
        dedup = dict(value=round(v_d, 1), unit="pairs/s", steps=n_d, ms_per_step=round(1e3 * dt_d / n_d, 3),
                     predict_value=round(gb * n_d / dt_pd, 1), predict_ms_per_step=round(1e3 * dt_pd / n_d, 3),
                     distinct_per_step=round(float(np.mean([d_.n_encoded for d_, _ in dds])), 1), instances_per_step=2 * PAIRS_PER_GPU,
                     rows_encoded_per_step=round(float(np.mean([d_.pb_enc.n_rows for d_, _ in dds]))),
                     speedup=round(v_d / value, 3),
                     what="every DISTINCT molecule of a step encoded once (bmp/enclayout.py, dedup=True), co-attention, MLP, loss, backward and Adam as "
                          "in `value`; same pairs, same result up to float32 summation order.  `value`, `roofline` and `whole_step` "
                          "are per-instance figures and do not include this.  predict_value: the forward-only leg the same way "
                          "(the reference's evaluators run predict over the train and validation sets every epoch, "
                          "train_ddi_modify.py:305-372)")
        del dds

    # ---- roofline leg: HIP events around every launch (all classes), same workload, rank 0 -------------------
    whole = env.whole(value) if rank == 0 else None
    roof = env.roofline() if (rank == 0 and not args.no_extras) else None

    # ---- the other single-GPU configurations of BASELINE.json, where the driver sees them (default run only) ----
    others = None
    if rank == 0 and world == 1 and not args.no_extras and args.config == "c2" and os.environ.get("BMP_BENCH_OTHERS", "1") != "0":
        others = {}
        del batches
        for name in ("c3", "c4"):
            env.batches = env.batches[:1]          # (frees the epoch's packed batches of the previous configuration)
            torch.cuda.empty_cache()
            e = Env(name, n_batches=24)
            v, dt_o, _, loss_o = e.resident(20, 4)
            r = e.roofline()
            others[name] = dict(value=round(v, 1), unit="pairs/s", ms_per_step=round(1e3 * dt_o / 20, 3), steps=20, warmup=4,
                                host_ms_per_step=e.host_ms_per_step,
                                whole_step=e.whole(v), loss=round(loss_o, 5), workload=e.cfg["workload"],
                                dominant_kernel=None if r is None else dict(kernel=r["kernel"], frac=r["frac"], achieved=r["achieved"],
                                                                            avg_launch_us=r["avg_launch_us"], traffic=r["traffic"],
                                                                            kernel_class_frac=r["kernel_class"]["frac"]),
                                per_kernel_ms_per_step=None if r is None else r["per_kernel_ms_per_step"])
            env = e

    # ---- the reference's PUBLISHED model (DDI.md:6, RECORD.txt:246-251; train_binary.py:165-187,226-227): GGNN hidden 32, 8
    #      steps, untied, fp_out 16 + Nie + NTN / HolE without hidden layers, at the reference's batch of 32 and at 1024 pairs ----
    ref_head = None
    if rank == 0 and world == 1 and not args.no_extras and args.config == "c2" and os.environ.get("BMP_BENCH_REF", "1") != "0":
        ref_head = {}
        for name in ("ref_ntn", "ref_hole"):
            torch.cuda.empty_cache()
            e = Env(name, n_batches=24)
            v, dt_r, _, loss_r = e.resident(30, 6)
            ent = dict(batch1024=dict(value=round(v, 1), unit="pairs/s", ms_per_step=round(1e3 * dt_r / 30, 3), steps=30,
                                      f32_frac=e.whole(v)["f32_frac"], alg_mflop_per_pair=e.whole(v)["alg_mflop_per_pair"],
                                      loss=round(loss_r, 5)), workload=e.cfg["workload"])
            if name == "ref_ntn":
                # the same model with its propagation steps on the unfused operators (gather + message row GEMM + the GRU's
                # row GEMMs: what d = 32 ran before the one-wave-per-block kernels of csrc/bmp_fused_small.hip): A/B, same box
                e.model.graph_conv.fused_small = False
                e.opt = FlatAdam(e.model, alpha=1e-3)
                v_u, dt_u, _, _ = e.resident(30, 6)
                ent["batch1024"]["unfused_value"] = round(v_u, 1)
                ent["batch1024"]["fused_speedup"] = round(v / v_u, 3)
                e.model.graph_conv.fused_small = True
                e.opt = FlatAdam(e.model, alpha=1e-3)
            # batch 32, end to end (collate inside), the reference's default (train_binary.py:330)
            for lay in ("instance", "encoder"):
                b32s = [e.collate(e.idx1, e.idx2, e.label, k, B=32, layout=lay) for k in range(64)]
                for i in range(10):
                    e.train_step(*b32s[i])
                n_b = 200
                t_host = [0.0]
                def body32(i):
                    t0 = time.perf_counter()
                    e.train_step(*b32s[i % 64])
                    t_host[0] += time.perf_counter() - t0
                dt_b, _ = timed(n_b, body32)
                ent[f"batch32_{lay}"] = dict(value=round(32 * n_b / dt_b, 1), unit="pairs/s", ms_per_step=round(1e3 * dt_b / n_b, 3),
                                             host_ms_per_step=round(1e3 * t_host[0] / n_b, 3), steps=n_b,
                                             f32_frac=round(32 * n_b / dt_b * e.alg_f / (PEAK_F32_TFLOPS * 1e12), 5))
                del b32s
            ent["batch32_graph"] = graphed_batch32(e, epoch=3, n_steps=200)
            ref_head[name] = ent
            del e

    cpu = cpu_more = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cpu_more = cpu_baseline(cfg, store, idx1, idx2, label)

    if rank == 0:
        names = {"c2": "drug-pairs/sec fwd+bwd, binary-DDI GGNN d=128", "c3": "drug-pairs/sec fwd+bwd, binary-DDI RelGCN d=128",
                 "c4": "drug-pairs/sec fwd+bwd, 37-class multi-label DDI GGNN d=256",
                 "ref_ntn": "drug-pairs/sec fwd+bwd, binary-DDI GGNN d=32 x 8 untied + Nie + NTN",
                 "ref_hole": "drug-pairs/sec fwd+bwd, binary-DDI GGNN d=32 x 8 untied + Nie + HolE"}
        line = {
            "metric": names[args.config], "value": round(value, 1), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["workload"] + f", {PAIRS_PER_GPU} pairs/GPU/step, every molecule instance encoded; the "
                                   f"{steps_per_epoch} batches of one epoch resident in HBM, cycled",
                       "layout": ("encoder layout: real atoms + one pad row per tile, tiles of 1..4 live 32-row blocks balanced over "
                                  "the 256 CUs (bmp/enclayout.py); readout and co-attention on the per-instance rows")
                       if hasattr(env_main_batch0, "pb_enc") else "per-instance packed layout, whole 128-row tiles",
                       "pairs_per_gpu": PAIRS_PER_GPU, "global_batch": gb, "parallelism": f"dp{world}",
                       "atoms_per_pair": round(atoms_per_pair_main, 2), "real_row_fraction": round(real_row_fraction_main, 4),
                       "loss": round(loss_val, 5)},
            "host_ms_per_step": env_main_host_ms,
            "roofline": roof, "whole_step": whole, "end_to_end": e2e, "batch32": b32, "predict": pred, "dedup": dedup, "other_configs": others,
            "ref_headline": ref_head, "cpu_baseline": cpu}
        if cpu_more:
            line["cpu_baseline_more"] = cpu_more
        if rank_ms:
            line["rank_ms_per_step"] = rank_ms
        if world > 1 or force_pg:          # what the process group itself reports (not the command line)
            line["dist"] = dict(world_size=dist.get_world_size(), backend=dist.get_backend(),
                                launcher=os.environ.get("TORCHELASTIC_RUN_ID") is not None)
        print(json.dumps(line))
    if world > 1 or force_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

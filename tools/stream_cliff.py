#!/usr/bin/env python
"""Reproducer of the 'fifth stream' cliff (DESIGN.md 3a', VERDICT r2 weak #6b): a dependent chain on the caller's stream with
K auxiliary streams forked / joined by events inside it, timed per 'step' for K = 0..7.  Each step: NCHAIN kernels on the
main stream; after kernel j (j % FORK_EVERY == 0) aux stream (j / FORK_EVERY) % K waits for the main stream, runs one
kernel, and the main stream waits for it two kernels later -- the shape of the planned training step (side stream for the
weight gradients, second forward chain, collate stream).  Run under rocprofv3 --kernel-trace to see which hardware queue
every stream's kernels were dispatched on (Queue_Id column).

    python tools/stream_cliff.py [--prio low|normal|mixed] [--size 1024] [--kmax 7]
"""
import argparse
import json
import os
import sys
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prio", default="mixed", help="aux stream priorities: normal | low | mixed (first aux low, rest normal)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--kmax", type=int, default=7)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--chain", type=int, default=24)
    ap.add_argument("--fork-every", type=int, default=3)
    ap.add_argument("--only", type=int, default=-1, help="run only this K (for a profiler trace)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n = a.size
    x = torch.randn(n, n, device=dev)
    w = torch.randn(n, n, device=dev) * 0.01
    ys = [torch.empty(n, n, device=dev) for _ in range(8)]
    aux_bufs = [torch.empty(n, n, device=dev) for _ in range(8)]
    main_s = torch.cuda.current_stream()
    lo, hi = -1, 0
    try:
        lo, hi = torch.cuda.Stream.priority_range()
    except Exception:
        pass

    def mk(k):
        if a.prio == "normal":
            return torch.cuda.Stream(device=dev)
        if a.prio == "low" or (a.prio == "mixed" and k == 0):
            return torch.cuda.Stream(device=dev, priority=0)      # torch: 0 = lowest, negative = higher
        return torch.cuda.Stream(device=dev)

    res = []
    ks = range(0, a.kmax + 1) if a.only < 0 else [a.only]
    for K in ks:
        aux = [mk(k) for k in range(K)]

        def step():
            pend = []
            for j in range(a.chain):
                torch.mm(x, w, out=ys[j % 8])
                if K and j % a.fork_every == 0:
                    s = aux[(j // a.fork_every) % K]
                    s.wait_stream(main_s)
                    with torch.cuda.stream(s):
                        torch.mm(ys[j % 8], w, out=aux_bufs[(j // a.fork_every) % 8])
                    pend.append((j + 2, s))
                while pend and pend[0][0] <= j:
                    main_s.wait_stream(pend.pop(0)[1])
            for _j, s in pend:
                main_s.wait_stream(s)

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        res.append(dict(K=K, ms_per_step=round(ms, 4)))
        print(json.dumps(res[-1]), flush=True)
    print(json.dumps(dict(prio=a.prio, size=n, chain=a.chain, fork_every=a.fork_every,
                          GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES"), results=res)))


if __name__ == "__main__":
    main()

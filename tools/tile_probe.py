#!/usr/bin/env python
"""Time the fused step kernels on ranges of an encoder-layout tile table: tall tiles alone, short tiles alone, all of them
(one launch), against the per-instance layout's whole tiles.  python tools/tile_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gcn-bmp_amd")]
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from bmp import synth, packed, enclayout, _lib
from bmp._lib import check, ptr, stream
from bmp.functional import pack_k4
L = _lib.lib()
dev = torch.device("cuda:0")
store = synth.make_store(); ms = packed.MolStore(store); ds = packed.DeviceMolStore(ms, dev)
i1, i2, lab = synth.make_pairs()
d = 128
g = torch.Generator().manual_seed(0)
r = lambda *s: (0.2 * torch.randn(*s, generator=g)).to(dev)
WTp, bE, ATp, UcTp, b = pack_k4(r(4 * d, d)), r(4, d), pack_k4(r(2 * d, 3 * d)), pack_k4(r(d, d)), r(3 * d)

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for n_cu in (256,):
    eb = enclayout.encode_from_store_device(ds, [i1[:1024], i2[:1024]], n_cu=n_cu)
    pbi = eb.pb
    for name, pb in (("instance", pbi), ("encoder", eb.pb_enc)):
        N = pb.n_rows
        h = r(N, d); m = torch.empty(N, d, device=dev); rz = torch.empty(N, 2 * d, device=dev); c = torch.empty(N, d, device=dev); ho = torch.empty(N, d, device=dev)
        nb = pb.mt_nblk.cpu().numpy() if pb.mt_nblk is not None else np.full(pb.n_tiles, 4)
        def run(t0, nt, first=0):
            check(L.bmp_ggnn_step_fwd(ptr(h), t0, nt, d, first, ptr(pb.csr_ptr), ptr(pb.csr_col), ptr(pb.csr_val), ptr(WTp), ptr(bE), ptr(ATp),
                                      ptr(UcTp), ptr(b), ptr(m), ptr(rz), ptr(c), ptr(ho), ptr(pb.mt_row0), ptr(pb.mt_nblk), pb.n_rows, 0, stream()), "f")
        T = pb.n_mtiles
        print(name, "tiles", T, "heights", np.bincount(nb), "rows", N)
        print("  all tiles          %7.1f us" % timeit(lambda: run(0, T)))
        for lo, hi in ((0, 256), (256, T), (0, 128), (0, 64), (256, 384)):
            if hi <= T and hi > lo:
                print("  tiles [%3d, %3d) h=%s  %7.1f us" % (lo, hi, sorted(set(nb[lo:hi].tolist())), timeit(lambda: run(lo, hi - lo))))

import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "gcn-bmp_amd"))
import torch, numpy as np
from bmp import synth, packed, _lib, functional as Fn
from bmp._lib import ptr, stream, check
L = _lib.lib()
dev = torch.device("cuda:0")
store = synth.make_store(); ms = packed.MolStore(store); i1, i2, _ = synth.make_pairs()
pb = packed.pack_from_store(ms, [i1[:64], i2[:64]], device=dev)
d = 128; N = pb.n_rows
torch.manual_seed(0)
for first in (1, 0):
    h = torch.randn(N, d, device=dev); rz = torch.rand(N, 2*d, device=dev); c = torch.rand(N, d, device=dev)*2-1
    g = torch.randn(N, d, device=dev)
    Wnat = torch.randn(d, 4*d, device=dev)*0.1; A = torch.randn(3*d, 2*d, device=dev)*0.1; Uc = torch.randn(d, d, device=dev)*0.1
    Wn4 = Fn.pack_k4(Wnat.t().contiguous().t()) if False else None
    # kernel wants K4 packed [K/4][N][4] of the K-major matrices: Wnat (K=d rows) etc.
    pk = lambda W: W.reshape(W.shape[0]//4, 4, W.shape[1]).permute(0, 2, 1).contiguous()
    dh = torch.empty(N, d, device=dev); gda = torch.full((N, 7*d), float('nan'), device=dev)
    check(L.bmp_ggnn_step_bwd(ptr(g), ptr(h), ptr(rz), ptr(c), pb.n_tiles, d, first, ptr(pb.csrT_ptr), ptr(pb.csrT_col), ptr(pb.csrT_val),
          ptr(pk(Wnat)), ptr(pk(A)), ptr(pk(Uc)), ptr(dh), ptr(gda), stream()), "bwd")
    torch.cuda.synchronize()
    r, z = rz[:, :d], rz[:, d:]
    dac = g*z*(1-c*c)
    hh = h if not first else torch.zeros_like(h)
    dz = g*(c-hh)*z*(1-z)
    print("first", first, "dac err", (gda[:, 6*d:]-dac).abs().max().item(), "dz err", (gda[:, 5*d:6*d]-dz).abs().max().item(),
          "nan in gda", torch.isnan(gda).sum().item())
    if not first:
        drh = dac @ Uc
        dar = drh*h*r*(1-r)
        print("  dar err", (gda[:, 4*d:5*d]-dar).abs().max().item())
    # ---- where are the mismatches? ----
    for name, blk, ref in (("dac", 6, dac), ("dz", 5, dz)):
        got = gda[:, blk*d:(blk+1)*d]
        bad = (got - ref).abs() > 1e-4
        print(" ", name, "bad frac", bad.float().mean().item(), "bad rows mod 8 hist", torch.bincount((bad.any(1).nonzero().flatten() % 8), minlength=8).tolist(),
              "rows mod 128 //8 hist", torch.bincount(((bad.any(1).nonzero().flatten() % 128) // 8), minlength=16).tolist())
        if bad.any():
            rr, cc = bad.nonzero()[0].tolist()
            val = got[rr, cc].item()
            print("   first bad at", rr, cc, "got", val, "exp", ref[rr, cc].item())
            for nm2, arr in (("dac", dac), ("dz", dz), ("g", g), ("c", c), ("z", z)):
                hit = ((arr - val).abs() < 1e-6).nonzero()
                if len(hit): print("   value found in", nm2, "at", hit[:3].tolist())

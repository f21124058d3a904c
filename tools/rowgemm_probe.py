"""Time the row GEMM (bmp_linear_fwd) alone on the shapes of config C4 (d = 256): usage
   [BMP_ROWGEMM_FORM=1] python tools/rowgemm_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import torch
from bmp import _lib
from bmp._lib import ptr, stream, check
L = _lib.lib()
dev = torch.device("cuda:0")
n_tiles = 455
N = n_tiles * 128
for K, Nout in ((1024, 256), (512, 768), (256, 256), (256, 1024), (768, 512)):
    X = torch.randn(N, K, device=dev); W = torch.randn(K, Nout, device=dev); Y = torch.empty(N, Nout, device=dev)
    b = torch.randn(Nout, device=dev)
    for _ in range(3):
        check(L.bmp_linear_fwd(ptr(X), K, n_tiles, K, Nout, ptr(W), Nout, ptr(b), 0, ptr(Y), Nout, stream()), "lin")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        check(L.bmp_linear_fwd(ptr(X), K, n_tiles, K, Nout, ptr(W), Nout, ptr(b), 0, ptr(Y), Nout, stream()), "lin")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = X[:256] @ W + b
    err = (Y[:256] - ref).abs().max().item() / ref.abs().max().item()
    print(f"K={K} Nout={Nout}: {ms*1e3:.1f} us  {2.0*N*K*Nout/ms/1e9:.1f} TFLOP/s  relerr {err:.1e}")
    # the same launch with every row tile reading the same rows (ldx = 0): operands from L1 / L2 only
    e0.record()
    for _ in range(10):
        check(L.bmp_linear_fwd(ptr(X), 0, n_tiles, K, Nout, ptr(W), Nout, ptr(b), 0, ptr(Y), Nout, stream()), "lin")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"      rows from cache: {ms*1e3:.1f} us  {2.0*N*K*Nout/ms/1e9:.1f} TFLOP/s")

"""Worker of tests/test_gpu_rowgemm_forms.py: GRU (both epilogue kinds), a message layer with self connection and a
linear layer with the gate multiplicand, forward and backward, at a size where the row GEMM launcher takes its big-shape
branch; results to the file named on the command line.  BMP_ROWGEMM_DIRECT in the environment selects the direct form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd")):
    sys.path.insert(0, p)
import numpy as np          # noqa: E402
import torch                # noqa: E402

from bmp import functional as Fn, packed, synth      # noqa: E402


def main(out_path):
    dev = torch.device("cuda:0")
    store = synth.make_store(400, seed=3, n_lo=20, n_hi=90, n_mean=60)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(0)
    pb = packed.pack_from_store(ms, [rs.randint(0, 400, 330), rs.randint(0, 400, 330)], device=dev)
    assert pb.n_tiles > 256, pb.n_tiles
    d = 72
    g = torch.Generator().manual_seed(1)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.2).to(dev).requires_grad_()
    h, m = mk(pb.n_rows, d), mk(pb.n_rows, d)
    AT, UcT, b = mk(2 * d, 3 * d), mk(d, d), mk(3 * d)
    WT, bE, WsT, bs = mk(4 * d, d), mk(4, d), mk(d, d), mk(d)
    cw = (torch.randn(pb.n_rows, d, generator=g)).to(dev)
    res = {}
    for first in (False, True):
        y = Fn.GRUFn.apply(h, m, AT, UcT, b, pb, first)
        gs = torch.autograd.grad((y * cw).sum(), [h, m, AT, UcT, b], allow_unused=True)
        res[f"gru{int(first)}"] = [y] + [x if x is not None else torch.zeros(1, device=dev) for x in gs]
    y = Fn.MsgFn.apply(h, WT, bE, WsT, bs, pb, Fn.ACT["tanh"])
    res["msg"] = [y] + list(torch.autograd.grad((y * cw).sum(), [h, WT, bE, WsT, bs]))
    torch.save({k: [t.detach().cpu() for t in v] for k, v in res.items()}, out_path)


if __name__ == "__main__":
    main(sys.argv[1])

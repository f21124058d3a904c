"""``BiMPM`` with the reference's constructor and call signatures (models/coattention/bimpm.py:17-199; built at
train_binary.py:253-256 with head = fp_out_dim and aggr = F.sum).

``__call__(atoms_1, g1, atoms_2, g2) -> (mol_1, mol_2)``, each (mb, n_match * head): the reference concatenates its three
matchings and never applies an output layer (bimpm.py:35 is commented out), so ``out_dim`` is 3 * head here, which is what
the link predictor's lazily sized first layer (models/mlp.py:34-37) then sees.
"""
from __future__ import annotations

import math

import torch
from torch import nn
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr, stream
from .coattention import pair_rows
from .ggnn import as_packed_atoms


class BiMPMFn(Function):
    """bmp_bimpm_fwd / _bwd (include/bmp.h)."""

    @staticmethod
    def forward(ctx, X1, X2, P, Q, R, w1, w2, meta, maxn):
        L = _lib.lib()
        X1, X2 = X1.contiguous(), X2.contiguous()
        P, Q, R = P.contiguous(), Q.contiguous(), R.contiguous()
        d, H, B = X1.shape[1], P.shape[0], meta["B"]
        if not L.bmp_bimpm_supported(d, H, maxn):          # (any molecule size since round 4: big pairs stage their rows in the workspace)
            raise ValueError(f"BiMPM: unsupported shape (d = {d}, head = {H}, {maxn} rows)")
        out1 = torch.empty(B, 3 * H, dtype=torch.float32, device=X1.device)
        out2 = torch.empty_like(out1)
        nws = L.bmp_bimpm_ws_floats(d, H, maxn, B, 0)
        ws = torch.empty(nws, dtype=torch.float32, device=X1.device)
        check(L.bmp_bimpm_fwd(ptr(X1), ptr(X2), d, H, ptr(w1), ptr(meta["r1"]), ptr(meta["n1"]), ptr(w2), ptr(meta["r2"]),
                              ptr(meta["n2"]), B, maxn, ptr(P), ptr(Q), ptr(R), ptr(out1), ptr(out2), ptr(ws), nws, stream()),
              "bmp_bimpm_fwd")
        ctx.save_for_backward(X1, X2, P, Q, R, w1, w2)
        ctx.meta, ctx.maxn = meta, maxn
        return out1, out2

    @staticmethod
    def backward(ctx, d1, d2):
        L = _lib.lib()
        X1, X2, P, Q, R, w1, w2 = ctx.saved_tensors
        meta, maxn = ctx.meta, ctx.maxn
        d, H, B = X1.shape[1], P.shape[0], meta["B"]
        dX1, dX2 = torch.zeros_like(X1), torch.zeros_like(X2)          # rows of no pair (dead rows) stay zero
        dP, dQ, dR = torch.empty_like(P), torch.empty_like(Q), torch.empty_like(R)
        nws = L.bmp_bimpm_ws_floats(d, H, maxn, B, 1)
        ws = torch.empty(nws, dtype=torch.float32, device=X1.device)
        d1, d2 = d1.contiguous(), d2.contiguous()                      # bound to names: a temporary would be freed before the launch
        check(L.bmp_bimpm_bwd(ptr(d1), ptr(d2), ptr(X1), ptr(X2), d, H, ptr(w1), ptr(meta["r1"]),
                              ptr(meta["n1"]), ptr(w2), ptr(meta["r2"]), ptr(meta["n2"]), B, maxn, ptr(P), ptr(Q), ptr(R),
                              ptr(dX1), ptr(dX2), ptr(dP), ptr(dQ), ptr(dR), ptr(ws), nws, stream()), "bmp_bimpm_bwd")
        return dX1, dX2, dP, dQ, dR, None, None, None, None


class BiMPM(nn.Module):
    def __init__(self, hidden_dim, out_dim, head, with_max_pool=True, with_att_mean=True, with_att_max=True, aggr="sum"):
        super().__init__()
        if not (with_max_pool and with_att_mean and with_att_max):
            raise NotImplementedError("BiMPM is built with all three matchings (its only use, train_binary.py:255-256)")
        if not (aggr == "sum" or getattr(aggr, "__name__", None) == "sum"):
            raise NotImplementedError("BiMPM: aggr must be the sum over atoms (F.sum, train_binary.py:256)")
        std = math.sqrt(2.0 / hidden_dim)                               # initializers.HeNormal, fan_in = hidden_dim (bimpm.py:25-31)
        self.max_pooling_W = nn.Parameter(torch.randn(head, hidden_dim) * std)
        self.att_mean_W = nn.Parameter(torch.randn(head, hidden_dim) * std)
        self.att_max_W = nn.Parameter(torch.randn(head, hidden_dim) * std)
        self.hidden_dim, self.head = hidden_dim, head
        self.out_dim = 3 * head          # what __call__ returns per molecule; the constructor's out_dim is unused (bimpm.py:35)

    def forward(self, atoms_1, g1, atoms_2, g2, **_):
        atoms_1, atoms_2 = as_packed_atoms(atoms_1), as_packed_atoms(atoms_2)
        X1, X2, w1, w2, meta, _joint = pair_rows(atoms_1, atoms_2)
        maxn = max(atoms_1.pb.max_rows_per_mol, atoms_2.pb.max_rows_per_mol)
        return BiMPMFn.apply(X1, X2, self.max_pooling_W, self.att_mean_W, self.att_max_W, w1, w2, meta, maxn)

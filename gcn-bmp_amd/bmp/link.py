"""Link predictors of the reference other than the plain MLP: ``NTN``, ``DistMult``, ``SymMLP``, ``HolE``
(models/mlp.py:48-193; chosen at train_binary.py:165-187, called as ``self.mlp(g1, g2)`` at :102-113).

Each is a pair feature computed from (left_x, right_x) -- bmp_pairfeat_fwd / _bwd -- followed by the relu-MLP
tail (bmp_mlp_fwd / _bwd).  Constructor arguments and parameter names follow the reference; ``in_dim`` /
``fp_dim`` replace Chainer's lazy ``Linear(None, ...)`` shape inference where it is needed.
"""
from __future__ import annotations

import math

import torch
from torch import nn
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr, stream
from .coattention import Bilinear
from .ggnn import Linear
from .mlp import MLPFn

SYM, HOLE, DISTMULT, NTN_KIND = 0, 1, 2, 3


class PairFeatFn(Function):
    """bmp_pairfeat_fwd / _bwd.  Parameters that a kind does not have are passed as None."""

    @staticmethod
    def forward(ctx, kind, K, x1, x2, W, V1, V2, b):
        L = _lib.lib()
        x1, x2 = x1.contiguous(), x2.contiguous()
        W, V1, V2, b = (None if t is None else t.contiguous() for t in (W, V1, V2, b))
        B, d1 = x1.shape
        d2 = x2.shape[1]
        out = torch.empty(B, L.bmp_pairfeat_cols(kind, d1, K), dtype=torch.float32, device=x1.device)
        check(L.bmp_pairfeat_fwd(kind, ptr(x1), ptr(x2), B, d1, d2, ptr(W), ptr(V1), ptr(V2), ptr(b), K, ptr(out), stream()),
              "bmp_pairfeat_fwd")
        ctx.save_for_backward(x1, x2, *(t for t in (W, V1, V2) if t is not None))
        ctx.meta = (kind, K, W is not None, V1 is not None, V2 is not None, b is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        kind, K, hW, hV1, hV2, hb = ctx.meta
        sv = list(ctx.saved_tensors)
        x1, x2 = sv[0], sv[1]
        rest = sv[2:]
        W = rest.pop(0) if hW else None
        V1 = rest.pop(0) if hV1 else None
        V2 = rest.pop(0) if hV2 else None
        B, d1 = x1.shape
        d2 = x2.shape[1]
        dout = dout.contiguous()
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        dW = torch.empty_like(W) if hW else None
        dV1 = torch.empty_like(V1) if hV1 else None
        dV2 = torch.empty_like(V2) if hV2 else None
        db = torch.empty(K, dtype=torch.float32, device=x1.device) if hb else None
        check(L.bmp_pairfeat_bwd(kind, ptr(dout), ptr(x1), ptr(x2), B, d1, d2, ptr(W), ptr(V1), ptr(V2), K, ptr(dx1),
                                 ptr(dx2), ptr(dW), ptr(dV1), ptr(dV2), ptr(db), stream()), "bmp_pairfeat_bwd")
        return None, None, dx1, dx2, dW, dV1, dV2, db


def _tail(layers, l_out, h):
    ls = list(layers) + [l_out]
    return MLPFn.apply(None, h, None, None, None, *[l.W for l in ls], *[l.b for l in ls])


def _check_act(activation):
    if activation not in (torch.relu, torch.nn.functional.relu):
        raise NotImplementedError("only the reference's default activation (relu) is supported")


class NTN(nn.Module):
    """models/mlp.py:48-72."""
    is_link_predictor = True

    def __init__(self, left_dim, right_dim, out_dim, ntn_out_dim=8, hidden_dims=(16,), activation=torch.relu):
        super().__init__()
        _check_act(activation)
        self.ntn_layer = Bilinear(left_dim, right_dim, ntn_out_dim)
        dims = [ntn_out_dim] + list(hidden_dims)
        self.mlp_layers = nn.ModuleList([Linear(dims[i], dims[i + 1]) for i in range(len(hidden_dims))])
        self.l_out = Linear(dims[-1], out_dim)
        self.left_dim, self.right_dim, self.out_dim, self.hidden_dims = left_dim, right_dim, out_dim, hidden_dims

    def forward(self, left_x, right_x):
        E = self.ntn_layer
        h = PairFeatFn.apply(NTN_KIND, E.b.shape[0], left_x, right_x, E.W, E.V1, E.V2, E.b)
        return _tail(self.mlp_layers, self.l_out, h)


class BilinearDiag(nn.Module):
    """models/mlp.py:153-193: W [out x left], default initialiser (LeCunNormal over `left`)."""

    def __init__(self, left_size, right_size, out_size):
        super().__init__()
        assert left_size == right_size                                         # :160
        self.W = nn.Parameter(torch.randn(out_size, left_size) / math.sqrt(left_size))


class DistMult(nn.Module):
    """models/mlp.py:75-93."""
    is_link_predictor = True

    def __init__(self, left_dim, right_dim, out_dim, dm_out_dim=8, hidden_dims=(16,), activation=torch.relu):
        super().__init__()
        _check_act(activation)
        self.dm_layer = BilinearDiag(left_dim, right_dim, dm_out_dim)
        dims = [dm_out_dim] + list(hidden_dims)
        self.mlp_layers = nn.ModuleList([Linear(dims[i], dims[i + 1]) for i in range(len(hidden_dims))])
        self.l_out = Linear(dims[-1], out_dim)

    def forward(self, left_x, right_x):
        W = self.dm_layer.W
        h = PairFeatFn.apply(DISTMULT, W.shape[0], left_x, right_x, W, None, None, None)
        return _tail(self.mlp_layers, self.l_out, h)


class SymMLP(nn.Module):
    """models/mlp.py:96-110.  ``fp_dim`` = width of left_x / right_x."""
    is_link_predictor = True

    def __init__(self, out_dim, hidden_dims=(32, 16), activation=torch.relu, fp_dim=None):
        super().__init__()
        _check_act(activation)
        dims = [None if fp_dim is None else 2 * fp_dim] + list(hidden_dims)          # None: lazy, models/mlp.py:99-102
        self.layers = nn.ModuleList([Linear(dims[i], dims[i + 1]) for i in range(len(hidden_dims))])
        self.l_out = Linear(dims[-1], out_dim)

    def materialize_input(self, fp_dim: int) -> None:
        (self.layers[0] if len(self.layers) else self.l_out).materialize(2 * fp_dim)

    def forward(self, left_x, right_x):
        self.materialize_input(left_x.shape[-1])
        h = PairFeatFn.apply(SYM, 0, left_x, right_x, None, None, None, None)
        return _tail(self.layers, self.l_out, h)


class HolE(nn.Module):
    """models/mlp.py:113-151.  ``fp_dim`` = width of left_x / right_x."""
    is_link_predictor = True

    def __init__(self, out_dim, hidden_dims=(32, 16), activation=torch.relu, fp_dim=None):
        super().__init__()
        _check_act(activation)
        dims = [fp_dim] + list(hidden_dims)                                           # None: lazy, models/mlp.py:116-119
        self.layers = nn.ModuleList([Linear(dims[i], dims[i + 1]) for i in range(len(hidden_dims))])
        self.l_out = Linear(dims[-1], out_dim)

    def materialize_input(self, fp_dim: int) -> None:
        (self.layers[0] if len(self.layers) else self.l_out).materialize(fp_dim)

    def forward(self, left_x, right_x):
        self.materialize_input(left_x.shape[-1])
        return _tail(self.layers, self.l_out, self.circular_correlation(left_x, right_x))

    def circular_correlation(self, left_x, right_x):
        """models/mlp.py:126-151 (there through fft/ifft; here the direct sum, same values)."""
        return PairFeatFn.apply(HOLE, 0, left_x, right_x, None, None, None, None)

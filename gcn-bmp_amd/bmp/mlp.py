"""Link predictor ``MLP`` with the reference's signature (models/mlp.py:20-45) and the pair loss.

On the device the whole MLP is one forward launch and two backward launches (bmp_mlp_fwd / bmp_mlp_bwd) and the
sigmoid cross entropy one launch each way, instead of ~40 framework launches of a few microseconds.  Host tensors
(the CPU tests of the data-parallel plumbing) take the plain torch ops.
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn
from torch.autograd import Function

from .ggnn import Linear

_MAXL, _MAXW, _MAXIN = 4, 64, 1024


def _parr(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class MLPFn(Function):
    """relu-MLP on [x1 | x2].  ``G`` (a dict name -> tensor from bmp.plan) receives the weight gradients instead of
    autograd when given; ``tape`` is the plan's dummy differentiable input."""

    @staticmethod
    def forward(ctx, tape, x1, x2, G, state, *wb):
        from . import _lib
        from ._lib import check, ptr, stream
        L = _lib.lib()
        nl = len(wb) // 2
        Ws = [w.contiguous() for w in wb[:nl]]
        bs = [None if b is None else b.contiguous() for b in wb[nl:]]
        x1 = x1.contiguous()
        x2 = None if x2 is None else x2.contiguous()
        B, d1 = x1.shape
        d2 = 0 if x2 is None else x2.shape[1]
        dims = [d1 + d2] + [w.shape[0] for w in Ws]
        cd = (ctypes.c_int * len(dims))(*dims)
        acts = [torch.empty(B, n, dtype=torch.float32, device=x1.device) for n in dims[1:]]
        check(L.bmp_mlp_fwd(ptr(x1), d1, ptr(x2), d2, B, nl, cd, _parr(Ws), _parr(bs), _parr(acts), stream()), "bmp_mlp_fwd")
        ctx.save_for_backward(x1, *( [x2] if x2 is not None else [] ), *Ws, *acts)
        ctx.meta = (nl, dims, x2 is not None, [b is not None for b in bs], G, state)
        return acts[-1]

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        from ._lib import check, ptr, stream
        L = _lib.lib()
        nl, dims, has2, has_b, G, state = ctx.meta
        sv = list(ctx.saved_tensors)
        x1 = sv.pop(0)
        x2 = sv.pop(0) if has2 else None
        Ws, acts = sv[:nl], sv[nl:]
        B, d1 = x1.shape
        d2 = dims[0] - d1
        dev = x1.device
        dy = dy.contiguous()
        cd = (ctypes.c_int * len(dims))(*dims)
        dx1 = torch.empty_like(x1)
        dx2 = torch.empty_like(x2) if has2 else None
        if G is not None:
            dW = [G[f"dW{l}"] for l in range(nl)]
            db = [G[f"db{l}"] if has_b[l] else None for l in range(nl)]
        else:
            dW = [torch.empty_like(w) for w in Ws]
            db = [torch.empty(w.shape[0], dtype=torch.float32, device=dev) if has_b[l] else None for l, w in enumerate(Ws)]
        nws = L.bmp_mlp_bwd_ws_floats(B, nl, cd)
        ws = torch.empty(max(nws, 4), dtype=torch.float32, device=dev)
        from .functional import _side_handle
        # the weight-gradient partials and their fold beside the chain (planned path): that launch reads dy, the inputs and the
        # saved activations, which therefore stay alive until the streams have joined
        st_w = _side_handle(state, (ws, dy, x1, x2, *acts)) if G is not None else None
        check(L.bmp_mlp_bwd(ptr(dy), ptr(x1), d1, ptr(x2), d2, B, nl, cd, _parr(Ws), _parr(acts), ptr(dx1), ptr(dx2),
                            _parr(dW), _parr(db), ptr(ws), nws, stream(), st_w), "bmp_mlp_bwd")
        if G is not None:
            return (None, dx1, dx2, None, None) + (None,) * (2 * nl)
        return (None, dx1, dx2, None, None) + tuple(dW) + tuple(db)


_TICKETS = {}


def _ticket(device, stream_handle) -> torch.Tensor:
    """The zero word of bmp_mlp_sce_fwdbwd (zero before every launch, put back to zero by the launch): one per (device,
    STREAM).  Launches on one stream run one behind the other, so they can share the word; two head launches in flight on
    different streams of a device (a second model, training beside evaluation) each fold on their own."""
    k = (device.type, device.index, int(stream_handle or 0))
    if k not in _TICKETS:
        _TICKETS[k] = torch.zeros(1, dtype=torch.int32, device=device)
    return _TICKETS[k]


class MLPLossFn(Function):
    """relu-MLP on [x1 | x2] AND the mean sigmoid cross entropy of its logits against ``t`` -- what the reference's
    Classifier(predictor, lossfun=F.sigmoid_cross_entropy) computes around the link predictor (train_ddi_modify.py:284-286) -- as
    ONE forward launch that also takes the loss gradient back to the input rows (bmp_mlp_sce_fwdbwd); the backward scales those
    rows by the gradient that arrives at the loss and folds the weight gradients beside the chain (bmp_mlp_bwd_w).
    Returns (loss, logits); the logits are not differentiable here (the loss is the only way back)."""

    @staticmethod
    def forward(ctx, tape, x1, x2, t, G, state, consumer_scales, *wb):
        from . import _lib
        from ._lib import check, ptr, stream
        L = _lib.lib()
        nl = len(wb) // 2
        Ws = [w.contiguous() for w in wb[:nl]]
        bs = [None if b is None else b.contiguous() for b in wb[nl:]]
        x1 = x1.contiguous()
        x2 = None if x2 is None else x2.contiguous()
        B, d1 = x1.shape
        d2 = 0 if x2 is None else x2.shape[1]
        dims = [d1 + d2] + [w.shape[0] for w in Ws]
        dev = x1.device
        t = t.to(torch.int32).contiguous()
        if t.numel() != B * dims[-1]:
            raise ValueError(f"labels: {tuple(t.shape)} for {B} rows of {dims[-1]} logits")
        cd = (ctypes.c_int * len(dims))(*dims)
        f = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)
        acts = [f(B, n) for n in dims[1:]]
        dy, dx = f(B, dims[-1]), f(B * (d1 + d2))
        dx1, dx2 = dx[:B * d1].view(B, d1), (dx[B * d1:].view(B, d2) if d2 else None)
        loss, sums, part = f(()), f(2), f(max(int(L.bmp_mlp_sce_ws_floats(B)), 1))
        st = stream()
        check(L.bmp_mlp_sce_fwdbwd(ptr(x1), d1, ptr(x2), d2, B, nl, cd, _parr(Ws), _parr(bs), _parr(acts), ptr(t), ptr(dy),
                                   ptr(dx1), ptr(dx2), ptr(loss), ptr(sums), ptr(part), ptr(_ticket(dev, st.value)), st),
              "bmp_mlp_sce_fwdbwd")
        ctx.kept = (x1, x2, Ws, acts, dy, dx, dx1, dx2)
        ctx.meta = (nl, dims, [b is not None for b in bs], G, state)
        ctx.mark_non_differentiable(acts[-1])
        ctx.set_materialize_grads(False)
        ctx.consumer_scales = consumer_scales
        return loss, acts[-1]

    @staticmethod
    def backward(ctx, gloss, _gy):
        from . import _lib
        from ._lib import check, ptr, stream
        L = _lib.lib()
        nl, dims, has_b, G, state = ctx.meta
        x1, x2, Ws, acts, dy, dx, dx1, dx2 = ctx.kept
        B, d1 = x1.shape
        d2 = dims[0] - d1
        dev = x1.device
        cd = (ctypes.c_int * len(dims))(*dims)
        if G is not None:
            dW = [G[f"dW{l}"] for l in range(nl)]
            db = [G[f"db{l}"] if has_b[l] else None for l in range(nl)]
        else:
            dW = [torch.empty_like(w) for w in Ws]
            db = [torch.empty(w.shape[0], dtype=torch.float32, device=dev) if has_b[l] else None for l, w in enumerate(Ws)]
        nws = L.bmp_mlp_bwd_ws_floats(B, nl, cd)
        ws = torch.empty(max(nws, 4), dtype=torch.float32, device=dev)
        if gloss is None:                                    # the loss does not reach the root of this backward
            gloss = torch.zeros(1, dtype=torch.float32, device=dev)
        gloss = gloss.contiguous().to(torch.float32).reshape(1)
        if ctx.consumer_scales and state is not None:
            state["head_gscale"] = (gloss, dx1, dx2)         # PNieFn.backward (the node both input blocks come from) multiplies
            gx = dx
        else:
            gx = dx * gloss                                  # one launch over both input blocks
        def weight_grads():
            st_w = None
            if G is not None:
                from .functional import _side_handle
                st_w = _side_handle(state, (ws, dy, x1, x2, gloss, *acts))
            if st_w is not None:
                state["side"].stream.wait_stream(torch.cuda.current_stream())
            check(L.bmp_mlp_bwd_w(ptr(dy), ptr(x1), d1, ptr(x2), d2, B, nl, cd, _parr(Ws), _parr(acts), _parr(dW), _parr(db),
                                  ptr(gloss), ptr(ws), nws, st_w if st_w is not None else stream()), "bmp_mlp_bwd_w")
        if ctx.consumer_scales and state is not None:
            # behind the co-attention's backward launches in the weight-gradient stream's queue (flushed by PNieFn.backward):
            # in front of them, these partials -- nobody's input -- delayed the pair kernels' largest size class, which the
            # backward chain waits for
            state.setdefault("deferred_bwd", []).append(weight_grads)
        else:
            weight_grads()
        gx1 = gx[:B * d1].view(B, d1)
        gx2 = gx[B * d1:].view(B, d2) if d2 else None
        if G is not None:
            return (None, gx1, gx2, None, None, None, None) + (None,) * (2 * nl)
        return (None, gx1, gx2, None, None, None, None) + tuple(dW) + tuple(db)


class SCEFn(Function):
    """chainer.functions.sigmoid_cross_entropy (train_ddi_modify.py:285)."""

    @staticmethod
    def forward(ctx, y, t):
        from . import _lib
        from ._lib import check, ptr, stream
        y = y.contiguous()
        t = t.to(torch.int32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=y.device)
        sums = torch.empty(2, dtype=torch.float32, device=y.device)         # numerator | count
        check(_lib.lib().bmp_sce_fwd(ptr(y), ptr(t), y.numel(), ptr(loss), ptr(sums), stream()), "bmp_sce_fwd")
        ctx.save_for_backward(y, t, sums)
        return loss

    @staticmethod
    def backward(ctx, gout):
        from . import _lib
        from ._lib import check, ptr, stream
        y, t, sums = ctx.saved_tensors
        dy = torch.empty_like(y)
        gout = gout.contiguous().to(torch.float32)
        check(_lib.lib().bmp_sce_bwd(ptr(y), ptr(t), y.numel(), ptr(sums), ptr(gout), ptr(dy), stream()), "bmp_sce_bwd")
        return dy, None


def sigmoid_cross_entropy(y: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Mean over the elements with t != -1 of softplus(y) - t*y."""
    if y.is_cuda:
        return SCEFn.apply(y, t)
    tf = t.to(y.dtype)
    mask = t != -1
    loss = torch.nn.functional.softplus(y) - tf * y
    loss = torch.where(mask, loss, torch.zeros_like(loss))
    return loss.sum() / mask.sum().clamp(min=1).to(y.dtype)


class MLP(nn.Module):
    is_link_predictor = True

    def __init__(self, out_dim, hidden_dims=(32, 16), activation=torch.relu, in_dim=None):
        """models/mlp.py:32-38.  ``in_dim=None`` (the reference's call form, train_ddi_modify.py:136) leaves the first
        layer's input width open like Chainer's ``Linear(None, ...)``: it is fixed by ``materialize_input`` -- called by
        GraphConvPredictorForPair from the encoder's / co-attention's out_dim, or at the first forward."""
        super().__init__()
        dims = [in_dim] + list(hidden_dims)
        self.layers = nn.ModuleList([Linear(dims[i], dims[i + 1]) for i in range(len(hidden_dims))])
        self.l_out = Linear(dims[-1], out_dim)
        self.activation = activation
        self.in_dim, self.out_dim = in_dim, out_dim

    def materialize_input(self, fp_dim: int) -> None:
        """Fix the input width from the width of ONE molecule vector (the MLP sees [g1 | g2])."""
        (self.layers[0] if len(self.layers) else self.l_out).materialize(2 * fp_dim)
        self.in_dim = 2 * fp_dim

    def _linears(self):
        return list(self.layers) + [self.l_out]

    def _kernel_ok(self) -> bool:
        ls = self._linears()
        return (self.activation in (torch.relu, torch.nn.functional.relu) and len(ls) <= _MAXL and self.in_dim is not None
                and self.in_dim <= _MAXIN
                and all(l.out_size <= _MAXW for l in ls))

    # ---- layout plan protocol (bmp/plan.py): identity layouts, the kernels write the gradients in place ----
    def plannable(self) -> bool:
        return self._kernel_ok()

    def primary_layouts(self):
        out = {}
        for k, l in enumerate(self._linears()):
            out[f"W{k}"] = l.W
            out[f"b{k}"] = l.b
        return out

    prepared_layouts = primary_layouts

    def gk_spec(self):
        spec = {}
        for k, l in enumerate(self._linears()):
            spec[f"dW{k}"] = tuple(l.W.shape)
            spec[f"db{k}"] = tuple(l.b.shape)
        return spec

    def primary_grads(self, gk):
        out = {}
        for k in range(len(self._linears())):
            out[f"W{k}"] = [gk[f"dW{k}"]]
            out[f"b{k}"] = [gk[f"db{k}"]]
        return out

    def forward(self, x, x2=None):
        """models/mlp.py:40-45.  ``x2``: optional second half of the input row ([x | x2] without the concatenation)."""
        if self.in_dim is None:                                 # Chainer's lazy Linear: fixed by the first input
            width = x.shape[-1] + (0 if x2 is None else x2.shape[-1])
            (self.layers[0] if len(self.layers) else self.l_out).materialize(width)
            self.in_dim = width
        elif x.shape[-1] + (0 if x2 is None else x2.shape[-1]) != self.in_dim:
            raise ValueError(f"MLP was built for {self.in_dim} input features, got "
                             f"{x.shape[-1] + (0 if x2 is None else x2.shape[-1])}")
        ls = self._linears()
        if x.is_cuda and self._kernel_ok():
            fast = getattr(self, "_fast", None)
            if fast is not None:
                P, G, state, tape = fast
                wb = [P[f"W{k}"] for k in range(len(ls))] + [P[f"b{k}"] for k in range(len(ls))]
                return MLPFn.apply(tape, x, x2, G, state, *wb)
            return MLPFn.apply(None, x, x2, None, None, *[l.W for l in ls], *[l.b for l in ls])
        h = x if x2 is None else torch.cat((x, x2), dim=-1)
        for l in self.layers:                                   # models/mlp.py:42-43
            h = self.activation(torch.nn.functional.linear(h, l.W, l.b))
        return torch.nn.functional.linear(h, self.l_out.W, self.l_out.b)

    def forward_loss(self, x, x2, t):
        """(mean sigmoid cross entropy of the logits against ``t``, logits): the reference's Classifier around this link
        predictor (train_ddi_modify.py:284-286).  On the device, with gradients on, one launch each way (MLPLossFn); otherwise
        ``forward`` followed by ``sigmoid_cross_entropy``."""
        ls = self._linears()
        dims = [self.in_dim] + [l.out_size for l in ls]
        # (the launch keeps every layer's weights in LDS beside 48 KB of row buffers)
        fits = self.in_dim is not None and 4 * (dims[0] * (dims[1] + 1) + sum(dims[k + 1] * (dims[k] + 1) for k in range(1, len(ls)))) \
            <= 160 * 1024 - 49152
        if x.is_cuda and torch.is_grad_enabled() and fits and self._kernel_ok() and \
                x.shape[-1] + (0 if x2 is None else x2.shape[-1]) == self.in_dim:
            fast = getattr(self, "_fast", None)
            if fast is not None:
                P, G, state, tape = fast
                wb = [P[f"W{k}"] for k in range(len(ls))] + [P[f"b{k}"] for k in range(len(ls))]
                # both input blocks straight out of ONE planned co-attention node: that node's backward takes the factor that
                # arrives at the loss (a device scalar) into its pair kernels, instead of a launch that scales the rows here
                fn1, fn2 = getattr(x, "grad_fn", None), getattr(x2, "grad_fn", None)
                # (the node says so itself -- PNieFn.forward sets ``takes_head_gscale`` -- and must hold the SAME plan state this
                #  module hands the factor through; FlatAdam.collect_grads checks that the factor was consumed)
                direct = (fn1 is not None and fn1 is fn2 and state is not None and getattr(fn1, "takes_head_gscale", False)
                          and getattr(fn1, "state", None) is state)
                loss, y = MLPLossFn.apply(tape, x, x2, t, G, state, direct, *wb)
            else:
                loss, y = MLPLossFn.apply(None, x, x2, t, None, None, False, *[l.W for l in ls], *[l.b for l in ls])
            return loss, y
        y = self.forward(x, x2)
        return sigmoid_cross_entropy(y, t), y

"""torch.autograd.Functions over the C ABI: one Function per reference operator.

Every Function's forward/backward is a 1:1 call into libbmp_hip (include/bmp.h).  torch is
used for device memory, the stream and the autograd graph only; weight layout changes
(reference [out x in] -> kernel K-major) are differentiable torch views done by the callers
in bmp/ggnn.py etc., so parameter gradients come back in the reference layout.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr, stream, require_rows
from .packed import PackedMolBatch

ACT = {"identity": 0, None: 0, "sigmoid": 1, "tanh": 2, "relu": 3}


def _ws(nfloats: int, device) -> torch.Tensor:
    return torch.empty(max(int(nfloats), 4), dtype=torch.float32, device=device)


def _check_pb(pb: PackedMolBatch, x: torch.Tensor) -> None:
    L = _lib.lib()
    if pb.R != L.bmp_tile_rows():
        raise ValueError(f"packed batch tile size R={pb.R} != library tile rows {L.bmp_tile_rows()}")
    if x.shape[0] != pb.n_rows:
        raise ValueError(f"row tensor has {x.shape[0]} rows, packed batch has {pb.n_rows}")


class EmbedFn(Function):
    """EmbedAtomID (models/ggnn.py:85,603)."""

    @staticmethod
    def forward(ctx, W, ids):
        L = _lib.lib()
        if W.dtype != torch.float32 or not W.is_cuda or not W.is_contiguous():
            raise ValueError("embed: W must be a contiguous float32 CUDA tensor")
        if ids.dtype != torch.int32 or not ids.is_contiguous():
            raise ValueError("embed: ids must be contiguous int32")
        V, d = W.shape
        N = ids.numel()
        out = torch.empty(N, d, dtype=torch.float32, device=W.device)
        check(L.bmp_embed_fwd(ptr(ids), ptr(W), N, d, V, ptr(out), stream()), "bmp_embed_fwd")
        ctx.save_for_backward(ids)
        ctx.V = V
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        (ids,) = ctx.saved_tensors
        dout = dout.contiguous()
        N, d = dout.shape
        dW = torch.empty(ctx.V, d, dtype=torch.float32, device=dout.device)
        nws = L.bmp_embed_bwd_ws_floats(N, d, ctx.V)
        ws = _ws(nws, dout.device)
        check(L.bmp_embed_bwd(ptr(ids), ptr(dout), N, d, ctx.V, ptr(dW), ptr(ws), nws, stream()), "bmp_embed_bwd")
        return dW, None


class MsgFn(Function):
    """Message / RelGCN layer (models/ggnn.py:215-243; models/update/relgcn_update.py:24-44).
    WT [4*d_in x d_out], bE [4 x d_out], optional self connection WsT [d_in x d_out], bs [d_out]."""

    @staticmethod
    def forward(ctx, x, WT, bE, WsT, bs, pb, act):
        L = _lib.lib()
        require_rows(x, "msg: x")
        _check_pb(pb, x)
        d_in = x.shape[1]
        d_out = WT.shape[1]
        if WT.shape[0] != 4 * d_in or tuple(bE.shape) != (4, d_out):
            raise ValueError("msg: weight shapes do not match x")
        WT = WT.contiguous(); bE = bE.contiguous()
        WsT = None if WsT is None else WsT.contiguous()
        bs = None if bs is None else bs.contiguous()
        N = x.shape[0]
        agg = torch.empty(N, 4 * d_in, dtype=torch.float32, device=x.device)
        wdeg = torch.empty(N, 4, dtype=torch.float32, device=x.device)
        out = torch.empty(N, d_out, dtype=torch.float32, device=x.device)
        check(L.bmp_msg_fwd(ptr(x), d_in, pb.n_tiles, d_in, d_out, ptr(pb.csr_ptr), ptr(pb.csr_col), ptr(pb.csr_val),
                            ptr(WT), ptr(bE), ptr(WsT), ptr(bs), act, ptr(agg), ptr(wdeg), ptr(out), d_out, stream()),
              "bmp_msg_fwd")
        ctx.save_for_backward(x, WT, WsT if WsT is not None else torch.empty(0), agg, wdeg, out)
        ctx.pb, ctx.act, ctx.has_self, ctx.has_bs = pb, act, WsT is not None, bs is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x, WT, WsT, agg, wdeg, out = ctx.saved_tensors
        pb = ctx.pb
        dout = dout.contiguous()
        N, d_out = dout.shape
        d_in = x.shape[1]
        dev = x.device
        Wnat = WT.t().contiguous()
        Ws = WsT.t().contiguous() if ctx.has_self else None
        dx = torch.empty(N, d_in, dtype=torch.float32, device=dev)
        dWT = torch.empty_like(WT)
        dbE = torch.empty(4, d_out, dtype=torch.float32, device=dev)
        dWsT = torch.empty(d_in, d_out, dtype=torch.float32, device=dev) if ctx.has_self else None
        dbs = torch.empty(d_out, dtype=torch.float32, device=dev) if ctx.has_self else None
        nws = L.bmp_msg_bwd_ws_floats(pb.n_tiles, d_in, d_out)
        ws = _ws(nws, dev)
        trf, trc = type_rows(pb, forward=True)
        check(L.bmp_msg_bwd(ptr(dout), d_out, ptr(out), d_out, ctx.act, ptr(x), d_in, pb.n_tiles, d_in, d_out,
                            ptr(pb.csrT_ptr), ptr(pb.csrT_col), ptr(pb.csrT_val), ptr(Wnat), ptr(Ws), ptr(agg), ptr(wdeg),
                            ptr(dx), ptr(dWT), ptr(dbE), ptr(dWsT), ptr(dbs), 0, ptr(trf), ptr(trc), ptr(ws), nws, stream(), None), "bmp_msg_bwd")
        return dx, dWT, dbE, dWsT, (dbs if ctx.has_bs else None), None, None


class GRUFn(Function):
    """GRU node update (chainer StatefulGRU at models/ggnn.py:132,254-262).
    AT [2d x 3d], UcT [d x d], b [3d] in the folded kernel layout (see bmp/ggnn.py)."""

    @staticmethod
    def forward(ctx, h, m, AT, UcT, b, pb, first):
        L = _lib.lib()
        require_rows(h, "gru: h")
        require_rows(m, "gru: m", h.shape[1])
        _check_pb(pb, h)
        N, d = h.shape
        if tuple(AT.shape) != (2 * d, 3 * d) or tuple(UcT.shape) != (d, d) or tuple(b.shape) != (3 * d,):
            raise ValueError("gru: weight shapes do not match h")
        AT = AT.contiguous(); UcT = UcT.contiguous(); b = b.contiguous()
        rz = torch.empty(N, 2 * d, dtype=torch.float32, device=h.device)
        c = torch.empty(N, d, dtype=torch.float32, device=h.device)
        hout = torch.empty(N, d, dtype=torch.float32, device=h.device)
        check(L.bmp_gru_fwd(ptr(h), ptr(m), pb.n_tiles, d, int(first), ptr(AT), ptr(UcT), ptr(b), ptr(rz), ptr(c),
                            ptr(hout), stream()), "bmp_gru_fwd")
        ctx.save_for_backward(h, m, AT, UcT, rz, c)
        ctx.pb, ctx.first = pb, int(first)
        return hout

    @staticmethod
    def backward(ctx, dhout):
        L = _lib.lib()
        h, m, AT, UcT, rz, c = ctx.saved_tensors
        pb = ctx.pb
        dhout = dhout.contiguous()
        N, d = h.shape
        dev = h.device
        A = AT.t().contiguous()
        Uc = UcT.t().contiguous()
        dh = torch.empty_like(h); dm = torch.empty_like(m)
        dAT = torch.empty_like(AT); dUcT = torch.empty_like(UcT)
        db = torch.empty(3 * d, dtype=torch.float32, device=dev)
        nws = L.bmp_gru_bwd_ws_floats(pb.n_tiles, d)
        ws = _ws(nws, dev)
        check(L.bmp_gru_bwd(ptr(dhout), ptr(h), ptr(m), ptr(rz), ptr(c), pb.n_tiles, d, ctx.first, ptr(A), ptr(Uc),
                            ptr(dh), ptr(dm), ptr(dAT), ptr(dUcT), ptr(db), 0, ptr(ws), nws, stream(), None), "bmp_gru_bwd")
        return dh, dm, dAT, dUcT, db, None, None


class GRUStateFn(Function):
    """GRU update with its state apart from its input (bmp_gru_state_fwd / _bwd): what F.dropout on the step output
    (models/ggnn.py:626-627) makes of the later calls.  WT [2d x 3d], UrzT [d x 2d], UcT [d x d], b [3d]."""

    @staticmethod
    def forward(ctx, hd, m, s, WT, UrzT, UcT, b, pb):
        L = _lib.lib()
        require_rows(hd, "gru: hd")
        require_rows(m, "gru: m", hd.shape[1])
        require_rows(s, "gru: s", hd.shape[1])
        _check_pb(pb, hd)
        N, d = hd.shape
        if tuple(WT.shape) != (2 * d, 3 * d) or tuple(UrzT.shape) != (d, 2 * d) or tuple(UcT.shape) != (d, d) or tuple(b.shape) != (3 * d,):
            raise ValueError("gru: weight shapes do not match hd")
        WT, UrzT, UcT, b = WT.contiguous(), UrzT.contiguous(), UcT.contiguous(), b.contiguous()
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=hd.device)
        rz, c, sout = f(N, 2 * d), f(N, d), f(N, d)
        check(L.bmp_gru_state_fwd(ptr(hd), ptr(m), ptr(s), pb.n_tiles, d, ptr(WT), ptr(UrzT), ptr(UcT), ptr(b), ptr(rz), ptr(c),
                                  ptr(sout), stream()), "bmp_gru_state_fwd")
        ctx.save_for_backward(hd, m, s, WT, UrzT, UcT, rz, c)
        ctx.pb = pb
        return sout

    @staticmethod
    def backward(ctx, dsout):
        L = _lib.lib()
        hd, m, s, WT, UrzT, UcT, rz, c = ctx.saved_tensors
        pb = ctx.pb
        N, d = hd.shape
        dev = hd.device
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        dhd, dm, ds = f(N, d), f(N, d), f(N, d)
        dWT, dUrzT, dUcT, db = f(2 * d, 3 * d), f(d, 2 * d), f(d, d), f(3 * d)
        nws = L.bmp_gru_state_bwd_ws_floats(pb.n_tiles, d)
        ws = _ws(nws, dev)
        dsout = dsout.contiguous()
        A, Urz, Uc = WT.t().contiguous(), UrzT.t().contiguous(), UcT.t().contiguous()      # kept alive across the launch
        check(L.bmp_gru_state_bwd(ptr(dsout), ptr(hd), ptr(m), ptr(s), ptr(rz), ptr(c), pb.n_tiles, d,
                                  ptr(A), ptr(Urz), ptr(Uc), ptr(dhd), ptr(dm),
                                  ptr(ds), ptr(dWT), ptr(dUrzT), ptr(dUcT), ptr(db), ptr(ws), nws, stream()), "bmp_gru_state_bwd")
        return dhd, dm, ds, dWT, dUrzT, dUcT, db, None


# The step's weight-gradient launches walk the rows that have a bond of the type only (PackedMolBatch.type_rows_T; DESIGN.md
# section 3a round 4).  BMP_WGRAD_LISTS=0: every row, as before (A/B switch of bench.py and the tests).
_WGRAD_LISTS = os.environ.get("BMP_WGRAD_LISTS", "1") != "0"


def type_rows(pb, forward: bool = False):
    """(idx, cnt) tensors of the batch's transposed (default) or forward CSR, or (None, None)."""
    if not _WGRAD_LISTS or pb is None:
        return None, None
    tr = pb.type_rows_T(forward=forward)
    return tr if tr is not None else (None, None)


def step_lists(pb, N: int, d: int):
    """(idx, cnt, skip) for a fused step / layer backward: the batch's transposed-CSR row lists when the weight-gradient launch
    will read gda's per-type blocks through them (bmp_step_wgrad_lists_used) -- the backward tile kernel then skips the
    blocks of rows without a bond of the type (skip = 1) --, else (None, None, 0)."""
    tri, trc = type_rows(pb)
    if tri is None or not _lib.lib().bmp_step_wgrad_lists_used(int(N), int(d)):
        return None, None, 0
    return tri, trc, 1


def _rel_mt(pb):
    """Tile table for the RelGCN layer kernels: they do not clear dead blocks, so a fixed-stride table (pb.tile_stride, the
    fixed-shape batch) is not handed to them -- its tiles run whole."""
    return (None, None) if pb.tile_stride else (pb.mt_row0, pb.mt_nblk)


def step_supported(d: int) -> bool:
    return bool(_lib.lib().bmp_ggnn_step_supported(int(d)))


def pack_k4(W: torch.Tensor) -> torch.Tensor:
    """K-major [K x N] weight -> the fused kernels' B-operand layout [K/4][N][4]: a lane's four
    consecutive k values of one output column are one 16-byte load, 64 lanes are 1 KiB contiguous."""
    K, N = W.shape
    return W.detach().reshape(K // 4, 4, N).permute(0, 2, 1).contiguous()


def _cached(cache, tag, src, make):
    """The packed copy of ``src`` in the per-call ``cache``, keyed by the storage address.  The entry holds ``src`` itself:
    under no_grad nothing else keeps a per-step weight tensor alive, and a freed block handed to the next layer's weights
    would otherwise answer with the previous layer's packed copy."""
    if cache is None:
        return make()
    key = (tag, src.data_ptr())
    if key not in cache:
        cache[key] = (make(), src)
    return cache[key][0]


class GGNNStepFn(Function):
    """One whole propagation step (message + GRU) as ONE fused kernel per tile
    (models/ggnn.py:215-263).  Same weight layouts as MsgFn / GRUFn; ``cache`` (a dict that lives
    for one encoder call) shares the packed weight copies between the steps of tied layers."""

    @staticmethod
    def forward(ctx, h, WT, bE, AT, UcT, b, pb, first, cache=None):
        L = _lib.lib()
        require_rows(h, "step: h")
        _check_pb(pb, h)
        N, d = h.shape
        if tuple(WT.shape) != (4 * d, d) or tuple(bE.shape) != (4, d) or tuple(AT.shape) != (2 * d, 3 * d) \
                or tuple(UcT.shape) != (d, d) or tuple(b.shape) != (3 * d,):
            raise ValueError("step: weight shapes do not match h")
        bE, b = bE.contiguous(), b.contiguous()
        WTp = _cached(cache, "f", WT, lambda: pack_k4(WT))
        ATp = _cached(cache, "f", AT, lambda: pack_k4(AT))
        UcTp = _cached(cache, "f", UcT, lambda: pack_k4(UcT))
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=h.device)
        m, rz, c, hout = f(N, d), f(N, 2 * d), f(N, d), f(N, d)
        check(L.bmp_ggnn_step_fwd(ptr(h), 0, pb.n_mtiles, d, int(first), ptr(pb.csr_ptr), ptr(pb.csr_col), ptr(pb.csr_val),
                                  ptr(WTp), ptr(bE), ptr(ATp), ptr(UcTp), ptr(b), ptr(m), ptr(rz), ptr(c), ptr(hout),
                                  ptr(pb.mt_row0), ptr(pb.mt_nblk), pb.n_rows, pb.tile_stride, stream()), "bmp_ggnn_step_fwd")
        ctx.save_for_backward(h, WT, AT, UcT, m, rz, c)
        ctx.pb, ctx.first, ctx.cache = pb, int(first), cache
        if cache is not None:       # how many steps of this call share both weight sets (see backward)
            key = ("n", WT.data_ptr(), AT.data_ptr())
            cache[key] = cache.get(key, 0) + 1
        return hout

    @staticmethod
    def backward(ctx, dhout):
        L = _lib.lib()
        h, WT, AT, UcT, m, rz, c = ctx.saved_tensors
        pb, first, cache = ctx.pb, ctx.first, ctx.cache
        N, d = h.shape
        dev = h.device
        dhout = dhout.contiguous()
        Wnat = _cached(cache, "b", WT, lambda: pack_k4(WT.t()))
        A = _cached(cache, "b", AT, lambda: pack_k4(AT.t()))
        Uc = _cached(cache, "b", UcT, lambda: pack_k4(UcT.t()))
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        dh, gda = f(N, d), f(N, 7 * d)
        tri, trc, skip = step_lists(pb, N, d)
        check(L.bmp_ggnn_step_bwd(ptr(dhout), ptr(h), ptr(rz), ptr(c), pb.n_mtiles, d, first, ptr(pb.csrT_ptr),
                                  ptr(pb.csrT_col), ptr(pb.csrT_val), ptr(Wnat), ptr(A), ptr(Uc), ptr(dh), ptr(gda),
                                  ptr(pb.mt_row0), ptr(pb.mt_nblk), pb.n_rows, pb.tile_stride, skip, stream()), "bmp_ggnn_step_bwd")
        # Steps that share BOTH weight sets (tied layers) accumulate their weight gradients in one set of
        # buffers inside the kernels; only the last of them to run hands the sums to autograd.
        grp = ("g", WT.data_ptr(), AT.data_ptr())
        st = None
        if cache is not None and cache.get(("n",) + grp[1:], 1) > 1:
            st = cache.setdefault(grp, dict(seen=0, buf=None))
        if st is None or st["buf"] is None:
            o1, o2, dUcT, cs = f(d, 7 * d), f(d, 3 * d), f(d, d), f(7 * d)
            if st is not None:
                st["buf"] = (o1, o2, dUcT, cs)
        else:
            o1, o2, dUcT, cs = st["buf"]
        acc = 1 if (st is not None and st["seen"] > 0) else 0
        nws = L.bmp_ggnn_step_wgrad_ws_floats(N, d)
        ws = _ws(nws, dev)
        check(L.bmp_ggnn_step_wgrad(ptr(h), ptr(m), ptr(rz), ptr(gda), N, d, first, ptr(o1), ptr(o2), ptr(dUcT), ptr(cs),
                                    acc, ptr(tri), ptr(trc), None, None, ptr(ws), nws, stream()), "bmp_ggnn_step_wgrad")
        if st is not None:
            st["seen"] += 1
            if st["seen"] < cache[("n",) + grp[1:]]:
                return dh, None, None, None, None, None, None, None, None
            # the group's last step of THIS backward: a later backward over the same graph (retain_graph, a second
            # autograd.grad) starts a fresh accumulation in fresh buffers instead of adding to these sums
            st["seen"], st["buf"] = 0, None
        dWT = o1[:, :4 * d].reshape(d, 4, d).permute(1, 0, 2).reshape(4 * d, d)      # [k][e*d+c] -> [e*d+k][c]
        dbE = cs[:4 * d].reshape(4, d)
        dAT = torch.cat((o1[:, 4 * d:], o2), dim=0)
        return dh, dWT, dbE, dAT, dUcT, cs[4 * d:], None, None, None


def flush_deferred_bwd(state) -> None:
    """Backward launches held back so that they queue on the weight-gradient stream BEHIND the launches of the node that
    follows (bmp.mlp.MLPLossFn: the link predictor's weight-gradient partials are nobody's input, the largest size class of the
    pair kernels -- same stream -- is the chain's)."""
    if state is not None and state.get("deferred_bwd"):
        fns = state["deferred_bwd"]
        state["deferred_bwd"] = []
        for fn in fns:
            fn()


def flush_deferred(state) -> None:
    """Launch what PReadoutFn's off-chain form held back."""
    if state is not None and state.get("deferred"):
        todo, state["deferred"] = state["deferred"], []
        for launch in todo:
            launch()


def _readout_fwd(h, h0, pb, WT, WTp, b, act_j, o, st=None, out=None, infer=False):
    """The readout forward: one kernel per tile when the shape allows (WTp = pack_k4(WT), made here if not given),
    else row GEMM + segment sum.  Returns (ij, g); ``infer`` (forward-only evaluation): the tile kernel keeps no ij."""
    L = _lib.lib()
    st = stream() if st is None else st
    N, d = h.shape
    d0 = 0 if h0 is None else h0.shape[1]
    tile = pb.row_mol is not None and not pb.oversized and bool(L.bmp_readout_tile_supported(d, d0, o))
    if out is not None:
        ij, g = out
    else:
        ij = None if (infer and tile) else torch.empty(N, 2 * o, dtype=torch.float32, device=h.device)
        g = torch.empty(pb.n_mols, o, dtype=torch.float32, device=h.device)
    if tile:
        if WTp is None:
            WTp = pack_k4(WT)
        check(L.bmp_readout_tile_fwd(ptr(h), ptr(h0), pb.n_tiles, d, ptr(WTp), ptr(b), act_j, ptr(pb.row_w), ptr(pb.row_mol),
                                     ptr(pb.mol_nrows), ptr(ij), ptr(g), st), "bmp_readout_tile_fwd")
    else:
        check(L.bmp_readout_fwd(ptr(h), ptr(h0), pb.n_tiles, d, d0, o, ptr(WT), ptr(b), act_j, ptr(pb.row_w),
                                ptr(pb.mol_row0), ptr(pb.mol_nrows), pb.n_mols, ptr(ij), ptr(g), st), "bmp_readout_fwd")
    return ij, g


class ReadoutFn(Function):
    """Gated-sum readout (models/ggnn.py:333-341; models/readout/ggnn_readout.py:42-57).
    WT [(d+d0) x 2o] cols [i|j]; b [2o] or None; h0 may be None."""

    @staticmethod
    def forward(ctx, h, h0, WT, b, pb, act_j):
        L = _lib.lib()
        require_rows(h, "readout: h")
        _check_pb(pb, h)
        if h0 is not None:
            require_rows(h0, "readout: h0")
        N, d = h.shape
        d0 = 0 if h0 is None else h0.shape[1]
        o = WT.shape[1] // 2
        if WT.shape[0] != d + d0 or WT.shape[1] != 2 * o:
            raise ValueError("readout: weight shape does not match [h, h0]")
        WT = WT.contiguous()
        b = None if b is None else b.contiguous()
        ij, g = _readout_fwd(h, h0, pb, WT, None, b, act_j, o)
        ctx.save_for_backward(h, h0 if h0 is not None else torch.empty(0), WT, ij)
        ctx.pb, ctx.act_j, ctx.has_h0, ctx.has_b, ctx.o = pb, act_j, h0 is not None, b is not None, o
        return g

    @staticmethod
    def backward(ctx, dg):
        L = _lib.lib()
        h, h0, WT, ij = ctx.saved_tensors
        pb, o = ctx.pb, ctx.o
        dg = dg.contiguous()
        N, d = h.shape
        dev = h.device
        h0 = h0 if ctx.has_h0 else None
        d0 = h0.shape[1] if ctx.has_h0 else 0
        Wnat = WT.t().contiguous()
        dh = torch.empty_like(h)
        dh0 = torch.empty_like(h0) if ctx.has_h0 else None
        dWT = torch.empty_like(WT)
        db = torch.empty(2 * o, dtype=torch.float32, device=dev) if ctx.has_b else None
        nws = L.bmp_readout_bwd_ws_floats(pb.n_tiles, d, d0, o)
        ws = _ws(nws, dev)
        check(L.bmp_readout_bwd(ptr(dg), ptr(h), ptr(h0), pb.n_tiles, d, d0, o, ptr(Wnat), ptr(ij), ctx.act_j,
                                ptr(pb.row_w), ptr(pb.mol_row0), ptr(pb.mol_nrows), pb.n_mols, ptr(dh), ptr(dh0), ptr(dWT),
                                ptr(db), 0, ptr(ws), nws, stream(), None), "bmp_readout_bwd")
        return dh, dh0, dWT, db, None, None


class LinearRowsFn(Function):
    """GraphLinear on packed rows: Y = act(X . WT + b).  X [N x K] with N a multiple of the
    tile size; WT [K x Nout]."""

    @staticmethod
    def forward(ctx, X, WT, b, act):
        L = _lib.lib()
        require_rows(X, "linear: X")
        N, K = X.shape
        R = L.bmp_tile_rows()
        if N % R or K % 8:
            raise ValueError(f"linear: rows must be a multiple of {R} and K a multiple of 8")
        WT = WT.contiguous()
        b = None if b is None else b.contiguous()
        Nout = WT.shape[1]
        Y = torch.empty(N, Nout, dtype=torch.float32, device=X.device)
        check(L.bmp_linear_fwd(ptr(X), K, N // R, K, Nout, ptr(WT), Nout, ptr(b), act, ptr(Y), Nout, stream()),
              "bmp_linear_fwd")
        ctx.save_for_backward(X, WT, Y)
        ctx.act, ctx.has_b = act, b is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        L = _lib.lib()
        X, WT, Y = ctx.saved_tensors
        N, K = X.shape
        Nout = WT.shape[1]
        R = L.bmp_tile_rows()
        dY = dY.contiguous()
        if ctx.act == 1:
            dY = dY * Y * (1 - Y)
        elif ctx.act == 2:
            dY = dY * (1 - Y * Y)
        elif ctx.act == 3:
            dY = dY * (Y > 0)
        # dX = dY . W  (needs Nout % 8 == 0: pad the K axis of this GEMM with zeros otherwise)
        Wn = WT.t().contiguous()                                   # [Nout x K]
        dYp = dY
        if Nout % 8:
            pad = 8 - Nout % 8
            dYp = torch.nn.functional.pad(dY, (0, pad)).contiguous()
            Wn = torch.nn.functional.pad(Wn, (0, 0, 0, pad)).contiguous()
        dX = torch.empty_like(X)
        check(L.bmp_linear_fwd(ptr(dYp), dYp.shape[1], N // R, dYp.shape[1], K, ptr(Wn), K, None, 0, ptr(dX), K, stream()),
              "bmp_linear_fwd(dX)")
        dWT = torch.empty_like(WT)
        db = torch.empty(Nout, dtype=torch.float32, device=X.device) if ctx.has_b else None
        nws = L.bmp_wgrad_ws_floats_c(N, K, Nout)
        ws = _ws(nws, X.device)
        check(L.bmp_linear_wgrad(ptr(X), K, ptr(dY), Nout, N, K, Nout, ptr(dWT), ptr(db), ptr(ws), nws, stream()),
              "bmp_linear_wgrad")
        return dX, dWT, db, None


# ---------------------------------------------------------------------------------------------------------
# Prepared-weight forms (bmp/plan.py): weights arrive in kernel layout from the plan's one-launch gather and
# the weight gradients stay in the plan's buffers (folded into the flat gradient by one more launch), so
# autograd only carries the row tensors.  ``tape`` is a dummy 1-element tensor that requires grad: it makes
# autograd run the backward of ops whose only differentiable input would have been a weight.
# ---------------------------------------------------------------------------------------------------------
def _register(state, key) -> None:
    """Forward side of _first_write: counts the ops of this step that will write the gradient buffer ``key``."""
    state[("nf", key)] = state.get(("nf", key), 0) + 1


def _first_write(state, key) -> bool:
    """True for the first writer of a gradient buffer of the plan in a backward pass (the kernel overwrites the buffer);
    later writers of the same pass (tied steps; the encoder called once per side in the reference's four-array form)
    accumulate.  Writers are counted against the forward's registrations, so a SECOND backward over the same graph
    (retain_graph) overwrites again instead of adding to the first pass's sums."""
    nf = max(state.get(("nf", key), 1), 1)
    nb = state.get(("nb", key), 0)
    state[("nb", key)] = nb + 1
    return nb % nf == 0


def _at(t: torch.Tensor, row: int):
    """Device pointer of row ``row`` of a contiguous 1-D / 2-D tensor."""
    return _lib.c_void_p(t.data_ptr() + row * t.stride(0) * t.element_size())


def _fwd_parts(state, pb, keep=()):
    """[(tile0, n_tiles, stream handle)] for a tile-local forward launch of the planned encoder.  A step / layer reads and
    writes only its own tile's rows (molecules never straddle tiles), so the two halves of the batch -- the two sides of a
    pair batch -- can advance as two chains on two streams: a launch of 455 tiles on 256 CUs leaves its second round
    0.78 full, two chains of 228 and 227 tiles keep more of them busy (C2 3.00 -> 2.96 ms; four chains on four streams: 3.7-4.0
    ms, not adopted).  Call AFTER the launch's outputs are allocated:
    the second stream is ordered behind everything the current one holds at that moment (whatever used those blocks
    before), and the encoder joins the two before anything reads whole arrays (``_join_parts``).
    ``keep``: every tensor the second stream's launch reads or writes.  They were allocated on the current stream, and under
    ``torch.no_grad()`` nothing else holds them once the Function returns: the plan's state keeps them until the join, so the
    caching allocator cannot hand their blocks to a later main-stream launch while the part stream still uses them."""
    sp = state.get("split") if state is not None else None
    T = pb.n_mtiles
    # (the encoder layout's tile table lists the tallest tiles first: ONE launch in that order lets every CU that finishes
    #  a short tile pick up the next one; two chains would dispatch in an order nobody controls)
    if sp is None or T < 64 or (pb.mt_row0 is not None and not _ENC_SPLIT):
        return ((0, T, stream()),)
    T0 = pb.side_tiles[1] if (len(pb.side_tiles) == 3 and 0 < pb.side_tiles[1] < T) else T // 2
    cur = torch.cuda.current_stream()
    forked = state.get("split_forked", False)          # fork_parts: the chains were opened once for the whole encoder call
    if not forked:
        sp.stream.wait_stream(cur)
    state["split_open"] = True
    state.setdefault("split_keep", []).append(keep)
    if sp.more:              # diagnostic: 2 + len(more) chains (BMP_FWD_CHAINS), each side's range cut into equal pieces
        n = 2 + len(sp.more)
        handles = [stream(), sp.handle] + sp.more_handles
        if not forked:
            for s_ in sp.more:
                s_.wait_stream(cur)
        cuts = [0] + [((T0 * 2 * k) // n if (2 * k) // n == 0 else T0 + ((T - T0) * (2 * k - n)) // n) for k in range(1, n)] + [T]
        return tuple((cuts[k], cuts[k + 1] - cuts[k], handles[k]) for k in range(n) if cuts[k + 1] > cuts[k])
    return ((0, T0, stream()), (T0, T - T0, sp.handle))


def fork_parts(state, pb) -> bool:
    """Open the chains ONCE for a run of tile-local launches whose outputs the caller has already allocated (the planned
    encoder allocates every step's outputs first): the part stream(s) pick up behind the current stream here, and the
    launches that follow on either stream carry no further cross-stream wait until ``_join_parts`` -- each wait is a barrier
    packet between two hardware queues (tens of microseconds) and, placed before every step, it ties chain B's step t to
    chain A's step t-1 (rocprofv3 kernel trace).  Returns whether the chains are open."""
    sp = state.get("split") if state is not None else None
    if sp is None or pb.n_mtiles < 64 or not _FORK_ONCE or (pb.mt_row0 is not None and not _ENC_SPLIT):
        return False
    cur = torch.cuda.current_stream()
    sp.stream.wait_stream(cur)
    for s_ in sp.more:
        s_.wait_stream(cur)
    state["split_forked"] = True
    state["split_open"] = True
    return True


def _join_parts(state) -> None:
    if state is not None and state.get("split_open"):
        state["split"].join()
        state["split_open"] = False
        state["split_forked"] = False
        state["split_keep"] = []


_FORK_ONCE = os.environ.get("BMP_FWD_FORK_ONCE", "1") != "0"          # A/B switch of fork_parts
_ENC_SPLIT = os.environ.get("BMP_ENC_SPLIT", "0") == "1"             # A/B: two chains also over an encoder-layout tile table
_RO_DEFER = os.environ.get("BMP_READOUT_DEFER", "1") != "0"
_RO_OFF_CHAIN = os.environ.get("BMP_READOUT_OFF_CHAIN", "1") != "0"        # A/B switch of PReadoutFn's off_chain form
_RO_PART = os.environ.get("BMP_READOUT_STREAM", "part") == "part"          # A/B: the off-chain readout on the part / the side stream


def _on_side(state, keep, launch) -> None:
    """Weight-gradient launches of the planned path: nothing in the backward chain reads their outputs (the plan's gradient
    buffers, folded into the flat gradient by LayoutPlan.collect), so they may run beside the chain.  With a side stream in
    the plan's state (plan.SideStream: lowest priority, its own workspace) they are enqueued there, behind everything the
    current stream has been given so far: the chain's tile kernels leave the CUs of their last, partly filled round idle
    (455 tiles on 256 CUs) and these workgroups take them, giving way to the chain otherwise.  ``launch(st, ws)`` gets the
    stream handle and a workspace allocator; ``keep``: the row tensors it reads, held until the streams have joined."""
    side = state.get("side") if state is not None else None
    if side is None:
        launch(stream(), _ws)
        return
    side.stream.wait_stream(torch.cuda.current_stream())
    launch(side.handle, side.workspace)
    side.keep.append(keep)
    state["side_used"] = True


def _side_handle(state, keep):
    """The ``stream_w`` argument of bmp_msg_bwd / bmp_gru_bwd / bmp_readout_bwd: the plan's side stream (the entry point
    itself orders it behind the operands it produces), with ``keep`` held until the streams have joined; None = in line."""
    side = state.get("side") if state is not None else None
    if side is None:
        return None
    side.keep.append(keep)
    state["side_used"] = True
    return side.handle


class PEmbedFn(Function):
    @staticmethod
    def forward(ctx, tape, W, ids, dW, state):
        L = _lib.lib()
        V, d = W.shape
        N = ids.numel()
        out = torch.empty(N, d, dtype=torch.float32, device=W.device)
        check(L.bmp_embed_fwd(ptr(ids), ptr(W), N, d, V, ptr(out), stream()), "bmp_embed_fwd")
        ctx.ids, ctx.dW, ctx.V, ctx.state = ids, dW, V, state
        _register(state, "embed.dW")
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        dout = dout.contiguous()
        N, d = dout.shape
        nws = L.bmp_embed_bwd_ws_floats(N, d, ctx.V)
        ws = _ws(nws, dout.device)
        first = _first_write(ctx.state, "embed.dW")
        dst = ctx.dW if first else torch.empty_like(ctx.dW)
        check(L.bmp_embed_bwd(ptr(ctx.ids), ptr(dout), N, d, ctx.V, ptr(dst), ptr(ws), nws, stream()), "bmp_embed_bwd")
        if not first:
            ctx.dW.add_(dst)
        return None, None, None, None, None


def step_buffers(N: int, d: int, device, infer: bool = False):
    """(m, rz, c, hout) of one fused propagation step.  ``infer`` (forward-only evaluation under no-backprop,
    train_binary.py:120-127): m, rz and c -- the backward's inputs -- are not kept (None: the kernel skips their stores)."""
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
    if infer:
        return None, None, None, f(N, d)
    return f(N, d), f(N, 2 * d), f(N, d), f(N, d)


class PStepFn(Function):
    """GGNNStepFn on prepared weights.  W: WTp, bE, ATp, UcTp, b, Wnat_p, A_p, Uc_p; G: o1, o2, dUcT, cs of the
    step's weight group; ``state[gkey]`` says whether the group's buffers already hold a step of this backward."""

    @staticmethod
    def forward(ctx, h, pb, W, G, state, gkey, first, bufs=None):
        L = _lib.lib()
        require_rows(h, "step: h")
        _check_pb(pb, h)
        N, d = h.shape
        m, rz, c, hout = bufs if bufs is not None else step_buffers(N, d, h.device)          # m = rz = c = None: forward only
        for t0, nt, st in _fwd_parts(state, pb, (h, m, rz, c, hout)):
            check(L.bmp_ggnn_step_fwd(ptr(h), t0, nt, d, int(first), ptr(pb.csr_ptr), ptr(pb.csr_col), ptr(pb.csr_val),
                                      ptr(W["WTp"]), ptr(W["bE"]), ptr(W["ATp"]), ptr(W["UcTp"]), ptr(W["b"]), ptr(m), ptr(rz),
                                      ptr(c), ptr(hout), ptr(pb.mt_row0), ptr(pb.mt_nblk), pb.n_rows, pb.tile_stride, st), "bmp_ggnn_step_fwd")
        ctx.save_for_backward(h, m, rz, c)
        ctx.pb, ctx.W, ctx.G, ctx.state, ctx.gkey, ctx.first = pb, W, G, state, gkey, int(first)
        if m is not None:
            type_rows(pb)               # (built once per batch, here on the chain's stream: the backward's side stream finds them)
        _register(state, gkey)
        return hout

    @staticmethod
    def backward(ctx, dhout):
        L = _lib.lib()
        h, m, rz, c = ctx.saved_tensors
        pb, W, G, first = ctx.pb, ctx.W, ctx.G, ctx.first
        N, d = h.shape
        dhout = dhout.contiguous()
        dh = torch.empty(N, d, dtype=torch.float32, device=h.device)
        gda = torch.empty(N, 7 * d, dtype=torch.float32, device=h.device)
        tri, trc, skip = step_lists(pb, N, d)
        lvi, lvc = pb._cache.get("live_rows", (None, None)) if tri is not None else (None, None)      # (a fixed-stride batch)
        check(L.bmp_ggnn_step_bwd(ptr(dhout), ptr(h), ptr(rz), ptr(c), pb.n_mtiles, d, first, ptr(pb.csrT_ptr),
                                  ptr(pb.csrT_col), ptr(pb.csrT_val), ptr(W["Wnat_p"]), ptr(W["A_p"]), ptr(W["Uc_p"]),
                                  ptr(dh), ptr(gda), ptr(pb.mt_row0), ptr(pb.mt_nblk), pb.n_rows, pb.tile_stride, skip, stream()), "bmp_ggnn_step_bwd")
        acc = 0 if _first_write(ctx.state, ctx.gkey) else 1

        def wgrad(st, ws_of):
            nws = L.bmp_ggnn_step_wgrad_ws_floats(N, d)
            ws = ws_of(nws, h.device)
            check(L.bmp_ggnn_step_wgrad(ptr(h), ptr(m), ptr(rz), ptr(gda), N, d, first, ptr(G["o1"]), ptr(G["o2"]),
                                        ptr(G["dUcT"]), ptr(G["cs"]), acc, ptr(tri), ptr(trc), ptr(lvi), ptr(lvc), ptr(ws), nws, st), "bmp_ggnn_step_wgrad")

        _on_side(ctx.state, (h, m, rz, gda), wgrad)
        return dh, None, None, None, None, None, None, None


def _ptr_array(tensors):
    """ctypes array of device pointers (NULL for None) -- a HOST array, as the multi-step entry point takes them."""
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class PReadoutFn(Function):
    """ReadoutFn on prepared weights.  W: WT, b, Wnat; G: dWT, db.

    ``off_chain``: the caller knows that nothing reads the molecule vectors (the fine co-attention family ignores g_1 / g_2,
    nie_coattention.py:335-370, while the encoder computes them all the same, train_binary.py:91-96).  With a side stream in
    the plan's state the readout then runs there, beside the co-attention; its output is complete once the streams have
    joined (LayoutPlan.collect / prepare) and must not be differentiated."""

    @staticmethod
    def forward(ctx, h, h0, pb, W, G, act_j, state, off_chain=False, infer=False):
        L = _lib.lib()
        require_rows(h, "readout: h")
        _check_pb(pb, h)
        N, d = h.shape
        d0 = 0 if h0 is None else h0.shape[1]
        ctx.state = state
        WT = W["WT"]
        o = WT.shape[1] // 2
        side = state.get("side") if (off_chain and state is not None and _RO_OFF_CHAIN) else None
        ctx.off_chain = side is not None
        if side is not None:
            # enqueued LATER (flush_deferred: once the co-attention's forward launches are in their queue): launched here,
            # its 455 one-per-CU workgroups take the CUs before the chain's next launches get to them
            # (no backward will read ij -- the output cannot be differentiated in this form --, so the tile kernel keeps none:
            #  60 MB less to write per C2 step)
            tile = pb.row_mol is not None and not pb.oversized and bool(L.bmp_readout_tile_supported(d, d0, o))
            ij = None if tile else torch.empty(N, 2 * o, dtype=torch.float32, device=h.device)
            g = torch.empty(pb.n_mols, o, dtype=torch.float32, device=h.device)

            sp = state.get("split") if _RO_PART else None

            def launch():
                if sp is not None:
                    # on the forward's second-chain stream, idle by now: the weight-gradient stream's queue stays free for the
                    # launches of the backward that follow at once (the link predictor's partials, the largest size class of the
                    # pair kernels), which otherwise wait behind this one
                    sp.stream.wait_stream(torch.cuda.current_stream())
                    _readout_fwd(h, h0, pb, WT, W.get("WTp"), W.get("b"), act_j, o, sp.handle, out=(ij, g))
                    state.setdefault("split_keep", []).append((h, h0, ij, g))
                    state["split_open"] = True
                    return
                side.stream.wait_stream(torch.cuda.current_stream())
                _readout_fwd(h, h0, pb, WT, W.get("WTp"), W.get("b"), act_j, o, side.handle, out=(ij, g))
                side.keep.append((h, h0, ij, g))
                state["side_used"] = True
            if _RO_DEFER:
                state.setdefault("deferred", []).append(launch)
            else:
                launch()
            return g
        _register(state, "ro")
        ij, g = _readout_fwd(h, h0, pb, WT, W.get("WTp"), W.get("b"), act_j, o, infer=infer)
        ctx.save_for_backward(h, ij, *([h0] if h0 is not None else []))
        ctx.pb, ctx.W, ctx.G, ctx.act_j, ctx.o = pb, W, G, act_j, o
        return g

    @staticmethod
    def backward(ctx, dg):
        L = _lib.lib()
        if ctx.off_chain:
            raise RuntimeError("readout: the molecule vectors were declared unused (off_chain) and computed beside the chain; "
                               "they cannot be differentiated")
        sv = ctx.saved_tensors
        h, ij = sv[0], sv[1]
        h0 = sv[2] if len(sv) > 2 else None
        pb, o, W, G = ctx.pb, ctx.o, ctx.W, ctx.G
        dg = dg.contiguous()
        N, d = h.shape
        d0 = 0 if h0 is None else h0.shape[1]
        dh = torch.empty_like(h)
        dh0 = None if h0 is None else torch.empty_like(h0)
        nws = L.bmp_readout_bwd_ws_floats(pb.n_tiles, d, d0, o)
        ws = _ws(nws, h.device)
        acc = 0 if _first_write(ctx.state, "ro") else 1
        check(L.bmp_readout_bwd(ptr(dg), ptr(h), ptr(h0), pb.n_tiles, d, d0, o, ptr(W["Wnat"]), ptr(ij), ctx.act_j,
                                ptr(pb.row_w), ptr(pb.mol_row0), ptr(pb.mol_nrows), pb.n_mols, ptr(dh), ptr(dh0),
                                ptr(G["dWT"]), ptr(G.get("db")), acc, ptr(ws), nws, stream(),
                                _side_handle(ctx.state, (h, h0, ws))), "bmp_readout_bwd")
        return dh, dh0, None, None, None, None, None, None, None


class PGRUFn(Function):
    """GRUFn on prepared weights (bmp/plan.py), for the widths the fused step kernels do not cover.
    W: AT, UcT, b, A (= AT^T), Uc (= UcT^T); G: dAT, dUcT, db of the GRU mode's group."""

    @staticmethod
    def forward(ctx, h, m, pb, W, G, state, gkey, first):
        L = _lib.lib()
        require_rows(h, "gru: h")
        require_rows(m, "gru: m", h.shape[1])
        _check_pb(pb, h)
        N, d = h.shape
        f = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=h.device)
        rz, c, hout = f(N, 2 * d), f(N, d), f(N, d)
        R = pb.R
        parts = _fwd_parts(state, pb, (h, m, rz, c, hout)) if pb.mt_row0 is None else ((0, pb.n_tiles, stream()),)
        for t0, nt, st in parts:          # row-wise: a tile range is a pointer offset
            r0 = t0 * R
            check(L.bmp_gru_fwd(_at(h, r0), _at(m, r0), nt, d, int(first), ptr(W["AT"]), ptr(W["UcT"]), ptr(W["b"]), _at(rz, r0),
                                _at(c, r0), _at(hout, r0), st), "bmp_gru_fwd")
        ctx.save_for_backward(h, m, rz, c)
        ctx.pb, ctx.W, ctx.G, ctx.state, ctx.gkey, ctx.first = pb, W, G, state, gkey, int(first)
        if m is not None:
            type_rows(pb)               # (built once per batch, here on the chain's stream: the backward's side stream finds them)
        _register(state, gkey)
        return hout

    @staticmethod
    def backward(ctx, dhout):
        L = _lib.lib()
        h, m, rz, c = ctx.saved_tensors
        pb, W, G = ctx.pb, ctx.W, ctx.G
        N, d = h.shape
        dhout = dhout.contiguous()
        dh, dm = torch.empty_like(h), torch.empty_like(m)
        acc = 0 if _first_write(ctx.state, ctx.gkey) else 1          # a tied step's later calls add into the slots
        nws = L.bmp_gru_bwd_ws_floats(pb.n_tiles, d)
        ws = _ws(nws, h.device)
        check(L.bmp_gru_bwd(ptr(dhout), ptr(h), ptr(m), ptr(rz), ptr(c), pb.n_tiles, d, ctx.first, ptr(W["A"]), ptr(W["Uc"]),
                            ptr(dh), ptr(dm), ptr(G["dAT"]), ptr(G["dUcT"]), ptr(G["db"]), acc, ptr(ws), nws, stream(),
                            _side_handle(ctx.state, (h, m, rz, ws))), "bmp_gru_bwd")
        return dh, dm, None, None, None, None, None, None


class PMsgFn(Function):
    """MsgFn (message / RelGCN layer) on prepared weights.  W: WT, bE, WsT, bs, Wnat, Ws; G: dWT, dbE, dWsT, dbs."""

    @staticmethod
    def forward(ctx, x, pb, W, G, state, gkey, act):
        L = _lib.lib()
        require_rows(x, "msg: x")
        _check_pb(pb, x)
        d_in, d_out = x.shape[1], W["WT"].shape[1]
        N = x.shape[0]
        f = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=x.device)
        agg, wdeg, out = f(N, 4 * d_in), f(N, 4), f(N, d_out)
        # Without a self connection x is only the gather's source, addressed by the absolute row numbers of the CSR: a tile
        # range is then an offset into the row pointers and the outputs (the entries they index stay where they are).
        parts = (_fwd_parts(state, pb, (x, agg, wdeg, out)) if (W.get("WsT") is None and pb.mt_row0 is None)
                 else ((0, pb.n_tiles, stream()),))
        R = pb.R
        for t0, nt, st in parts:
            r0 = t0 * R
            check(L.bmp_msg_fwd(ptr(x), d_in, nt, d_in, d_out, _at(pb.csr_ptr, r0), ptr(pb.csr_col), ptr(pb.csr_val),
                                ptr(W["WT"]), ptr(W["bE"]), ptr(W.get("WsT")), ptr(W.get("bs")), act, _at(agg, r0), _at(wdeg, r0),
                                _at(out, r0), d_out, st), "bmp_msg_fwd")
        type_rows(pb, forward=True)      # (once per batch, on the chain's stream: the backward's side-stream launch walks the lists)
        ctx.save_for_backward(x, agg, wdeg, out)
        ctx.pb, ctx.W, ctx.G, ctx.state, ctx.gkey, ctx.act = pb, W, G, state, gkey, act
        _register(state, gkey)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x, agg, wdeg, out = ctx.saved_tensors
        pb, W, G = ctx.pb, ctx.W, ctx.G
        dout = dout.contiguous()
        N, d_out = dout.shape
        d_in = x.shape[1]
        dx = torch.empty(N, d_in, dtype=torch.float32, device=x.device)
        acc = 0 if _first_write(ctx.state, ctx.gkey) else 1
        nws = L.bmp_msg_bwd_ws_floats(pb.n_tiles, d_in, d_out)
        ws = _ws(nws, x.device)
        trf, trc = type_rows(pb, forward=True)           # (built in the forward, on the chain's stream)
        check(L.bmp_msg_bwd(ptr(dout), d_out, ptr(out), d_out, ctx.act, ptr(x), d_in, pb.n_tiles, d_in, d_out,
                            ptr(pb.csrT_ptr), ptr(pb.csrT_col), ptr(pb.csrT_val), ptr(W["Wnat"]), ptr(W.get("Ws")), ptr(agg),
                            ptr(wdeg), ptr(dx), ptr(G["dWT"]), ptr(G["dbE"]), ptr(G.get("dWsT")), ptr(G.get("dbs")), acc, ptr(trf), ptr(trc),
                            ptr(ws), nws, stream(), _side_handle(ctx.state, (x, agg, wdeg, dout, ws))), "bmp_msg_bwd")
        return dx, None, None, None, None, None, None


def rel_layer_supported(d_in: int, d_out: int) -> bool:
    return bool(_lib.lib().bmp_relgcn_layer_supported(int(d_in), int(d_out)))


def rel_buffers(N: int, d: int, device, infer: bool = False):
    """(out, wdeg) of one fused RelGCN layer; ``infer``: wdeg (the backward's operand) is not kept."""
    return (torch.empty(N, d, dtype=torch.float32, device=device),
            None if infer else torch.empty(N, 4, dtype=torch.float32, device=device))


def _rel_fwd(x, pb, WTp, bE, WsTp, bs, act, state=None, bufs=None):
    L = _lib.lib()
    N, d = x.shape
    out, wdeg = bufs if bufs is not None else rel_buffers(N, d, x.device)
    if wdeg is not None:
        type_rows(pb)                   # (once per batch, on the chain's stream: the backward's weight-gradient launches walk them)
    for t0, nt, st in _fwd_parts(state, pb, (x, out, wdeg)):
        check(L.bmp_relgcn_layer_fwd(ptr(x), t0, nt, d, ptr(pb.csr_ptr), ptr(pb.csr_col), ptr(pb.csr_val), ptr(WTp), ptr(bE),
                                     ptr(WsTp), ptr(bs), act, ptr(out), ptr(wdeg), ptr(_rel_mt(pb)[0]), ptr(_rel_mt(pb)[1]), pb.n_rows, st),
              "bmp_relgcn_layer_fwd")
    return out, wdeg


def _rel_bwd(dout, out, x, wdeg, pb, Wnat_p, Ws_p, act, o1, dbE, cs, accumulate, state=None):
    L = _lib.lib()
    N, d = x.shape
    dx = torch.empty(N, d, dtype=torch.float32, device=x.device)
    gda = torch.empty(N, 5 * d, dtype=torch.float32, device=x.device)
    tri, trc, skip = step_lists(pb, N, d)
    check(L.bmp_relgcn_layer_bwd(ptr(dout), ptr(out), act, pb.n_mtiles, d, ptr(pb.csrT_ptr), ptr(pb.csrT_col), ptr(pb.csrT_val),
                                 ptr(Wnat_p), ptr(Ws_p), ptr(dx), ptr(gda), ptr(_rel_mt(pb)[0]), ptr(_rel_mt(pb)[1]), pb.n_rows, skip, stream()),
              "bmp_relgcn_layer_bwd")

    def wgrad(st, ws_of):
        nws = L.bmp_relgcn_layer_wgrad_ws_floats(N, d)
        ws = ws_of(nws, x.device)
        check(L.bmp_relgcn_layer_wgrad(ptr(x), ptr(wdeg), ptr(gda), N, d, ptr(o1), ptr(dbE), ptr(cs), int(accumulate), ptr(tri), ptr(trc),
                                       ptr(ws), nws, st), "bmp_relgcn_layer_wgrad")

    _on_side(state, (x, wdeg, gda), wgrad)
    return dx


class RelLayerFn(Function):
    """One whole RelGCN layer as ONE fused kernel per tile (models/update/relgcn_update.py:24-44 + models/relgcn.py:71),
    d_in == d_out in {64, 128}.  Weight layouts as MsgFn: WT [4d x d], bE [4 x d], WsT [d x d], bs [d]."""

    @staticmethod
    def forward(ctx, x, WT, bE, WsT, bs, pb, act):
        require_rows(x, "relgcn layer: x")
        _check_pb(pb, x)
        d = x.shape[1]
        if tuple(WT.shape) != (4 * d, d) or tuple(WsT.shape) != (d, d):
            raise ValueError("relgcn layer: weight shapes do not match x")
        out, wdeg = _rel_fwd(x, pb, pack_k4(WT), bE.contiguous(), pack_k4(WsT), bs.contiguous(), act)
        ctx.save_for_backward(x, WT, WsT, out, wdeg)
        ctx.pb, ctx.act = pb, act
        return out

    @staticmethod
    def backward(ctx, dout):
        x, WT, WsT, out, wdeg = ctx.saved_tensors
        d = x.shape[1]
        f = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=x.device)
        o1, dbE, cs = f(d, 5 * d), f(4, d), f(5 * d)
        dx = _rel_bwd(dout.contiguous(), out, x, wdeg, ctx.pb, pack_k4(WT.t()), pack_k4(WsT.t()), ctx.act, o1, dbE, cs, 0)
        dWT = o1[:, :4 * d].reshape(d, 4, d).permute(1, 0, 2).reshape(4 * d, d)        # [k][e*d+c] -> [e*d+k][c]
        return dx, dWT, dbE, o1[:, 4 * d:], cs[4 * d:], None, None


class PRelLayerFn(Function):
    """RelLayerFn on prepared weights (bmp/plan.py).  W: WTp, bE, WsTp, bs, Wnat_p, Ws_p; G: o1 [d x 5d], dbE, cs [5d]."""

    @staticmethod
    def forward(ctx, x, pb, W, G, state, gkey, act, bufs=None):
        require_rows(x, "relgcn layer: x")
        _check_pb(pb, x)
        out, wdeg = _rel_fwd(x, pb, W["WTp"], W["bE"], W["WsTp"], W["bs"], act, state, bufs)
        ctx.save_for_backward(x, out, wdeg)
        ctx.pb, ctx.W, ctx.G, ctx.state, ctx.gkey, ctx.act = pb, W, G, state, gkey, act
        _register(state, gkey)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, wdeg = ctx.saved_tensors
        W, G = ctx.W, ctx.G
        first = _first_write(ctx.state, ctx.gkey)
        dx = _rel_bwd(dout.contiguous(), out, x, wdeg, ctx.pb, W["Wnat_p"], W["Ws_p"], ctx.act, G["o1"], G["dbE"], G["cs"],
                      0 if first else 1, ctx.state)
        return dx, None, None, None, None, None, None, None

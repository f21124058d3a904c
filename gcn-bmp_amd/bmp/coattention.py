"""Drug-pair co-attention modules with the reference's signatures
(models/coattention/*.py: ``X(hidden_dim, out_dim, head, activation=...)``,
``__call__(atoms_1, g_1, atoms_2, g_2) -> (compact_1, compact_2)``).

``atoms_k`` is what ``graph_conv.get_atom_array()`` returned: a PackedAtoms (per-row atom
states of a packed batch).  The fine family ignores g_1/g_2 (nie_coattention.py:335-370).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch
from torch import nn
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr, stream
from .functional import ACT, _side_handle, _ws, flush_deferred
from .ggnn import Linear, PackedAtoms, as_packed_atoms


class Bilinear(nn.Module):
    """Parameters of chainer.links.Bilinear(l, r, o): W [l x r x o], V1 [l x o], V2 [r x o], b [o]."""

    def __init__(self, left: int, right: int, out: int):
        super().__init__()
        self.W = nn.Parameter(torch.randn(left, right, out) / math.sqrt(left))
        self.V1 = nn.Parameter(torch.randn(left, out) / math.sqrt(left))
        self.V2 = nn.Parameter(torch.randn(right, out) / math.sqrt(right))
        self.b = nn.Parameter(torch.zeros(out))


def _size_classes(nr1: np.ndarray, nr2: np.ndarray, device):
    """Pairs grouped by size class ceil(max(n1, n2) / 32): the pair kernels are launched once per
    class with LDS sized for it (most drug pairs are <= 32 or <= 64 atoms).  Class 4: a molecule of more than 128 rows
    (more than one tile); those pairs run the same kernels out of global memory (bmp_coattn_nie_*: the `nbig` class).
    Returns (order, counts[5], order_f, counts_f[5], np_big = largest row count among the class-4 pairs)."""
    big = np.maximum(nr1, nr2)
    cls = np.minimum((big + 31) // 32 - 1, 4)
    order = np.argsort(cls, kind="stable").astype(np.int32)
    counts = np.bincount(cls, minlength=5)
    # Forward: the classes 0..3 as ONE launch sized by the largest of them, biggest pairs first.  Its LDS need is small
    # (48 KB at 96 rows: three workgroups per CU), and three launches of 35..550 workgroups each cost one workgroup's
    # latency apiece.  (The backward keeps the per-class launches: at 108 KB per 96-row pair one launch would run one pair
    # per CU.)  The oversized pairs come first in the order and are a launch of their own.
    counts_f = [0, 0, 0, 0, int(counts[4])]
    if len(cls) > counts[4]:
        counts_f[int(np.nonzero(counts[:4])[0].max())] = int(len(cls) - counts[4])
    order_f = np.argsort(-cls, kind="stable").astype(np.int32)
    np_big = int(big[cls == 4].max()) if counts[4] else 0
    return (torch.from_numpy(order).to(device), [int(c) for c in counts],
            torch.from_numpy(order_f).to(device), counts_f, np_big)


def _cbuf_floats(n1: np.ndarray, n2: np.ndarray) -> np.ndarray:
    """Floats a pair keeps between forward and backward: C (n2 x n1) and the softmax statistics of its columns and rows
    (cmax, 1/D2 per side-1 atom; rmax, 1/D1 per side-2 atom) -- include/bmp.h, bmp_coattn_cbuf_floats."""
    return n1 * n2 + 2 * (n1 + n2)


def pair_rows(at1: PackedAtoms, at2: PackedAtoms):
    """Row tensors and per-pair row ranges of the two sides.

    * one two-sided batch (fast path): side 1 = tiles [0, T1), side 2 = tiles [T1, T); the row
      tensors are views of the same buffer and side-2 row offsets are made relative to its view;
    * two single-sided batches (the reference's call order: graph_conv twice, then attn)."""
    pb1, pb2 = at1.pb, at2.pb
    R = pb1.R
    if pb1 is pb2 and len(pb1.side_mols) == 3:
        key = "pair_meta"
        if key not in pb1._cache:
            B = pb1.side_mols[1]
            if pb1.side_mols[2] != 2 * B:
                raise ValueError("co-attention needs as many side-2 as side-1 molecules")
            T1 = pb1.side_tiles[1]
            nr = pb1.mol_nrows_host
            coff = np.concatenate(([0], np.cumsum(_cbuf_floats(nr[:B], nr[B:]))))
            pb1._cache[key] = dict(
                B=B, T1=T1, T2=pb1.side_tiles[2] - T1,
                r1=pb1.mol_row0[:B].contiguous(), n1=pb1.mol_nrows[:B].contiguous(),
                r2=(pb1.mol_row0[B:] - T1 * R).contiguous(), n2=pb1.mol_nrows[B:].contiguous(),
                coff=torch.from_numpy(coff[:-1].astype(np.int64)).to(pb1.device), ctotal=int(coff[-1]))
            (pb1._cache[key]["order"], pb1._cache[key]["counts"], pb1._cache[key]["order_f"],
             pb1._cache[key]["counts_f"], pb1._cache[key]["np_big"]) = _size_classes(nr[:B], nr[B:], pb1.device)
        m = pb1._cache[key]
        N1 = m["T1"] * R
        X1, X2 = at1.rows[:N1], at1.rows[N1:]
        w1, w2 = pb1.row_w[:N1], pb1.row_w[N1:]
        return X1, X2, w1, w2, m, True
    if at1.rows is at2.rows:
        raise ValueError("both sides point at the same rows of a one-sided batch")
    if pb1.n_mols != pb2.n_mols:
        raise ValueError("co-attention needs as many side-2 as side-1 molecules")
    nr1, nr2 = pb1.mol_nrows_host, pb2.mol_nrows_host
    coff = np.concatenate(([0], np.cumsum(_cbuf_floats(nr1, nr2))))
    m = dict(B=pb1.n_mols, T1=pb1.n_tiles, T2=pb2.n_tiles, r1=pb1.mol_row0, n1=pb1.mol_nrows, r2=pb2.mol_row0,
             n2=pb2.mol_nrows, coff=torch.from_numpy(coff[:-1].astype(np.int64)).to(pb1.device), ctotal=int(coff[-1]))
    m["order"], m["counts"], m["order_f"], m["counts_f"], m["np_big"] = _size_classes(nr1, nr2, pb1.device)
    return at1.rows, at2.rows, pb1.row_w, pb2.row_w, m, False


def _big_ws(meta, H, o, device):
    """Workspace of the pair kernels' oversized class (pairs with a molecule of more than 128 rows): (tensor or None, floats)."""
    nbig = meta["counts_f"][4]
    if not nbig:
        return None, 0
    n = _lib.lib().bmp_coattn_big_ws_floats(meta["np_big"], H, o, nbig, 0)
    return _ws(n, device), n


class NieCoattnFn(Function):
    """bmp_coattn_nie_fwd / _bwd.  Inputs in kernel layout (see include/bmp.h)."""

    @staticmethod
    def forward(ctx, X1, X2, WbT, ZW1T, ZW2T, zb, wa1, wa2, cbias, w1, w2, meta, d, o, H, act, mode=0):
        L = _lib.lib()
        dev = X1.device
        B, T1, T2 = meta["B"], meta["T1"], meta["T2"]
        ZC = L.bmp_coattn_zcols(o, H)
        X1 = X1.contiguous(); X2 = X2.contiguous()
        WbT, ZW1T, ZW2T, zb = WbT.contiguous(), ZW1T.contiguous(), ZW2T.contiguous(), zb.contiguous()
        wa1, wa2, cbias = wa1.contiguous(), wa2.contiguous(), cbias.contiguous()
        N1, N2 = X1.shape[0], X2.shape[0]
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        Q2, Z1, Z2 = f(N2, d), f(N1, ZC), f(N2, ZC)
        Cbuf = f(max(meta["ctotal"], 1))
        H1, H2, al1, al2 = f(N1, H), f(N2, H), f(N1), f(N2)      # written / read only at rows that belong to a pair
        out1, out2 = f(B, o), f(B, o)
        wsb, nwsb = _big_ws(meta, H, o, dev)
        check(L.bmp_coattn_nie_fwd(ptr(X1), T1, ptr(X2), T2, d, o, H, act, mode, ptr(w1), ptr(meta["r1"]), ptr(meta["n1"]),
                                   ptr(w2), ptr(meta["r2"]), ptr(meta["n2"]), ptr(meta["coff"]), B, ptr(meta["order_f"]),
                                   *meta["counts_f"], meta["np_big"], ptr(WbT), ptr(ZW1T), ptr(ZW2T), ptr(zb), ptr(wa1), ptr(wa2),
                                   ptr(cbias), ptr(Q2), ptr(Z1), ptr(Z2), ptr(Cbuf), ptr(H1), ptr(H2), ptr(al1), ptr(al2), ptr(out1),
                                   ptr(out2), ptr(wsb), nwsb, stream()), "bmp_coattn_nie_fwd")
        ctx.save_for_backward(X1, X2, WbT, ZW1T, ZW2T, wa1, wa2, w1, w2, Q2, Z1, Z2, Cbuf, H1, H2, al1, al2)
        ctx.meta, ctx.dims = meta, (d, o, H, act, ZC, mode)
        return out1, out2

    @staticmethod
    def backward(ctx, dout1, dout2):
        L = _lib.lib()
        X1, X2, WbT, ZW1T, ZW2T, wa1, wa2, w1, w2, Q2, Z1, Z2, Cbuf, H1, H2, al1, al2 = ctx.saved_tensors
        meta = ctx.meta
        d, o, H, act, ZC, mode = ctx.dims
        dev = X1.device
        B, T1, T2 = meta["B"], meta["T1"], meta["T2"]
        dout1, dout2 = dout1.contiguous(), dout2.contiguous()
        Wb = WbT.t().contiguous()
        ZW1, ZW2 = ZW1T.t().contiguous(), ZW2T.t().contiguous()
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        dX1, dX2 = f(*X1.shape), f(*X2.shape)
        dWbT, dZW1T, dZW2T, dwa = f(d, d), f(d, ZC), f(d, ZC), f(2 * H + 1)
        dzb = f(2, ZC) if mode & 2 else f(ZC)
        nws = L.bmp_coattn_nie_bwd_ws_floats(T1, T2, d, o, H, B, meta["counts"][4], meta["np_big"])
        ws = _ws(nws, dev)
        check(L.bmp_coattn_nie_bwd(ptr(dout1), ptr(dout2), ptr(X1), T1, ptr(X2), T2, d, o, H, act, mode, ptr(w1),
                                   ptr(meta["r1"]), ptr(meta["n1"]), ptr(w2), ptr(meta["r2"]), ptr(meta["n2"]),
                                   ptr(meta["coff"]), B, ptr(meta["order"]), *meta["counts"], meta["np_big"], ptr(Wb), ptr(ZW1), ptr(ZW2), ptr(wa1),
                                   ptr(wa2),
                                   ptr(Q2), ptr(Z1), ptr(Z2), ptr(Cbuf), ptr(H1), ptr(H2), ptr(al1), ptr(al2), ptr(dX1),
                                   ptr(dX2), ptr(dWbT), ptr(dZW1T), ptr(dZW2T), ptr(dzb), ptr(dwa), ptr(ws), nws,
                                   stream(), None, None, None, None), "bmp_coattn_nie_bwd")
        return (dX1, dX2, dWbT, dZW1T, dZW2T, dzb, dwa[:H], dwa[H:2 * H], dwa[2 * H:], None, None, None, None, None,
                None, None, None)


class PNieFn(Function):
    """NieCoattnFn on prepared weights (bmp/plan.py).  W: WbT, ZW1T, ZW2T, zb, wa1, wa2, cbias, Wb, ZW1, ZW2;
    G: dWbT, dZW1T, dZW2T, dzb, dwa."""

    @staticmethod
    def forward(ctx, X1, X2, W, G, w1, w2, meta, d, o, H, act, mode, state=None, rm1=None, rm2=None, infer=False):
        L = _lib.lib()
        dev = X1.device
        ctx.state = state
        ctx.takes_head_gscale = state is not None      # bmp.mlp.MLPLossFn may leave the loss-gradient factor in ``state`` for this node
        ctx.rm = (rm1, rm2) if (rm1 is not None and rm2 is not None) else (None, None)      # row -> molecule maps (dead rows)
        B, T1, T2 = meta["B"], meta["T1"], meta["T2"]
        ZC = L.bmp_coattn_zcols(o, H)
        ctx.joint = X2 is None          # one row tensor for both sides: its gradient comes back as ONE tensor too
        if ctx.joint:                   # (two slices would cost autograd two zero-fills, two copies and an add)
            rows = X1.contiguous()
            X1, X2 = rows[:T1 * _lib.lib().bmp_tile_rows()], rows[T1 * _lib.lib().bmp_tile_rows():]
        else:
            X1 = X1.contiguous(); X2 = X2.contiguous()
        N1, N2 = X1.shape[0], X2.shape[0]
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        Q2, Z1, Z2 = f(N2, d), f(N1, ZC), f(N2, ZC)
        if infer:         # forward-only evaluation (predict): C, its statistics, H and alpha are the backward's and are not kept
            Cbuf = H1 = H2 = al1 = al2 = None
        else:
            Cbuf = f(max(meta["ctotal"], 1))
            H1, H2, al1, al2 = f(N1, H), f(N2, H), f(N1), f(N2)
        out1, out2 = f(B, o), f(B, o)
        wsb, nwsb = _big_ws(meta, H, o, dev)
        check(L.bmp_coattn_nie_fwd(ptr(X1), T1, ptr(X2), T2, d, o, H, act, mode, ptr(w1), ptr(meta["r1"]), ptr(meta["n1"]),
                                   ptr(w2), ptr(meta["r2"]), ptr(meta["n2"]), ptr(meta["coff"]), B, ptr(meta["order_f"]),
                                   *meta["counts_f"], meta["np_big"], ptr(W["WbT"]), ptr(W["ZW1T"]), ptr(W["ZW2T"]), ptr(W["zb"]),
                                   ptr(W["wa1"]), ptr(W["wa2"]), ptr(W["cbias"]), ptr(Q2), ptr(Z1), ptr(Z2), ptr(Cbuf), ptr(H1),
                                   ptr(H2), ptr(al1), ptr(al2), ptr(out1), ptr(out2), ptr(wsb), nwsb, stream()),
              "bmp_coattn_nie_fwd")
        ctx.save_for_backward(X1, X2, w1, w2, Q2, Z1, Z2, Cbuf, H1, H2, al1, al2)
        ctx.meta, ctx.dims, ctx.W, ctx.G = meta, (d, o, H, act, ZC, mode), W, G
        flush_deferred(state)          # the readout the encoder held back: behind this call's launches in the queues
        return out1, out2

    @staticmethod
    def backward(ctx, dout1, dout2):
        L = _lib.lib()
        X1, X2, w1, w2, Q2, Z1, Z2, Cbuf, H1, H2, al1, al2 = ctx.saved_tensors
        meta, W, G = ctx.meta, ctx.W, ctx.G
        d, o, H, act, ZC, mode = ctx.dims
        B, T1, T2 = meta["B"], meta["T1"], meta["T2"]
        dout1, dout2 = dout1.contiguous(), dout2.contiguous()
        # bmp.mlp.MLPLossFn hands over the gradients of the mean loss for d loss = 1 and leaves the factor that arrived at the
        # loss (a device scalar) here: the pair kernels multiply on load
        gs = None
        hs = ctx.state.pop("head_gscale", None) if ctx.state is not None else None
        if hs is not None:
            g, u1, u2 = hs
            if dout1.data_ptr() == u1.data_ptr() and dout2.data_ptr() == u2.data_ptr():
                gs = g
            else:       # something else was added to the head's gradients on the way here: put the factor on the head's share
                dout1, dout2 = dout1 + (g - 1.0) * u1, dout2 + (g - 1.0) * u2
        if ctx.joint:
            dX = torch.empty(X1.shape[0] + X2.shape[0], d, dtype=torch.float32, device=X1.device)
            dX1, dX2 = dX[:X1.shape[0]], dX[X1.shape[0]:]
        else:
            dX1, dX2 = torch.empty_like(X1), torch.empty_like(X2)
        nws = L.bmp_coattn_nie_bwd_ws_floats(T1, T2, d, o, H, B, meta["counts"][4], meta["np_big"])
        ws = _ws(nws, X1.device)
        check(L.bmp_coattn_nie_bwd(ptr(dout1), ptr(dout2), ptr(X1), T1, ptr(X2), T2, d, o, H, act, mode, ptr(w1),
                                   ptr(meta["r1"]), ptr(meta["n1"]), ptr(w2), ptr(meta["r2"]), ptr(meta["n2"]),
                                   ptr(meta["coff"]), B, ptr(meta["order"]), *meta["counts"], meta["np_big"], ptr(W["Wb"]), ptr(W["ZW1"]),
                                   ptr(W["ZW2"]), ptr(W["wa1"]), ptr(W["wa2"]),
                                   ptr(Q2), ptr(Z1), ptr(Z2), ptr(Cbuf), ptr(H1), ptr(H2), ptr(al1), ptr(al2), ptr(dX1),
                                   ptr(dX2), ptr(G["dWbT"]), ptr(G["dZW1T"]), ptr(G["dZW2T"]), ptr(G["dzb"]), ptr(G["dwa"]),
                                   ptr(ws), nws, stream(), _side_handle(ctx.state, (X1, X2, ws)), ptr(ctx.rm[0]), ptr(ctx.rm[1]),
                                   ptr(gs)), "bmp_coattn_nie_bwd")
        from .functional import flush_deferred_bwd
        flush_deferred_bwd(ctx.state)
        if ctx.joint:
            return (dX,) + (None,) * 15
        return (dX1, dX2) + (None,) * 14


class _FinePlanMixin:
    """Layout plan protocol (bmp/plan.py) of the fine co-attention family, on top of ``_kernel_weights()``."""

    # the family's __call__ never reads g_1 / g_2 (nie_coattention.py:335-370): the pair predictor may tell the encoder so
    ignores_graph_vectors = True

    def plannable(self) -> bool:
        return True

    def primary_layouts(self):
        names = ("WbT", "ZW1T", "ZW2T", "zb", "wa1", "wa2", "cbias")
        return {k: v.contiguous() for k, v in zip(names, self._kernel_weights())}

    def prepared_layouts(self):
        p = self.primary_layouts()
        p.update(Wb=p["WbT"].t().contiguous(), ZW1=p["ZW1T"].t().contiguous(), ZW2=p["ZW2T"].t().contiguous())
        return p

    def gk_spec(self):
        d, o, H = self.hidden_dim, self.out_dim, self._heads()
        ZC = _lib.lib().bmp_coattn_zcols(o, H)
        return {"dWbT": (d, d), "dZW1T": (d, ZC), "dZW2T": (d, ZC), "dzb": (ZC,), "dwa": (2 * H + 1,)}

    def primary_grads(self, gk):
        H = self._heads()
        return {"WbT": [gk["dWbT"]], "ZW1T": [gk["dZW1T"]], "ZW2T": [gk["dZW2T"]], "zb": [gk["dzb"]],
                "wa1": [gk["dwa"][:H]], "wa2": [gk["dwa"][H:2 * H]], "cbias": [gk["dwa"][2 * H:]]}

    def _forward_fast(self, atoms_1, atoms_2, fast, mode):
        P, G, state, _tape = fast
        X1, X2, w1, w2, meta, joint = pair_rows(atoms_1, atoms_2)
        rm1, rm2 = atoms_1.pb.row_mol, atoms_2.pb.row_mol
        if joint:
            n1 = X1.shape[0]
            X1, X2 = atoms_1.rows, None
            if rm1 is not None:
                rm1, rm2 = rm1[:n1], rm1[n1:]
        return PNieFn.apply(X1, X2, P, G, w1, w2, meta, self.hidden_dim, self.out_dim, self._heads(),
                            ACT[self.activation], mode, state, rm1, rm2, not torch.is_grad_enabled())


class NieFineCoattention(_FinePlanMixin, nn.Module):
    """models/coattention/nie_coattention.py:312-396."""

    def __init__(self, hidden_dim, out_dim, head, activation="identity"):
        super().__init__()
        if head >= 16:
            raise ValueError("head must be < 16")
        if hidden_dim % 8 or out_dim % 4:
            raise ValueError("hidden_dim must be a multiple of 8 and out_dim of 4")
        self.energy_layer = Bilinear(hidden_dim, hidden_dim, 1)
        self.attention_layer_1 = Linear(head, 1, nobias=True)
        self.attention_layer_2 = Linear(head, 1, nobias=True)
        self.lt_layer_1 = Linear(hidden_dim, head, nobias=True)
        self.lt_layer_2 = Linear(hidden_dim, head, nobias=True)
        self.j_layer = Linear(hidden_dim, out_dim)
        self.hidden_dim, self.out_dim, self.head = hidden_dim, out_dim, head
        if callable(activation):
            activation = getattr(activation, "__name__", str(activation))
        if activation not in ("identity", "tanh", "sigmoid", "relu"):
            raise ValueError(f"unsupported activation {activation!r}")
        self.activation = activation

    def _heads(self) -> int:
        return self.head

    def _kernel_weights(self):
        d, o, H = self.hidden_dim, self.out_dim, self.head
        ZC = _lib.lib().bmp_coattn_zcols(o, H)
        E = self.energy_layer
        dev = E.W.device
        WbT = E.W[:, :, 0].t()                                                  # [q][p] = W[p][q]
        pad = torch.zeros(d, ZC - o - H - 1, device=dev, dtype=E.W.dtype)
        ZW1T = torch.cat((self.j_layer.W.t(), self.lt_layer_1.W.t(), E.V1, pad), dim=1)
        ZW2T = torch.cat((self.j_layer.W.t(), self.lt_layer_2.W.t(), E.V2, pad), dim=1)
        zb = torch.cat((self.j_layer.b, torch.zeros(ZC - o, device=dev, dtype=E.W.dtype)))
        return WbT, ZW1T, ZW2T, zb, self.attention_layer_1.W[0], self.attention_layer_2.W[0], E.b

    def forward(self, atoms_1, g_1, atoms_2, g_2, **_) -> Tuple[torch.Tensor, torch.Tensor]:
        atoms_1, atoms_2 = as_packed_atoms(atoms_1), as_packed_atoms(atoms_2)      # dense (mb, N, hid) arrays: :335-341
        fast = getattr(self, "_fast", None)
        if fast is not None:
            return self._forward_fast(atoms_1, atoms_2, fast, 0)
        X1, X2, w1, w2, meta, joint = pair_rows(atoms_1, atoms_2)
        WbT, ZW1T, ZW2T, zb, wa1, wa2, cb = self._kernel_weights()
        return NieCoattnFn.apply(X1, X2, WbT, ZW1T, ZW2T, zb, wa1, wa2, cb, w1, w2, meta, self.hidden_dim,
                                 self.out_dim, self.head, ACT[self.activation])


class _DeepNie(NieFineCoattention):
    """models/coattention/nie_coattention.py:13-309 (Deep / VeryDeep / ExtremeDeep): the Nie computation whose head and
    j projections act on ``prev_lt`` chains of the atoms -- n_lt affine layers per side with NO activation between them
    (:54-59, :155-163) -- while the energy C still sees the original atoms (:47).  A chain of affine maps is one
    affine map, so the chain is folded into the projection operands here (d x d products, done by the framework so
    that autograd carries the gradients back to every layer) and the pair kernels run unchanged, with one bias row
    per side.  Rounding differs from the layer-by-layer evaluation at the 1e-6 level."""
    n_lt_layers = 1
    single_layer_names = False

    def __init__(self, hidden_dim, out_dim, head, activation="identity"):
        super().__init__(hidden_dim, out_dim, head, activation)
        d = hidden_dim
        if self.single_layer_names:                                            # nie_coattention.py:27-28
            self.prev_lt_layer_1, self.prev_lt_layer_2 = Linear(d, d), Linear(d, d)
        else:                                                                  # :122-129, :225-232
            self.prev_lt_layers_1 = nn.ModuleList([Linear(d, d) for _ in range(self.n_lt_layers)])
            self.prev_lt_layers_2 = nn.ModuleList([Linear(d, d) for _ in range(self.n_lt_layers)])

    def _chains(self):
        if self.single_layer_names:
            return [self.prev_lt_layer_1], [self.prev_lt_layer_2]
        return list(self.prev_lt_layers_1), list(self.prev_lt_layers_2)

    def plannable(self) -> bool:
        return False                     # the folded operands are products of parameters, not a 0/1 map of them

    @staticmethod
    def _fold(layers):
        """x -> layers[-1](...layers[0](x)) as (A^T [d x d], c [d]): row form x A^T + c."""
        AT, c = None, None
        for lin in layers:
            WT = lin.W.t()
            AT = WT if AT is None else AT @ WT
            c = lin.b if c is None else c @ WT + lin.b
        return AT, c

    def _kernel_weights(self):
        d, o, H = self.hidden_dim, self.out_dim, self.head
        ZC = _lib.lib().bmp_coattn_zcols(o, H)
        E = self.energy_layer
        dev, dt = E.W.device, E.W.dtype
        WbT = E.W[:, :, 0].t()
        pad = torch.zeros(d, ZC - o - H - 1, device=dev, dtype=dt)
        ZW, zb = [], []
        ch1, ch2 = self._chains()
        for layers, lt, V in ((ch1, self.lt_layer_1, E.V1), (ch2, self.lt_layer_2, E.V2)):
            AT, c = self._fold(layers)
            P = torch.cat((self.j_layer.W.t(), lt.W.t()), dim=1)              # [d x (o + H)] on the transformed atoms
            ZW.append(torch.cat((AT @ P, V, pad), dim=1))
            zb.append(torch.cat((c @ P + torch.cat((self.j_layer.b, torch.zeros(H, device=dev, dtype=dt))),
                                 torch.zeros(ZC - o - H, device=dev, dtype=dt))))
        return WbT, ZW[0], ZW[1], torch.stack(zb), self.attention_layer_1.W[0], self.attention_layer_2.W[0], E.b

    def forward(self, atoms_1, g_1, atoms_2, g_2, **_):
        atoms_1, atoms_2 = as_packed_atoms(atoms_1), as_packed_atoms(atoms_2)      # dense (mb, N, hid) arrays: :335-341
        X1, X2, w1, w2, meta, joint = pair_rows(atoms_1, atoms_2)
        WbT, ZW1T, ZW2T, zb, wa1, wa2, cb = self._kernel_weights()
        return NieCoattnFn.apply(X1, X2, WbT, ZW1T, ZW2T, zb, wa1, wa2, cb, w1, w2, meta, self.hidden_dim,
                                 self.out_dim, self.head, ACT[self.activation], 2)


class DeepNieFineCoattention(_DeepNie):
    """nie_coattention.py:13-102; the reference names the single layers prev_lt_layer_{1,2}."""
    n_lt_layers = 1
    single_layer_names = True


class VeryDeepNieFineCoattention(_DeepNie):
    """nie_coattention.py:105-205."""
    n_lt_layers = 2


class ExtremeDeepNieFineCoattention(_DeepNie):
    """nie_coattention.py:208-309."""
    n_lt_layers = 3


class FourierFineCoattention(NieFineCoattention):
    """models/coattention/nie_coattention.py:399-513: the Nie computation with the energy taken between the discrete
    Fourier transforms (over the feature axis, functions.fft :507-515) of the atom states:
    C = act(Bilinear(Re F a1, Re F a2) + Bilinear(Im F a1, Im F a2)) (:489), one Bilinear link called twice, so its
    linear terms and bias enter twice.  With the real DFT matrices Cm[n][k] = cos(2 pi n k / d), Sm = -sin(.) this is
    the ordinary bilinear energy with W' = Cm W Cm + Sm W Sm, V' = (Cm + Sm) V, b' = 2 b: the transform is folded
    into the energy operands (by the framework, so autograd returns the gradients to W, V1, V2, b) and the pair kernels
    run unchanged -- instead of 2 FFTs per atom and two (mb N1 N2, d) Bilinear calls per pair."""

    def plannable(self) -> bool:
        return False                     # the folded operands are not a 0/1 map of the parameters

    def _dft(self, dev, dt):
        d = self.hidden_dim
        key = (str(dev), dt)
        if getattr(self, "_dft_cache", None) is None or self._dft_cache[0] != key:
            k = torch.arange(d, dtype=torch.float64)
            ang = 2.0 * math.pi * ((k[:, None] * k[None, :]) % d) / d
            self._dft_cache = (key, torch.cos(ang).to(device=dev, dtype=dt), (-torch.sin(ang)).to(device=dev, dtype=dt))
        return self._dft_cache[1], self._dft_cache[2]

    def _kernel_weights(self):
        WbT, ZW1T, ZW2T, zb, wa1, wa2, cb = super()._kernel_weights()
        E = self.energy_layer
        Cm, Sm = self._dft(E.W.device, E.W.dtype)
        W = E.W[:, :, 0]
        Wf = Cm @ W @ Cm + Sm @ W @ Sm
        o, H = self.out_dim, self.head
        CS = Cm + Sm
        ZW1T = torch.cat((ZW1T[:, :o + H], CS @ E.V1, ZW1T[:, o + H + 1:]), dim=1)
        ZW2T = torch.cat((ZW2T[:, :o + H], CS @ E.V2, ZW2T[:, o + H + 1:]), dim=1)
        return Wf.t(), ZW1T, ZW2T, zb, wa1, wa2, 2.0 * cb


class VQAParallelCoattention(NieFineCoattention):
    """models/coattention/vqa_parallel_coattention.py:13-102: the Nie computation with tanh default."""

    def __init__(self, hidden_dim, out_dim, head, activation="tanh"):
        super().__init__(hidden_dim, out_dim, head, activation)


class PoolingFineCoattention(_FinePlanMixin, nn.Module):
    """models/coattention/PoolingFineCoattention.py:13-81: same bilinear energy C as Nie; the atom
    weights are softmax(mean of C over the other side's padded positions); no head projections."""

    def __init__(self, hidden_dim, out_dim, activation="tanh"):
        super().__init__()
        if hidden_dim % 8 or out_dim % 4:
            raise ValueError("hidden_dim must be a multiple of 8 and out_dim of 4")
        self.energy_layer = Bilinear(hidden_dim, hidden_dim, 1)
        self.j_layer = Linear(hidden_dim, out_dim)
        self.hidden_dim, self.out_dim = hidden_dim, out_dim
        if callable(activation):
            activation = getattr(activation, "__name__", str(activation))
        if activation not in ("identity", "tanh", "sigmoid", "relu"):
            raise ValueError(f"unsupported activation {activation!r}")
        self.activation = activation

    def forward(self, atoms_1, g_1, atoms_2, g_2, **_):
        atoms_1, atoms_2 = as_packed_atoms(atoms_1), as_packed_atoms(atoms_2)      # dense (mb, N, hid) arrays: :335-341
        fast = getattr(self, "_fast", None)
        if fast is not None:
            return self._forward_fast(atoms_1, atoms_2, fast, 1)
        X1, X2, w1, w2, meta, _joint = pair_rows(atoms_1, atoms_2)
        WbT, ZW1T, ZW2T, zb, wa1, wa2, cb = self._kernel_weights()
        return NieCoattnFn.apply(X1, X2, WbT, ZW1T, ZW2T, zb, wa1, wa2, cb, w1, w2, meta, self.hidden_dim, self.out_dim,
                                 1, ACT[self.activation], 1)

    def _heads(self) -> int:
        return 1          # one (unused, zero-weight) head column keeps the row layout

    def _kernel_weights(self):
        d, o, H = self.hidden_dim, self.out_dim, 1
        ZC = _lib.lib().bmp_coattn_zcols(o, H)
        E = self.energy_layer
        dev = E.W.device
        pad = torch.zeros(d, ZC - o - H - 1, device=dev, dtype=E.W.dtype)
        zcol = torch.zeros(d, H, device=dev, dtype=E.W.dtype)
        ZW1T = torch.cat((self.j_layer.W.t(), zcol, E.V1, pad), dim=1)
        ZW2T = torch.cat((self.j_layer.W.t(), zcol, E.V2, pad), dim=1)
        zb = torch.cat((self.j_layer.b, torch.zeros(ZC - o, device=dev, dtype=E.W.dtype)))
        wa = torch.zeros(H, device=dev, dtype=E.W.dtype)
        return E.W[:, :, 0].t(), ZW1T, ZW2T, zb, wa, wa, E.b

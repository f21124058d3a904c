"""Coarse (atom x molecule-vector) co-attention modules with the reference's signatures:
ParallelCoattention (parallel_coattention.py:12-84), CircularParallelCoattention (:87-187), AlternatingCoattention
(alternating_coattention.py:11-86), GlobalCoattention (global_coattention.py:12-73) and
NeuralCoattention (neural_coattention.py:11-71).

Dense projections run through the row GEMM (LinearRowsFn), the per-molecule arithmetic through
the segment operators of csrc/bmp_seg.hip; only pointwise activations on small tensors are torch
ops.  All atom sums carry the row multiplicities (unmasked zero padding of the reference).
SURVEY.md 8(a) R8' rates this family "negligible" in cost.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr, stream
from .coattention import Bilinear
from .functional import ACT, LinearRowsFn
from .ggnn import Linear, PackedAtoms

_ACT = {"identity": lambda x: x, "tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}


def _f(dev, *s):
    return torch.empty(*s, dtype=torch.float32, device=dev)


class SegPoolFn(Function):
    """out[m] = sum_rows w * A * Y ; A [N x 1] (per-atom weight) or [N x o] (gate)."""

    @staticmethod
    def forward(ctx, A, Y, w, row0, nrows):
        L = _lib.lib()
        A, Y = A.contiguous(), Y.contiguous()
        N, o = Y.shape
        ca = A.shape[1]
        M = row0.numel()
        out = _f(Y.device, M, o)
        check(L.bmp_segpool_fwd(ptr(A), ca, ptr(Y), o, ptr(w), ptr(row0), ptr(nrows), M, ptr(out), stream()), "bmp_segpool_fwd")
        ctx.save_for_backward(A, Y, w, row0, nrows)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        A, Y, w, row0, nrows = ctx.saved_tensors
        N, o = Y.shape
        dA, dY = torch.empty_like(A), torch.empty_like(Y)
        dout = dout.contiguous()
        check(L.bmp_segpool_bwd(ptr(dout), ptr(A), A.shape[1], ptr(Y), o, ptr(w), ptr(row0), ptr(nrows),
                                row0.numel(), N, ptr(dA), ptr(dY), stream()), "bmp_segpool_bwd")
        return dA, dY, None, None, None


class SegSoftmaxFn(Function):
    @staticmethod
    def forward(ctx, s, w, row0, nrows):
        L = _lib.lib()
        s = s.contiguous()
        alpha = torch.empty_like(s)
        check(L.bmp_segsoftmax_fwd(ptr(s), ptr(w), ptr(row0), ptr(nrows), row0.numel(), s.numel(), ptr(alpha), stream()),
              "bmp_segsoftmax_fwd")
        ctx.save_for_backward(alpha, w, row0, nrows)
        return alpha

    @staticmethod
    def backward(ctx, dalpha):
        L = _lib.lib()
        alpha, w, row0, nrows = ctx.saved_tensors
        ds = torch.empty_like(alpha)
        dalpha = dalpha.contiguous()
        check(L.bmp_segsoftmax_bwd(ptr(dalpha), ptr(alpha), ptr(w), ptr(row0), ptr(nrows), row0.numel(),
                                   alpha.numel(), ptr(ds), stream()), "bmp_segsoftmax_bwd")
        return ds, None, None, None


class RowBcastFn(Function):
    """out[r] = q[row_mol[r]] -- the reference's expand_dims + tile of a molecule vector over the atoms."""

    @staticmethod
    def forward(ctx, q, row_mol, row0, nrows):
        L = _lib.lib()
        q = q.contiguous()
        N, c = row_mol.numel(), q.shape[1]
        out = _f(q.device, N, c)
        check(L.bmp_rowbcast_fwd(ptr(q), c, ptr(row_mol), N, ptr(out), stream()), "bmp_rowbcast_fwd")
        ctx.save_for_backward(row0, nrows)
        ctx.M = q.shape[0]
        return out

    @staticmethod
    def backward(ctx, d):
        L = _lib.lib()
        row0, nrows = ctx.saved_tensors
        d = d.contiguous()
        c = d.shape[1]
        dq = torch.zeros(ctx.M, c, dtype=torch.float32, device=d.device)
        check(L.bmp_rowbcast_bwd(ptr(d), c, ptr(row0), ptr(nrows), row0.numel(), ptr(dq), stream()), "bmp_rowbcast_bwd")
        return dq, None, None, None


class RowDotFn(Function):
    """s[r] = x[r] . u[row_mol[r]] + s0[row_mol[r]]."""

    @staticmethod
    def forward(ctx, x, u, s0, row_mol, row0, nrows):
        L = _lib.lib()
        x, u = x.contiguous(), u.contiguous()
        s0 = None if s0 is None else s0.contiguous()
        N, d = x.shape
        s = _f(x.device, N)
        check(L.bmp_rowdot_fwd(ptr(x), d, ptr(u), ptr(s0), ptr(row_mol), N, ptr(s), stream()), "bmp_rowdot_fwd")
        ctx.save_for_backward(x, u, row_mol, row0, nrows)
        ctx.has_s0 = s0 is not None
        return s

    @staticmethod
    def backward(ctx, ds):
        L = _lib.lib()
        x, u, row_mol, row0, nrows = ctx.saved_tensors
        N, d = x.shape
        ds = ds.contiguous()
        dx = torch.empty_like(x)
        du = torch.zeros_like(u)
        ds0 = torch.zeros(u.shape[0], dtype=torch.float32, device=x.device) if ctx.has_s0 else None
        check(L.bmp_rowdot_bwd(ptr(ds), ptr(x), d, ptr(u), ptr(row_mol), ptr(row0), ptr(nrows), row0.numel(), N, ptr(dx),
                               ptr(du), ptr(ds0), stream()), "bmp_rowdot_bwd")
        return dx, du, ds0, None, None, None


class RowCorrFn(Function):
    """e[r] = circular correlation of row a[r] with q[mol(r)] (parallel_coattention.py:162-187)."""

    @staticmethod
    def forward(ctx, a, q, row0, nrows):
        L = _lib.lib()
        a, q = a.contiguous(), q.contiguous()
        N, o = a.shape
        if q.shape != (row0.numel(), o):
            raise ValueError(f"circular correlation: molecule vectors {tuple(q.shape)} vs rows of width {o}")
        e = torch.zeros(N, o, dtype=torch.float32, device=a.device)
        check(L.bmp_rowcorr_fwd(ptr(a), o, ptr(q), ptr(row0), ptr(nrows), row0.numel(), ptr(e), stream()), "bmp_rowcorr_fwd")
        ctx.save_for_backward(a, q, row0, nrows)
        return e

    @staticmethod
    def backward(ctx, de):
        L = _lib.lib()
        a, q, row0, nrows = ctx.saved_tensors
        de = de.contiguous()
        da, dq = torch.zeros_like(a), torch.empty_like(q)
        check(L.bmp_rowcorr_bwd(ptr(de), ptr(a), a.shape[1], ptr(q), ptr(row0), ptr(nrows), row0.numel(), ptr(da), ptr(dq),
                                stream()), "bmp_rowcorr_bwd")
        return da, dq, None, None


def _mol_linear(q: torch.Tensor, WT: torch.Tensor, b, act: int = 0) -> torch.Tensor:
    """Linear on a [M x k] matrix of molecule vectors through the row GEMM (rows padded to the tile size)."""
    R = _lib.lib().bmp_tile_rows()
    M, k = q.shape
    Mp = (M + R - 1) // R * R
    kp = (k + 7) // 8 * 8
    qp = torch.nn.functional.pad(q, (0, kp - k, 0, Mp - M))
    if kp != k:
        WT = torch.nn.functional.pad(WT, (0, 0, 0, kp - k))
    return LinearRowsFn.apply(qp.contiguous(), WT, b, act)[:M]


class _Side:
    """Rows, multiplicities and molecule ranges of the atoms a coarse module attends over."""

    def __init__(self, at: PackedAtoms):
        pb = at.pb
        self.X, self.w, self.row0, self.nrows, self.M = at.rows, pb.row_w, pb.mol_row0, pb.mol_nrows, pb.n_mols
        if "row_mol" not in pb._cache:
            rm = torch.full((pb.n_rows,), -1, dtype=torch.int32, device=pb.device)
            idx = torch.repeat_interleave(torch.arange(pb.n_mols, device=pb.device, dtype=torch.int32), pb.mol_nrows.long())
            off = torch.arange(int(pb.mol_nrows_host.sum()), device=pb.device) - \
                torch.repeat_interleave((torch.cumsum(pb.mol_nrows.long(), 0) - pb.mol_nrows.long()), pb.mol_nrows.long())
            rm[(torch.repeat_interleave(pb.mol_row0.long(), pb.mol_nrows.long()) + off)] = idx
            pb._cache["row_mol"] = rm
            pb._cache["atot"] = torch.zeros(pb.n_mols, device=pb.device).index_add_(
                0, idx.long(), pb.row_w[(torch.repeat_interleave(pb.mol_row0.long(), pb.mol_nrows.long()) + off)])
        self.row_mol, self.atot = pb._cache["row_mol"], pb._cache["atot"]

    def pool(self, A, Y):
        return SegPoolFn.apply(A, Y, self.w, self.row0, self.nrows)

    def mean(self):
        """F.mean(atoms, axis=1) over ALL padded positions."""
        ones = torch.ones(self.X.shape[0], 1, device=self.X.device)
        return self.pool(ones, self.X) / self.atot[:, None]

    def bcast(self, q):
        return RowBcastFn.apply(q, self.row_mol, self.row0, self.nrows)

    def dot(self, x, u, s0=None):
        return RowDotFn.apply(x, u, s0, self.row_mol, self.row0, self.nrows)

    def softmax(self, s):
        return SegSoftmaxFn.apply(s, self.w, self.row0, self.nrows)


def _sides(atoms_1, atoms_2):
    """(side over which module 'focus 1' attends, side for focus 2, per-molecule permutation to the OTHER molecule).
    One two-sided batch: a single _Side with other[m] = partner molecule; two batches: one _Side each."""
    from .ggnn import as_packed_atoms
    atoms_1, atoms_2 = as_packed_atoms(atoms_1), as_packed_atoms(atoms_2)          # dense (mb, N, hid) arrays accepted
    if atoms_1.pb is atoms_2.pb and len(atoms_1.pb.side_mols) == 3:
        return _Side(atoms_1), None
    return _Side(atoms_1), _Side(atoms_2)


def _split(s):
    """Two single-side views of a two-sided _Side (same rows; molecule ranges of one side each)."""
    B = s.M // 2
    out = []
    for sl in (slice(0, B), slice(B, 2 * B)):
        v = _Side.__new__(_Side)
        v.__dict__.update(s.__dict__)
        v.row0, v.nrows, v.M, v.atot = s.row0[sl].contiguous(), s.nrows[sl].contiguous(), B, s.atot[sl]
        v.row_mol = torch.where((s.row_mol >= sl.start) & (s.row_mol < sl.stop), s.row_mol - sl.start,
                                torch.full_like(s.row_mol, -1))
        out.append(v)
    return out


def _run(mod, atoms_1, g_1, atoms_2, g_2, sequential=False):
    """Both sides of a coarse module: ``mod._side(side, query per molecule, focus) -> compact per molecule``.
    Tied weights on a two-sided batch run both sides in one pass; ``sequential``: the side-2 query is the
    side-1 OUTPUT (alternating co-attention)."""
    s1, s2 = _sides(atoms_1, atoms_2)
    if s2 is None and not sequential and not mod.untied:
        B = s1.M // 2
        c = mod._side(s1, mod.query(s1, g_1, g_2), 0)
        return c[:B], c[B:]
    if s2 is None:
        s1, s2 = _split(s1)
    c1 = mod._side(s1, mod.query_single(s2, g_2), 1)
    c2 = mod._side(s2, c1 if sequential else mod.query_single(s1, g_1), 2)
    return c1, c2


class ParallelCoattention(nn.Module):
    def __init__(self, hidden_dim, out_dim, head, activation="tanh", weight_tying=True):
        super().__init__()
        if head != 1:
            raise ValueError("head must be 1 (the reference tiles the head-wide energy to out_dim, "
                             "parallel_coattention.py:45)")
        self.energy_layers = nn.ModuleList([Bilinear(hidden_dim, out_dim, head) for _ in range(1 if weight_tying else 2)])
        self.j_layer = Linear(hidden_dim, out_dim)
        self.hidden_dim, self.out_dim, self.head, self.weight_tying = hidden_dim, out_dim, head, weight_tying
        self.activation = getattr(activation, "__name__", activation) if callable(activation) else activation
        self.untied = not weight_tying

    def query(self, side, g_1, g_2):
        return torch.cat((g_2, g_1), dim=0)            # molecule m attends with the OTHER molecule's readout

    def query_single(self, other_side, g_other):
        return g_other

    def _side(self, side, q, focus):
        E = self.energy_layers[0 if self.weight_tying else max(focus - 1, 0)]
        u = _mol_linear(q, E.W[:, :, 0].t(), E.V1[:, 0])                  # W q + V1           [M x hidden]
        s0 = _mol_linear(q, E.V2, E.b)[:, 0]                                 # q . V2 + b         [M]
        e = _ACT[self.activation](side.dot(side.X, u, s0))                   # parallel_coattention.py:77
        J = LinearRowsFn.apply(side.X, self.j_layer.W.t(), self.j_layer.b, 0)
        return side.pool(e[:, None], J)                                      # :45-49 (no softmax)

    def forward(self, atoms_1, g_1, atoms_2, g_2, **_):
        return _run(self, atoms_1, g_1, atoms_2, g_2)


class CircularParallelCoattention(nn.Module):
    """parallel_coattention.py:87-187: gate_k = act(circular correlation of j_layer(atom) with the OTHER molecule's
    readout vector), compact = sum over atoms of gate * j_layer(atom).  Only j_layer has parameters; ``weight_tying`` is
    accepted and unused, as in the reference (:95-99)."""

    def __init__(self, hidden_dim, out_dim, activation="tanh", weight_tying=True):
        super().__init__()
        self.j_layer = Linear(hidden_dim, out_dim)
        self.hidden_dim, self.out_dim, self.head, self.weight_tying = hidden_dim, out_dim, out_dim, weight_tying
        self.activation = getattr(activation, "__name__", activation) if callable(activation) else activation
        self.untied = False

    def query(self, side, g_1, g_2):
        return torch.cat((g_2, g_1), dim=0)

    def query_single(self, other_side, g_other):
        return g_other

    def _side(self, side, q, focus):
        J = LinearRowsFn.apply(side.X, self.j_layer.W.t(), self.j_layer.b, 0)           # :115, :129
        e = _ACT[self.activation](RowCorrFn.apply(J, q, side.row0, side.nrows))           # :156
        return side.pool(e, J)                                                            # :119-121

    def forward(self, atoms_1, g_1, atoms_2, g_2, **_):
        return _run(self, atoms_1, g_1, atoms_2, g_2)


class AlternatingCoattention(nn.Module):
    def __init__(self, hidden_dim, out_dim, head, weight_tying=False):
        super().__init__()
        n = 1 if weight_tying else 2
        self.energy_layers_1 = nn.ModuleList([Linear(hidden_dim + out_dim, head) for _ in range(n)])
        self.energy_layers_2 = nn.ModuleList([Linear(head, 1)])             # a single layer, as in the reference (:24-26)
        self.j_layer = Linear(hidden_dim, out_dim)
        self.hidden_dim, self.out_dim, self.head, self.weight_tying = hidden_dim, out_dim, head, weight_tying
        self.untied = True

    def query_single(self, other_side, g_other):
        return g_other

    def _side(self, side, q, focus):
        li = 0 if self.weight_tying else focus - 1
        E1 = self.energy_layers_1[li]
        E2 = self.energy_layers_2[li]          # IndexError for focus 2 when not weight_tying: reference behaviour (:69)
        o = self.out_dim
        xe = LinearRowsFn.apply(side.X, E1.W[:, o:].t(), None, 0)             # key part of concat((query, key)) (:80)
        qe = _mol_linear(q, E1.W[:, :o].t(), E1.b)
        t = torch.tanh(xe + side.bcast(qe))
        hp = (self.head + 7) // 8 * 8
        tp = torch.nn.functional.pad(t, (0, hp - self.head)) if hp != self.head else t
        W2 = torch.nn.functional.pad(E2.W.t(), (0, 0, 0, hp - self.head)) if hp != self.head else E2.W.t()
        s = LinearRowsFn.apply(tp.contiguous(), W2, E2.b, 0)[:, 0]
        alpha = side.softmax(s)                                               # :85 softmax over atoms
        J = LinearRowsFn.apply(side.X, self.j_layer.W.t(), self.j_layer.b, 0)
        return side.pool(alpha[:, None], J)

    def forward(self, atoms_1, g_1, atoms_2, g_2, **_):
        return _run(self, atoms_1, g_1, atoms_2, g_2, sequential=True)       # :56 query = compact_1


class GlobalCoattention(nn.Module):
    def __init__(self, hidden_dim, out_dim, weight_tying=True):
        super().__init__()
        self.att_layers = nn.ModuleList([Linear(2 * hidden_dim, out_dim) for _ in range(1 if weight_tying else 2)])
        self.lt_layer = Linear(hidden_dim, out_dim)
        self.hidden_dim, self.out_dim, self.weight_tying = hidden_dim, out_dim, weight_tying
        self.untied = not weight_tying

    def query(self, side, g_1, g_2):
        mean = side.mean()
        B = side.M // 2
        return torch.cat((mean[B:], mean[:B]), dim=0)

    def query_single(self, other_side, g_other):
        return other_side.mean()                                             # :37-39 mean over atoms (g_k ignored)

    def _side(self, side, q, focus):
        A = self.att_layers[0 if self.weight_tying else max(focus - 1, 0)]
        d = self.hidden_dim
        xa = LinearRowsFn.apply(side.X, A.W[:, :d].t(), None, 0)              # concat((key, query)) (:70)
        qa = _mol_linear(q, A.W[:, d:].t(), A.b)
        attn = torch.sigmoid(xa + side.bcast(qa))
        lt = LinearRowsFn.apply(side.X, self.lt_layer.W.t(), self.lt_layer.b, 0)
        return side.pool(attn, lt)                                            # :44, :50

    def forward(self, atoms_1, g_1, atoms_2, g_2, **_):
        return _run(self, atoms_1, g_1, atoms_2, g_2)


class NeuralCoattention(nn.Module):
    def __init__(self, hidden_dim, out_dim, activation="relu", weight_tying=True):
        super().__init__()
        self.att_layers = nn.ModuleList([Linear(hidden_dim, out_dim) for _ in range(1 if weight_tying else 2)])
        self.hidden_dim, self.out_dim, self.weight_tying = hidden_dim, out_dim, weight_tying
        self.activation = getattr(activation, "__name__", activation) if callable(activation) else activation
        self.untied = not weight_tying

    def query(self, side, g_1, g_2):
        mean = side.mean()
        B = side.M // 2
        return torch.cat((mean[B:], mean[:B]), dim=0)

    def query_single(self, other_side, g_other):
        return other_side.mean()

    def _side(self, side, q, focus):
        A = self.att_layers[0 if self.weight_tying else max(focus - 1, 0)]
        act = _ACT[self.activation]
        context = act(_mol_linear(q, A.W.t(), A.b))                          # :65
        doc = LinearRowsFn.apply(side.X, A.W.t(), A.b, ACT.get(self.activation, 0))      # :67
        if self.activation not in ACT:
            raise ValueError(f"unsupported activation {self.activation!r}")
        energy = torch.sigmoid(side.dot(doc, context))                        # :69
        return side.pool(energy[:, None], doc)

    def forward(self, atoms_1, g_1, atoms_2, g_2, **_):
        return _run(self, atoms_1, g_1, atoms_2, g_2)

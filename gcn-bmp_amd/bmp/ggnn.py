"""GGNN encoder with the reference's constructor and call signatures.

Mirrors ``models.ggnn.GGNN`` (models/ggnn.py:19-654) and ``models.ggnn_att.GGNN``
(models/ggnn_att.py:39-664, which adds ``self.atoms`` / ``get_atom_array()``) on the default
path: message_function='matrix_multiply', readout_function='graph_level', no attention, no
layer aggregator, no context BiLSTM, no batch normalisation.  Any other option
raises NotImplementedError (they are research ablations outside SURVEY.md section 8).

``dropout_rate`` (models/ggnn.py:626-627): identity under ``eval()``; in training the zero-padded positions of a molecule
are ONE row of the packed layout and share one mask, where the reference draws a mask per padded position -- same
expectation, not the same random process; the float-feature input form keeps every position a row of its own.

Parameter names and shapes follow the reference link tree (embed.W, message_layers.{i}.W/b,
update_layer.{W_r,W_z,W,U_r,U_z,U}.W/b, i_layers.{k}.W/b, j_layers.{k}.W/b) so a Chainer
snapshot maps key by key.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import functional as Fn
from .packed import PackedMolBatch, pack_from_dense

MAX_ATOMIC_NUM = 117      # chainer_chemistry.config.MAX_ATOMIC_NUM (models/ggnn.py:14)
NUM_EDGE_TYPE = 4         # models/ggnn.py:37


class Linear(nn.Module):
    """Parameter holder with Chainer's Linear attribute names (W [out x in], b [out]) and
    default initialisers (W ~ LeCunNormal, b = 0)."""

    def __init__(self, in_size: Optional[int], out_size: int, nobias: bool = False):
        """``in_size=None`` is Chainer's ``Linear(None, out)``: W is created at the first call that shows the input
        width (models/mlp.py:34-37), here by ``materialize``."""
        super().__init__()
        if in_size is None:
            self.W = nn.UninitializedParameter()
        else:
            self.W = nn.Parameter(torch.randn(out_size, in_size) / math.sqrt(in_size))
        self.b = None if nobias else nn.Parameter(torch.zeros(out_size))
        self.in_size, self.out_size = in_size, out_size

    def materialize(self, in_size: int) -> None:
        if self.in_size is not None:
            if self.in_size != in_size:
                raise ValueError(f"Linear was built for {self.in_size} input features, got {in_size}")
            return
        dev = self.b.device if self.b is not None else None
        self.W.materialize((self.out_size, in_size), device=dev, dtype=torch.float32)
        with torch.no_grad():
            self.W.copy_(torch.randn(self.out_size, in_size) / math.sqrt(in_size))        # LeCunNormal, as Chainer's Linear
        self.in_size = in_size


class EmbedID(nn.Module):
    """chainer EmbedID / EmbedAtomID: W [in_size x out_size] ~ N(0, 1), no ignore label."""

    def __init__(self, out_size: int, in_size: int = MAX_ATOMIC_NUM):
        super().__init__()
        self.W = nn.Parameter(torch.randn(in_size, out_size))


class GRU(nn.Module):
    """Parameters of chainer.links.GRU(2d, d) (StatefulGRU): W_r, W_z, W [d x 2d]; U_r, U_z, U [d x d]."""

    def __init__(self, in_size: int, out_size: int):
        super().__init__()
        self.W_r = Linear(in_size, out_size); self.U_r = Linear(out_size, out_size)
        self.W_z = Linear(in_size, out_size); self.U_z = Linear(out_size, out_size)
        self.W = Linear(in_size, out_size); self.U = Linear(out_size, out_size)

    def kernel_weights(self, first: bool):
        """(AT [2d x 3d], UcT [d x d], b [3d]) in the layout of bmp_gru_fwd.  For later calls the
        state terms are folded into the h-part (the GRU state equals h without dropout):
        W_r x + U_r s = (W_r[:, :d] + U_r) h + W_r[:, d:] m  (SURVEY.md A.2)."""
        d = self.U.W.shape[0]
        A = torch.cat((self.W_r.W, self.W_z.W, self.W.W), dim=0)               # [3d x 2d]
        if first:
            b = torch.cat((self.W_r.b, self.W_z.b, self.W.b))
        else:
            fold = torch.cat((self.U_r.W, self.U_z.W, torch.zeros_like(self.U.W)), dim=0)   # [3d x d]
            A = A + torch.cat((fold, torch.zeros_like(fold)), dim=1)
            b = torch.cat((self.W_r.b + self.U_r.b, self.W_z.b + self.U_z.b, self.W.b + self.U.b))
        return A.t().contiguous(), self.U.W.t().contiguous(), b

    def kernel_weights_state(self):
        """(WT [2d x 3d], UrzT [d x 2d], UcT [d x d], b [3d]) in the layout of bmp_gru_state_fwd: the later-call form with
        the state terms NOT folded into the h-part (under dropout the state differs from the h half of x)."""
        WT = torch.cat((self.W_r.W, self.W_z.W, self.W.W), dim=0).t().contiguous()
        UrzT = torch.cat((self.U_r.W, self.U_z.W), dim=0).t().contiguous()
        b = torch.cat((self.W_r.b + self.U_r.b, self.W_z.b + self.U_z.b, self.W.b + self.U.b))
        return WT, UrzT, self.U.W.t().contiguous(), b


def message_kernel_weights(lin: Linear):
    """Reference GraphLinear(d_in, 4*d_out) with output feature k = 4*c + e
    (models/ggnn.py:223-224) -> WT [4*d_in x d_out] (row e*d_in + k, col c), bE [4 x d_out]."""
    o4, d_in = lin.W.shape
    d_out = o4 // NUM_EDGE_TYPE
    WT = lin.W.view(d_out, NUM_EDGE_TYPE, d_in).permute(1, 2, 0).reshape(NUM_EDGE_TYPE * d_in, d_out)
    bE = lin.b.view(d_out, NUM_EDGE_TYPE).t() if lin.b is not None else torch.zeros(NUM_EDGE_TYPE, d_out, device=lin.W.device)
    return WT.contiguous(), bE.contiguous()


class PackedAtoms:
    """What ``get_atom_array()`` hands to the co-attention: the per-row atom states of a
    packed batch.  ``dense()`` expands to the reference's (mb, A, hidden_dim) array."""

    def __init__(self, rows: torch.Tensor, pb: PackedMolBatch, side: Optional[int] = None):
        self.rows, self.pb, self.side = rows, pb, side

    def dense(self, side: Optional[int] = None) -> torch.Tensor:
        return self.pb.to_dense(self.rows, self.side if side is None else side)


_DENSE_PB_CACHE: dict = {}


def packed_atoms_from_dense(x: torch.Tensor) -> PackedAtoms:
    """The reference hands the co-attention dense atom arrays (mb, N, hidden) (nie_coattention.py:335-341) and masks
    nothing, so every position is an atom of weight 1: lay them out as packed rows (floor(128 / N) molecules per tile,
    no bonds, no virtual pad row) and let the pair kernels run unchanged.  Differentiable in ``x``."""
    from . import _lib
    if x.dim() != 3 or x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError("dense atom arrays must be float32 CUDA tensors of shape (mb, N, hidden_dim)")
    mb, N, d = x.shape
    R = _lib.lib().bmp_tile_rows()
    if N < 1:
        raise ValueError("dense atom arrays need at least one position per molecule")
    key = (mb, N, x.device)
    if key not in _DENSE_PB_CACHE:
        b = np.arange(mb)
        if N <= R:
            per = R // N
            n_tiles = (mb + per - 1) // per
            row0 = (b // per) * R + (b % per) * N
        else:                                   # more positions than a tile has rows: whole consecutive tiles per molecule
            per_mol = (N + R - 1) // R
            n_tiles = mb * per_mol
            row0 = b * per_mol * R
        idx = (row0[:, None] + np.arange(N)[None, :]).reshape(-1)
        n_rows = n_tiles * R
        row_w = np.zeros(n_rows, np.float32); row_w[idx] = 1.0
        row_mol = np.full(n_rows, -1, np.int32); row_mol[idx] = np.repeat(b, N)
        dev = x.device
        zi = torch.zeros(n_rows + 1, dtype=torch.int32, device=dev)
        ze = torch.zeros(4, dtype=torch.int32, device=dev)
        pb = PackedMolBatch(
            R=R, n_tiles=n_tiles, n_mols=mb, atom_id=zi[:n_rows], row_w=torch.from_numpy(row_w).to(dev), csr_ptr=zi, csr_col=ze[:0],
            csr_val=ze[:0].float(), csrT_ptr=zi, csrT_col=ze[:0], csrT_val=ze[:0].float(),
            mol_row0=torch.from_numpy(row0.astype(np.int32)).to(dev), mol_nrows=torch.full((mb,), N, dtype=torch.int32, device=dev),
            side_tiles=(0, n_tiles), side_mols=(0, mb), n_real_atoms=mb * N, n_edges=0, max_rows_per_mol=N,
            mol_nrows_host=np.full(mb, N, np.int64), row_mol=torch.from_numpy(row_mol).to(dev))
        pb.dense_map = torch.from_numpy(idx.reshape(mb, N)).to(dev)
        pb.dense_maps = [pb.dense_map]
        if len(_DENSE_PB_CACHE) >= 4:                # the two sides of a pair batch (+ one more batch shape) stay cached
            _DENSE_PB_CACHE.clear()
        _DENSE_PB_CACHE[key] = pb
    pb = _DENSE_PB_CACHE[key]
    rows = x.new_zeros(pb.n_rows, d).index_copy(0, pb.dense_map.reshape(-1), x.reshape(mb * N, d))
    return PackedAtoms(rows, pb, 0)


def as_packed_atoms(atoms) -> PackedAtoms:
    """What a co-attention module accepts in its ``atoms_k`` slots: the PackedAtoms of ``get_atom_array()`` or the
    reference's dense (mb, N, hidden_dim) array."""
    if isinstance(atoms, PackedAtoms):
        return atoms
    if isinstance(atoms, torch.Tensor):
        return packed_atoms_from_dense(atoms)
    raise TypeError("atoms must be the PackedAtoms returned by get_atom_array() or a dense (mb, N, hidden_dim) tensor")


def pack_float_atoms(x, adj, device):
    """Float atom features (mb, A, hidden_dim) bypass the embedding (models/ggnn.py:600-605).  No position can be told
    apart as padding there, so every position becomes a row of its own (nothing is merged into the virtual pad row,
    whose weight is then 0); returns the batch and the (n_rows, hidden_dim) row tensor, differentiable in ``x``."""
    xt = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    if xt.dim() != 3:
        raise ValueError("float atom features must have shape (mb, A, hidden_dim)")
    mb, A, d = xt.shape
    ones = np.ones((mb, A), dtype=np.int32)
    if isinstance(adj, torch.Tensor) and adj.is_cuda:
        from .packed import pack_from_dense_device
        pb = pack_from_dense_device([ones], [adj.detach().float().contiguous()])
    else:
        j = adj.detach().cpu().numpy() if isinstance(adj, torch.Tensor) else np.asarray(adj)
        pb = pack_from_dense([ones], [j.astype(np.float32)], device=device)
    xt = xt.to(device=device, dtype=torch.float32)
    rows = xt.new_zeros(pb.n_rows, d).index_copy(0, pb.dense_maps[0].reshape(-1), xt.reshape(mb * A, d))
    return pb, rows


def _is_float_atoms(atom_array) -> bool:
    if isinstance(atom_array, torch.Tensor):
        return atom_array.is_floating_point()
    return not isinstance(atom_array, PackedMolBatch) and np.asarray(atom_array).dtype.kind == "f"


def as_packed(atom_array, adj, device) -> PackedMolBatch:
    """Accept the reference's dense batch (atom_array (mb, A) int32, adj (mb, 4, A, A) float32,
    numpy or torch) or an already packed batch in the first slot."""
    if isinstance(atom_array, PackedMolBatch):
        return atom_array
    if isinstance(adj, torch.Tensor) and adj.is_cuda:        # dense device arrays: the adjacency never crosses PCIe
        a = atom_array.detach().cpu().numpy() if isinstance(atom_array, torch.Tensor) else np.asarray(atom_array)
        if a.dtype.kind not in "iu":
            raise NotImplementedError("float atom features (embedding bypass, models/ggnn.py:604-605) are not supported")
        from .packed import pack_from_dense_device
        return pack_from_dense_device([a.astype(np.int32)], [adj.detach().float().contiguous()])
    a = atom_array.detach().cpu().numpy() if isinstance(atom_array, torch.Tensor) else np.asarray(atom_array)
    j = adj.detach().cpu().numpy() if isinstance(adj, torch.Tensor) else np.asarray(adj)
    if a.dtype.kind not in "iu":
        raise NotImplementedError("float atom features (embedding bypass, models/ggnn.py:604-605) are not supported")
    return pack_from_dense([a.astype(np.int32)], [j.astype(np.float32)], device=device)


class GGNN(nn.Module):
    NUM_EDGE_TYPE = NUM_EDGE_TYPE

    def __init__(self, out_dim, hidden_dim=16, n_layers=4, n_atom_types=MAX_ATOMIC_NUM, concat_hidden=False,
                 layer_aggregator=None, dropout_rate=0.0, batch_normalization=False, weight_tying=True,
                 use_attention=False, update_attention=False, attention_tying=True,
                 context=False, context_layers=1, context_dropout=0.,
                 message_function='matrix_multiply', edge_hidden_dim=16,
                 readout_function='graph_level', num_timesteps=3,
                 num_output_hidden_layers=0, output_hidden_dim=16, output_activation=None,
                 output_atoms=False):
        super().__init__()
        unsupported = dict(layer_aggregator=layer_aggregator, batch_normalization=batch_normalization,
                           use_attention=use_attention, update_attention=update_attention, context=context)
        for k, v in unsupported.items():
            if v:
                raise NotImplementedError(f"GGNN option {k}={v!r} is outside the MI355X hot path (SURVEY.md 2.1 #1)")
        if not 0.0 <= dropout_rate < 1.0:
            raise ValueError("dropout_rate must lie in [0, 1)")
        self.dropout_rate = dropout_rate        # models/ggnn.py:626-627; see forward()
        if message_function != 'matrix_multiply':
            if message_function == 'edge_network':
                raise NotImplementedError("message_function='edge_network' is not supported")
            raise ValueError('There is no such message function named {}'.format(message_function))  # models/ggnn.py:250
        if readout_function != 'graph_level':
            raise NotImplementedError("readout_function='set2vec' is not supported")
        if hidden_dim % 8:
            raise ValueError("hidden_dim must be a multiple of 8 for the MFMA kernels")
        if out_dim % 4:
            raise ValueError("out_dim must be a multiple of 4")
        self.out_dim, self.hidden_dim, self.n_layers = out_dim, hidden_dim, n_layers
        self.concat_hidden, self.weight_tying = concat_hidden, weight_tying
        self.n_readout_layer = n_layers if concat_hidden else 1
        self.n_message_layer = 1 if weight_tying else n_layers
        self.embed = EmbedID(out_size=hidden_dim, in_size=n_atom_types)
        self.message_layers = nn.ModuleList(
            [Linear(hidden_dim, NUM_EDGE_TYPE * hidden_dim) for _ in range(self.n_message_layer)])
        self.update_layer = GRU(2 * hidden_dim, hidden_dim)
        self.i_layers = nn.ModuleList([Linear(2 * hidden_dim, out_dim) for _ in range(self.n_readout_layer)])
        self.j_layers = nn.ModuleList([Linear(hidden_dim, out_dim) for _ in range(self.n_readout_layer)])
        self.atoms = None
        self.fused = True      # use the fused per-tile step kernel where the width allows (64, 128)

    # models/ggnn.py:333-341: i sees [h, h0], j sees h only -> j's h0 rows are zero in the kernel layout
    def _readout_weights(self, k: int):
        i, j = self.i_layers[k], self.j_layers[k]
        d = self.hidden_dim
        WT = torch.cat((i.W.t(), torch.cat((j.W.t(), torch.zeros(d, self.out_dim, device=j.W.device, dtype=j.W.dtype)), dim=0)), dim=1)
        return WT.contiguous(), torch.cat((i.b, j.b))

    def readout(self, h, h0, pb, step=0):
        k = step if self.concat_hidden else 0
        WT, b = self._readout_weights(k)
        return Fn.ReadoutFn.apply(h, h0, WT, b, pb, Fn.ACT["identity"])

    # ---- layout plan protocol (bmp/plan.py): the weight-layout code above as pure functions of the parameters ----
    def plannable(self) -> bool:
        return self.fused and not self.concat_hidden and self.dropout_rate == 0.0

    # d = 32 (the reference's published width) has fused step kernels of its own (csrc/bmp_fused_small.hip); ``fused_small =
    # False`` sends that width through the unfused operators again (bench.py's A/B of the two, tests)
    fused_small = True

    def _plan_fused(self) -> bool:
        """The plan's arrays are those of the fused step kernels (d = 32 / 64 / 128) or of the unfused operators (other widths)."""
        return Fn.step_supported(self.hidden_dim) and (self.hidden_dim >= 64 or self.fused_small)

    def _step_groups(self):
        """(message layer, GRU mode) of every step, in step order (models/ggnn.py:220, first call after reset)."""
        return [((0 if self.weight_tying else s), ("first" if s == 0 else "later")) for s in range(self.n_layers)]

    def primary_layouts(self):
        out = {"embed.W": self.embed.W}
        for li, lin in enumerate(self.message_layers):
            out[f"msg{li}.WT"], out[f"msg{li}.bE"] = message_kernel_weights(lin)
        for mode in ("first", "later"):
            out[f"gru_{mode}.AT"], UcT, out[f"gru_{mode}.b"] = self.update_layer.kernel_weights(first=(mode == "first"))
        out["gru.UcT"] = UcT
        out["ro.WT"], out["ro.b"] = self._readout_weights(0)
        return out

    def prepared_layouts(self):
        p = self.primary_layouts()
        if not self._plan_fused():          # unfused operators: K-major operands and their transposes, nothing packed
            out = {"embed.W": p["embed.W"], "gru.UcT": p["gru.UcT"], "gru.Uc": p["gru.UcT"].t().contiguous(),
                   "ro.WT": p["ro.WT"], "ro.b": p["ro.b"], "ro.Wnat": p["ro.WT"].t().contiguous()}
            for li in range(self.n_message_layer):
                out[f"msg{li}.WT"], out[f"msg{li}.bE"] = p[f"msg{li}.WT"], p[f"msg{li}.bE"]
                out[f"msg{li}.Wnat"] = p[f"msg{li}.WT"].t().contiguous()
            for mode in ("first", "later"):
                out[f"gru_{mode}.AT"], out[f"gru_{mode}.b"] = p[f"gru_{mode}.AT"], p[f"gru_{mode}.b"]
                out[f"gru_{mode}.A"] = p[f"gru_{mode}.AT"].t().contiguous()
            return out
        out = {"embed.W": p["embed.W"], "gru.UcTp": Fn.pack_k4(p["gru.UcT"]), "gru.Uc_p": Fn.pack_k4(p["gru.UcT"].t()),
               "ro.WT": p["ro.WT"], "ro.b": p["ro.b"], "ro.Wnat": p["ro.WT"].t().contiguous()}
        if p["ro.WT"].shape[0] % 4 == 0:
            out["ro.WTp"] = Fn.pack_k4(p["ro.WT"])             # the tile-kernel form of the readout forward
        for li in range(self.n_message_layer):
            WT = p[f"msg{li}.WT"]
            out[f"msg{li}.WTp"], out[f"msg{li}.bE"], out[f"msg{li}.Wnat_p"] = Fn.pack_k4(WT), p[f"msg{li}.bE"], Fn.pack_k4(WT.t())
        for mode in ("first", "later"):
            AT = p[f"gru_{mode}.AT"]
            out[f"gru_{mode}.ATp"], out[f"gru_{mode}.b"], out[f"gru_{mode}.A_p"] = Fn.pack_k4(AT), p[f"gru_{mode}.b"], Fn.pack_k4(AT.t())
        return out

    def gk_spec(self):
        d, o = self.hidden_dim, self.out_dim
        spec = {"embed.dW": tuple(self.embed.W.shape), "ro.dWT": (2 * d, 2 * o), "ro.db": (2 * o,)}
        if not self._plan_fused():
            for li in range(self.n_message_layer):
                spec[f"msg{li}.dWT"], spec[f"msg{li}.dbE"] = (4 * d, d), (4, d)
            for mode in dict.fromkeys(m for _l, m in self._step_groups()):
                spec[f"gru_{mode}.dAT"], spec[f"gru_{mode}.dUcT"], spec[f"gru_{mode}.db"] = (2 * d, 3 * d), (d, d), (3 * d,)
            return spec
        for li, mode in dict.fromkeys(self._step_groups()):
            for k, shp in (("o1", (d, 7 * d)), ("o2", (d, 3 * d)), ("dUcT", (d, d)), ("cs", (7 * d,))):
                spec[f"g{li}_{mode}.{k}"] = shp
        return spec

    def primary_grads(self, gk):
        """Gradients of the primary layouts as lists of terms taken from the kernels' buffers (the same slicing
        GGNNStepFn.backward does)."""
        d = self.hidden_dim
        groups = list(dict.fromkeys(self._step_groups()))
        if not self._plan_fused():
            modes = list(dict.fromkeys(m for _l, m in groups))
            out = {"embed.W": [gk["embed.dW"]], "ro.WT": [gk["ro.dWT"]], "ro.b": [gk["ro.db"]],
                   "gru.UcT": [gk[f"gru_{m}.dUcT"] for m in modes if m == "later"]}
            for li in range(self.n_message_layer):
                out[f"msg{li}.WT"], out[f"msg{li}.bE"] = [gk[f"msg{li}.dWT"]], [gk[f"msg{li}.dbE"]]
            for mode in ("first", "later"):
                out[f"gru_{mode}.AT"] = [gk[f"gru_{mode}.dAT"]] if mode in modes else []
                out[f"gru_{mode}.b"] = [gk[f"gru_{mode}.db"]] if mode in modes else []
            return out
        out = {"embed.W": [gk["embed.dW"]], "ro.WT": [gk["ro.dWT"]], "ro.b": [gk["ro.db"]],
               "gru.UcT": [gk[f"g{li}_{mode}.dUcT"] for li, mode in groups if mode == "later"]}
        for li in range(self.n_message_layer):
            mine = [f"g{l}_{mode}" for l, mode in groups if l == li]
            out[f"msg{li}.WT"] = [gk[g + ".o1"][:, :4 * d].reshape(d, 4, d).permute(1, 0, 2).reshape(4 * d, d) for g in mine]
            out[f"msg{li}.bE"] = [gk[g + ".cs"][:4 * d].reshape(4, d) for g in mine]
        for mode in ("first", "later"):
            mine = [f"g{l}_{m}" for l, m in groups if m == mode]
            out[f"gru_{mode}.AT"] = [torch.cat((gk[g + ".o1"][:, 4 * d:], gk[g + ".o2"]), dim=0) for g in mine]
            out[f"gru_{mode}.b"] = [gk[g + ".cs"][4 * d:] for g in mine]
        return out

    def _forward_fast(self, pb, fast, h_in=None):
        """The encoder on the plan's prepared weights: embed, fused steps, readout -- no layout work, no weight
        gradients through autograd."""
        h, h0 = self._encode_fast(pb, fast, h_in)
        self.atoms = PackedAtoms(h, pb, 0 if pb.dense_map is not None else None)
        return self._readout_fast(h, h0, pb, fast)

    def _readout_fast(self, h, h0, pb, fast):
        P, G, state, _tape = fast
        infer = not torch.is_grad_enabled()
        if not self._plan_fused():
            return Fn.PReadoutFn.apply(h, h0, pb, dict(WT=P["ro.WT"], b=P["ro.b"], Wnat=P["ro.Wnat"]),
                                       dict(dWT=G["ro.dWT"], db=G["ro.db"]), Fn.ACT["identity"], state,
                                       getattr(self, "_readout_off_chain", False))
        return Fn.PReadoutFn.apply(h, h0, pb, dict(WT=P["ro.WT"], b=P["ro.b"], Wnat=P["ro.Wnat"], WTp=P.get("ro.WTp")),
                                   dict(dWT=G["ro.dWT"], db=G["ro.db"]), Fn.ACT["identity"], state,
                                   getattr(self, "_readout_off_chain", False), infer)

    def _encode_fast(self, pb, fast, h_in=None):
        """embed + propagation steps on the plan's prepared weights: (h after the last step, h0)."""
        P, G, state, tape = fast
        if h_in is None:
            pb.check_atom_ids(P["embed.W"].shape[0])
            h = Fn.PEmbedFn.apply(tape, P["embed.W"], pb.atom_id, G["embed.dW"], state)
        else:
            h = h_in
        h0 = h
        if not self._plan_fused():
            for step, (li, mode) in enumerate(self._step_groups()):
                Wm = dict(WT=P[f"msg{li}.WT"], bE=P[f"msg{li}.bE"], Wnat=P[f"msg{li}.Wnat"])
                m = Fn.PMsgFn.apply(h, pb, Wm, dict(dWT=G[f"msg{li}.dWT"], dbE=G[f"msg{li}.dbE"]), state, f"msg{li}", Fn.ACT["identity"])
                Wg = dict(AT=P[f"gru_{mode}.AT"], UcT=P["gru.UcT"], b=P[f"gru_{mode}.b"], A=P[f"gru_{mode}.A"], Uc=P["gru.Uc"])
                Gg = dict(dAT=G[f"gru_{mode}.dAT"], dUcT=G[f"gru_{mode}.dUcT"], db=G[f"gru_{mode}.db"])
                h = Fn.PGRUFn.apply(h, m, pb, Wg, Gg, state, f"gru_{mode}", step == 0)
            Fn._join_parts(state)             # the steps ran as two chains of tiles
            return h, h0
        # every step's outputs first, then the two chains of tiles are opened ONCE (Fn.fork_parts) and run to the join
        # without another cross-stream wait
        infer = not torch.is_grad_enabled()      # predict under no-backprop: nothing is kept for a backward
        bufs = [Fn.step_buffers(h.shape[0], self.hidden_dim, h.device, infer) for _ in range(self.n_layers)]
        Fn.fork_parts(state, pb)
        # (the per-step operand dictionaries are views of the plan's two buffers: built once per plan, not once per step)
        per_step = getattr(self, "_per_step", None)
        if per_step is None or per_step[0] is not P:
            lst = []
            for step, (li, mode) in enumerate(self._step_groups()):
                W = dict(WTp=P[f"msg{li}.WTp"], bE=P[f"msg{li}.bE"], Wnat_p=P[f"msg{li}.Wnat_p"], ATp=P[f"gru_{mode}.ATp"],
                         b=P[f"gru_{mode}.b"], A_p=P[f"gru_{mode}.A_p"], UcTp=P["gru.UcTp"], Uc_p=P["gru.Uc_p"])
                g = f"g{li}_{mode}"
                Gs = dict(o1=G[g + ".o1"], o2=G[g + ".o2"], dUcT=G[g + ".dUcT"], cs=G[g + ".cs"])
                lst.append((W, Gs, g, step == 0))
            per_step = (P, lst)
            object.__setattr__(self, "_per_step", per_step)
        per_step = per_step[1]
        for step, (W, Gs, g, first) in enumerate(per_step):
            h = Fn.PStepFn.apply(h, pb, W, Gs, state, g, first, bufs[step])
        Fn._join_parts(state)                 # the steps ran as two chains of tiles: whole arrays are read from here on
        return h, h0

    def encode_rows(self, pb: PackedMolBatch):
        """embed + the propagation steps (models/ggnn.py:599-627) WITHOUT the readout: (h, h0) on the rows of ``pb`` -- the
        entry the pair predictor uses for a batch in the encoder layout (bmp/enclayout.py), whose readout runs on the
        per-instance rows (``readout_rows``)."""
        if self.concat_hidden or (self.dropout_rate != 0.0 and self.training):
            raise NotImplementedError("encode_rows: concat_hidden / training dropout take the per-instance batch form")
        fast = getattr(self, "_fast", None)
        if fast is not None and not pb.oversized:
            return self._encode_fast(pb, fast)
        pb.check_atom_ids(self.embed.W.shape[0])
        h = Fn.EmbedFn.apply(self.embed.W, pb.atom_id)
        h0 = h
        fused = self.fused and self._plan_fused() and not pb.oversized
        later, msgw, cache = None, {}, {}
        for step in range(self.n_layers):
            li = 0 if self.weight_tying else step
            if li not in msgw:
                msgw[li] = message_kernel_weights(self.message_layers[li])
            WT, bE = msgw[li]
            if step == 0:
                AT, UcT, b = self.update_layer.kernel_weights(first=True)
            else:
                if later is None:
                    later = self.update_layer.kernel_weights(first=False)
                AT, UcT, b = later
            if fused:
                h = Fn.GGNNStepFn.apply(h, WT, bE, AT, UcT, b, pb, step == 0, cache)
            else:
                m = Fn.MsgFn.apply(h, WT, bE, None, None, pb, Fn.ACT["identity"])
                h = Fn.GRUFn.apply(h, m, AT, UcT, b, pb, step == 0)
        return h, h0

    def readout_rows(self, h, h0, pb: PackedMolBatch):
        """The gated-sum readout (models/ggnn.py:333-341) of row tensors that live on ``pb``'s rows."""
        fast = getattr(self, "_fast", None)
        if fast is not None and not pb.oversized:
            return self._readout_fast(h, h0, pb, fast)
        return self.readout(h, h0, pb, 0)

    def forward(self, atom_array, adj=None):
        """models/ggnn.py:584-654.  ``atom_array`` is the dense int32 (mb, A) array with ``adj``
        (mb, 4, A, A), or a PackedMolBatch (then ``adj`` is ignored).  Returns (n_mols, out_dim)
        [(n_mols, n_layers*out_dim) with concat_hidden]."""
        # F.dropout after every step (models/ggnn.py:626-627) drops the step OUTPUT while the stateful GRU keeps its own
        # un-dropped state: later steps see x = [dropout(s), m(dropout(s))] next to the state s -- the separate-state GRU
        # (bmp_gru_state_*); the fused kernels fold the state into the h-part and are not used then.  Under model.eval()
        # (chainer's train=False: evaluators, predict) dropout is the identity and the usual path runs.
        drop = self.dropout_rate != 0.0 and self.training
        dev = self.embed.W.device
        h_in = None
        if _is_float_atoms(atom_array):                                 # :604-605: float features skip the embedding
            pb, h_in = pack_float_atoms(atom_array, adj, dev)
            if h_in.shape[1] != self.hidden_dim:
                raise ValueError(f"float atom features must be hidden_dim={self.hidden_dim} wide, got {h_in.shape[1]}")
        else:
            pb = as_packed(atom_array, adj, dev)
        fast = getattr(self, "_fast", None)
        big = pb.oversized              # a molecule spans tiles (train_ddi_modify.py:256 sets no size limit): row-wise operators
        if fast is not None and not drop and not big:
            return self._forward_fast(pb, fast, h_in)
        if h_in is None:
            pb.check_atom_ids(self.embed.W.shape[0])
        h = Fn.EmbedFn.apply(self.embed.W, pb.atom_id) if h_in is None else h_in      # :603
        h0 = h                                                          # :612
        later = None
        fused = self.fused and self._plan_fused() and not drop and not big
        state, state_w = None, None           # dropout: the GRU's own (un-dropped) state and its unfolded weights
        masks = getattr(self, "_dropout_masks", None)      # tests inject the masks (one (n_rows, d) tensor per step)
        g_list = []
        msgw, cache = {}, {}          # per-call: kernel-layout weights of each layer, packed copies
        for step in range(self.n_layers):                               # :616
            li = 0 if self.weight_tying else step                       # :220
            if li not in msgw:
                msgw[li] = message_kernel_weights(self.message_layers[li])
            WT, bE = msgw[li]
            if step == 0:
                AT, UcT, b = self.update_layer.kernel_weights(first=True)
            else:
                if later is None:
                    later = self.update_layer.kernel_weights(first=False)
                AT, UcT, b = later
            if fused:       # message + GRU in one kernel per tile, atom states resident in LDS
                h = Fn.GGNNStepFn.apply(h, WT, bE, AT, UcT, b, pb, step == 0, cache)
            elif drop:
                m = Fn.MsgFn.apply(h, WT, bE, None, None, pb, Fn.ACT["identity"])
                if state is None:
                    state = Fn.GRUFn.apply(h, m, AT, UcT, b, pb, True)             # first call after reset: no state yet
                else:
                    if state_w is None:
                        state_w = self.update_layer.kernel_weights_state()
                    state = Fn.GRUStateFn.apply(h, m, state, *state_w, pb)
                if masks is not None:
                    h = state * masks[step]
                else:                                                              # :626-627, chainer: mask / (1 - ratio)
                    h = torch.nn.functional.dropout(state, p=self.dropout_rate, training=True)
            else:
                m = Fn.MsgFn.apply(h, WT, bE, None, None, pb, Fn.ACT["identity"])
                h = Fn.GRUFn.apply(h, m, AT, UcT, b, pb, step == 0)    # :254-262, state reset at :599
            if self.concat_hidden:
                g_list.append(self.readout(h, h0, pb, step))
        self.atoms = PackedAtoms(h, pb, 0 if pb.dense_map is not None else None)     # models/ggnn_att.py:651
        if self.concat_hidden:
            return torch.cat(g_list, dim=1)
        return self.readout(h, h0, pb, 0)

    def get_atom_array(self):
        """models/ggnn_att.py:662-664.  Returns a PackedAtoms; ``.dense()`` gives (mb, A, hidden_dim)."""
        assert self.atoms is not None
        return self.atoms

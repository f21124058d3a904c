"""RelGCN encoder and the modular GGNN built from update/readout blocks, with the reference's
signatures: ``models.relgcn.RelGCN`` (models/relgcn.py:31-73), ``models.update.RelGCNUpdate``
(models/update/relgcn_update.py:12-44), ``models.update.GGNNUpdate``
(models/update/ggnn_update.py:15-65), ``models.readout.GGNNReadout``
(models/readout/ggnn_readout.py:13-57) and ``models.models.ggnn.GGNN`` (models/models/ggnn.py:26-108).
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import functional as Fn
from .ggnn import (EmbedID, GRU, Linear, MAX_ATOMIC_NUM, NUM_EDGE_TYPE, PackedAtoms, as_packed,
                   message_kernel_weights)
from .packed import PackedMolBatch


def rescale_adj(pb: PackedMolBatch) -> PackedMolBatch:
    """models/relgcn.py:20-28 on the packed batch: every bond value is divided by the SOURCE
    atom's degree (sum of adj over bond types and destination rows; 0 -> 1).  Index work +
    one exact fp32 reciprocal-multiply per bond, same rounding as the reference's adj * (1/deg).
    One launch of bmp_rescale_adj on the device; host batches (CPU tests of the layout) take the same arithmetic in torch."""
    if "rescaled" in pb._cache:
        return pb._cache["rescaled"]
    N, E = pb.n_rows, pb.csr_col.numel()
    if pb.csr_val.is_cuda:
        from . import _lib
        from ._lib import check, ptr, stream
        vals = torch.empty(2, max(E, 1), dtype=torch.float32, device=pb.device)
        check(_lib.lib().bmp_rescale_adj(ptr(pb.csr_col), ptr(pb.csr_val), E, ptr(pb.csrT_ptr), ptr(pb.csrT_val), N,
                                         ptr(vals[0]), ptr(vals[1]), stream()), "bmp_rescale_adj")
        out = dataclasses.replace(pb, csr_val=vals[0, :E], csrT_val=vals[1, :E], _cache={})
    else:
        src = (pb.csr_col >> 2).long()
        deg = torch.zeros(N, dtype=torch.float32, device=pb.device).index_add_(0, src, pb.csr_val)
        inv = 1.0 / torch.where(deg != 0, deg, torch.ones_like(deg))
        rowT = torch.repeat_interleave(torch.arange(N, device=pb.device), (pb.csrT_ptr[1:] - pb.csrT_ptr[:-1]).long())
        out = dataclasses.replace(pb, csr_val=(pb.csr_val * inv[src]).contiguous(),
                                  csrT_val=(pb.csrT_val * inv[rowT]).contiguous(), _cache={})
    pb._cache["rescaled"] = out
    return out


class RelGCNUpdate(nn.Module):
    def __init__(self, in_channels, out_channels, num_edge_type=4):
        super().__init__()
        if num_edge_type != NUM_EDGE_TYPE:
            raise NotImplementedError("num_edge_type must be 4")
        if in_channels % 8 or out_channels % 8:
            raise ValueError("channel counts must be multiples of 8 for the MFMA kernels")
        self.graph_linear_self = Linear(in_channels, out_channels)
        self.graph_linear_edge = Linear(in_channels, out_channels * num_edge_type)
        self.num_edge_type, self.in_channels, self.out_channels = num_edge_type, in_channels, out_channels

    def forward(self, h, pb, act="identity"):
        """hs + sum_e adj_e (W_e h + b_e) (relgcn_update.py:24-44); ``act`` lets the caller fuse the
        tanh of models/relgcn.py:71 into the same kernel."""
        WT, bE = message_kernel_weights(self.graph_linear_edge)
        if h.is_cuda and not pb.oversized and Fn.rel_layer_supported(self.in_channels, self.out_channels):
            return Fn.RelLayerFn.apply(h, WT, bE, self.graph_linear_self.W.t(), self.graph_linear_self.b, pb, Fn.ACT[act])
        return Fn.MsgFn.apply(h, WT, bE, self.graph_linear_self.W.t(), self.graph_linear_self.b, pb, Fn.ACT[act])


class GGNNUpdate(nn.Module):
    """One message linear + its OWN stateful GRU (ggnn_update.py:25-29)."""

    def __init__(self, hidden_dim=16, num_edge_type=4):
        super().__init__()
        if num_edge_type != NUM_EDGE_TYPE:
            raise NotImplementedError("num_edge_type must be 4")
        self.graph_linear = Linear(hidden_dim, num_edge_type * hidden_dim)
        self.update_layer = GRU(2 * hidden_dim, hidden_dim)
        self.num_edge_type = num_edge_type
        self._calls = 0

    def reset_state(self):
        self._calls = 0

    def forward(self, h, pb):
        WT, bE = message_kernel_weights(self.graph_linear)
        m = Fn.MsgFn.apply(h, WT, bE, None, None, pb, Fn.ACT["identity"])
        first = self._calls == 0
        AT, UcT, b = self.update_layer.kernel_weights(first)
        self._calls += 1
        return Fn.GRUFn.apply(h, m, AT, UcT, b, pb, first)


class GGNNReadout(nn.Module):
    def __init__(self, out_dim, hidden_dim=16, nobias=False, activation="identity", activation_agg="identity",
                 in_dim: Optional[int] = None):
        """``in_dim`` replaces Chainer's lazy GraphLinear(None, out_dim): hidden_dim when called
        without h0, 2*hidden_dim with h0."""
        super().__init__()
        in_dim = hidden_dim if in_dim is None else in_dim
        self.i_layer = Linear(in_dim, out_dim, nobias=nobias)
        self.j_layer = Linear(in_dim, out_dim, nobias=nobias)
        self.out_dim, self.hidden_dim, self.nobias = out_dim, hidden_dim, nobias
        self.activation, self.activation_agg = activation, activation_agg

    def forward(self, h, pb, h0=None, row_w=None):
        WT = torch.cat((self.i_layer.W.t(), self.j_layer.W.t()), dim=1).contiguous()
        b = None if self.nobias else torch.cat((self.i_layer.b, self.j_layer.b))
        if row_w is not None:
            pb = dataclasses.replace(pb, row_w=row_w, _cache={})
        g = Fn.ReadoutFn.apply(h, h0, WT, b, pb, Fn.ACT[self.activation])
        if self.activation_agg == "tanh":
            g = torch.tanh(g)
        elif self.activation_agg == "sigmoid":
            g = torch.sigmoid(g)
        elif self.activation_agg == "relu":
            g = torch.relu(g)
        return g


class RelGCN(nn.Module):
    def __init__(self, out_channels=64, num_edge_type=4, ch_list=None, n_atom_types=MAX_ATOMIC_NUM, input_type='int',
                 scale_adj=None):
        super().__init__()
        if ch_list is None:
            ch_list = [16, 128, 64]                                            # models/relgcn.py:37
        if input_type == 'int':
            self.embed = EmbedID(out_size=ch_list[0], in_size=n_atom_types)
        elif input_type == 'float':
            self.embed = Linear(None, ch_list[0])                              # GraphLinear(None, ch_list[0]), models/relgcn.py:42-43
        else:
            raise ValueError("[ERROR] Unexpected value input type={}".format(input_type))
        self.rgcn_convs = nn.ModuleList([RelGCNUpdate(ch_list[i], ch_list[i + 1], num_edge_type)
                                         for i in range(len(ch_list) - 1)])
        self.rgcn_readout = GGNNReadout(out_dim=out_channels, hidden_dim=ch_list[-1], nobias=True, activation="tanh")
        self.input_type, self.scale_adj = input_type, scale_adj
        self.out_dim, self.hidden_dim, self.n_layers = out_channels, ch_list[-1], len(ch_list) - 1
        self.atoms = None

    # ---- layout plan protocol (bmp/plan.py) ----
    def plannable(self) -> bool:
        return self.input_type == 'int'          # (the float form's embedding is a lazily sized GraphLinear)

    def primary_layouts(self):
        out = {"embed.W": self.embed.W}
        for l, conv in enumerate(self.rgcn_convs):
            out[f"c{l}.WT"], out[f"c{l}.bE"] = message_kernel_weights(conv.graph_linear_edge)
            out[f"c{l}.WsT"] = conv.graph_linear_self.W.t().contiguous()
            out[f"c{l}.bs"] = conv.graph_linear_self.b
        ro = self.rgcn_readout
        out["ro.WT"] = torch.cat((ro.i_layer.W.t(), ro.j_layer.W.t()), dim=1).contiguous()
        return out

    def _fused(self, l) -> bool:
        """Layer l runs as the fused tile kernel (d_in == d_out in {64, 128}); otherwise bmp_msg_fwd/bwd."""
        conv = self.rgcn_convs[l]
        return Fn.rel_layer_supported(conv.in_channels, conv.out_channels)

    def prepared_layouts(self):
        p = self.primary_layouts()
        out = dict(p)
        for l in range(len(self.rgcn_convs)):
            if self._fused(l):
                WT, WsT = p[f"c{l}.WT"], p[f"c{l}.WsT"]
                out[f"c{l}.WTp"], out[f"c{l}.WsTp"] = Fn.pack_k4(WT), Fn.pack_k4(WsT)
                out[f"c{l}.Wnat_p"], out[f"c{l}.Ws_p"] = Fn.pack_k4(WT.t()), Fn.pack_k4(WsT.t())
            else:
                out[f"c{l}.Wnat"] = p[f"c{l}.WT"].t().contiguous()
                out[f"c{l}.Ws"] = p[f"c{l}.WsT"].t().contiguous()
        out["ro.Wnat"] = p["ro.WT"].t().contiguous()
        if p["ro.WT"].shape[0] % 4 == 0:
            out["ro.WTp"] = Fn.pack_k4(p["ro.WT"])
        return out

    def gk_spec(self):
        spec = {"embed.dW": tuple(self.embed.W.shape)}
        for l, conv in enumerate(self.rgcn_convs):
            di, do = conv.in_channels, conv.out_channels
            if self._fused(l):
                spec.update({f"c{l}.o1": (di, 5 * di), f"c{l}.dbE": (4, do), f"c{l}.cs": (5 * di,)})
            else:
                spec.update({f"c{l}.dWT": (4 * di, do), f"c{l}.dbE": (4, do), f"c{l}.dWsT": (di, do), f"c{l}.dbs": (do,)})
        spec["ro.dWT"] = (self.hidden_dim, 2 * self.out_dim)
        return spec

    def primary_grads(self, gk):
        out = {"embed.W": [gk["embed.dW"]], "ro.WT": [gk["ro.dWT"]]}
        for l, conv in enumerate(self.rgcn_convs):
            if self._fused(l):
                d = conv.in_channels
                o1, cs = gk[f"c{l}.o1"], gk[f"c{l}.cs"]
                out[f"c{l}.WT"] = [o1[:, :4 * d].reshape(d, 4, d).permute(1, 0, 2).reshape(4 * d, d)]
                out[f"c{l}.WsT"] = [o1[:, 4 * d:]]
                out[f"c{l}.bE"] = [gk[f"c{l}.dbE"]]
                out[f"c{l}.bs"] = [cs[4 * d:]]
            else:
                for a, b in (("WT", "dWT"), ("bE", "dbE"), ("WsT", "dWsT"), ("bs", "dbs")):
                    out[f"c{l}.{a}"] = [gk[f"c{l}.{b}"]]
        return out

    def _forward_fast(self, pb, fast):
        x, _ = self._encode_fast(pb, fast)
        self.atoms = PackedAtoms(x, pb, 0 if pb.dense_map is not None else None)
        return self._readout_fast(x, pb, fast)

    def _readout_fast(self, x, pb, fast):
        P, G, state, _tape = fast
        return Fn.PReadoutFn.apply(x, None, pb, dict(WT=P["ro.WT"], Wnat=P["ro.Wnat"], WTp=P.get("ro.WTp")), dict(dWT=G["ro.dWT"]),
                                   Fn.ACT["tanh"], state, getattr(self, "_readout_off_chain", False), not torch.is_grad_enabled())

    def encode_rows(self, pb):
        """embed + the layers (models/relgcn.py:67-71) WITHOUT the readout: (x, None) on the rows of ``pb`` (the entry the pair
        predictor uses for a batch in the encoder layout, bmp/enclayout.py)."""
        if self.input_type != 'int':
            raise NotImplementedError("encode_rows needs integer atom ids")
        fast = getattr(self, "_fast", None)
        if fast is not None and not pb.oversized:
            return self._encode_fast(pb, fast)
        pb.check_atom_ids(self.embed.W.shape[0])
        x = Fn.EmbedFn.apply(self.embed.W, pb.atom_id)
        pbs = rescale_adj(pb) if self.scale_adj else pb
        for conv in self.rgcn_convs:
            x = conv(x, pbs, act="tanh")
        return x, None

    def readout_rows(self, x, h0, pb):
        fast = getattr(self, "_fast", None)
        if fast is not None and not pb.oversized:
            return self._readout_fast(x, pb, fast)
        return self.rgcn_readout(x, pb)

    def _encode_fast(self, pb, fast):
        P, G, state, tape = fast
        pb.check_atom_ids(P["embed.W"].shape[0])
        x = Fn.PEmbedFn.apply(tape, P["embed.W"], pb.atom_id, G["embed.dW"], state)
        pbs = rescale_adj(pb) if self.scale_adj else pb
        all_fused = all(self._fused(l) for l in range(len(self.rgcn_convs)))
        bufs = None
        if all_fused:         # every layer's outputs first, then the chains of tiles opened once (Fn.fork_parts)
            infer = not torch.is_grad_enabled()      # predict under no-backprop: nothing is kept for a backward
            bufs = [Fn.rel_buffers(x.shape[0], self.rgcn_convs[l].out_channels, x.device, infer) for l in range(len(self.rgcn_convs))]
            Fn.fork_parts(state, pb)
        for l in range(len(self.rgcn_convs)):
            if self._fused(l):
                W = {k: P[f"c{l}.{k}"] for k in ("WTp", "bE", "WsTp", "bs", "Wnat_p", "Ws_p")}
                Gl = {k: G[f"c{l}.{k}"] for k in ("o1", "dbE", "cs")}
                x = Fn.PRelLayerFn.apply(x, pbs, W, Gl, state, f"c{l}", Fn.ACT["tanh"], None if bufs is None else bufs[l])
                continue
            W = {k: P[f"c{l}.{k}"] for k in ("WT", "bE", "WsT", "bs", "Wnat", "Ws")}
            Gl = {k: G[f"c{l}.{k}"] for k in ("dWT", "dbE", "dWsT", "dbs")}
            Fn._join_parts(state)             # an unfused layer reads whole arrays
            x = Fn.PMsgFn.apply(x, pbs, W, Gl, state, f"c{l}", Fn.ACT["tanh"])
        Fn._join_parts(state)                 # fused layers ran as two chains of tiles
        return x, None

    def forward(self, h, adj=None):
        """models/relgcn.py:61-73."""
        if self.input_type == 'float':
            # float atom features (mb, A, f): every position is a row of its own (nothing can be told apart as padding);
            # the embedding is a GraphLinear applied to every position, padded ones included (they get its bias)
            from .ggnn import pack_float_atoms
            dev = self.rgcn_convs[0].graph_linear_self.W.device
            pb, rows = pack_float_atoms(h, adj, dev)
            self.embed.materialize(rows.shape[1])
            k = rows.shape[1]
            kp = (k + 7) // 8 * 8
            WT = self.embed.W.t()
            if kp != k:
                rows = torch.nn.functional.pad(rows, (0, kp - k))
                WT = torch.nn.functional.pad(WT, (0, 0, 0, kp - k))
            x = Fn.LinearRowsFn.apply(rows.contiguous(), WT.contiguous(), self.embed.b, Fn.ACT["identity"])
        else:
            pb = as_packed(h, adj, self.embed.W.device)
            fast = getattr(self, "_fast", None)
            if fast is not None and not pb.oversized:         # (a molecule spanning tiles: row-wise operators below)
                return self._forward_fast(pb, fast)
            pb.check_atom_ids(self.embed.W.shape[0])
            x = Fn.EmbedFn.apply(self.embed.W, pb.atom_id)
        pbs = rescale_adj(pb) if self.scale_adj else pb
        for conv in self.rgcn_convs:
            x = conv(x, pbs, act="tanh")                                       # :70-71
        self.atoms = PackedAtoms(x, pb, 0 if pb.dense_map is not None else None)
        return self.rgcn_readout(x, pb)

    def get_atom_array(self):
        """Not in the reference: last-layer atom states, so that RelGCN composes
        with the co-attention as BASELINE.json config 3 requires."""
        assert self.atoms is not None
        return self.atoms


class GGNNModular(nn.Module):
    """models/models/ggnn.py:26-108: GGNN assembled from GGNNUpdate / GGNNReadout blocks.  Each
    update layer owns its GRU, so with weight_tying=False every GRU call is a first call."""

    def __init__(self, out_dim, hidden_dim=16, n_layers=4, n_atom_types=MAX_ATOMIC_NUM, concat_hidden=False,
                 weight_tying=True, activation="identity", num_edge_type=4):
        super().__init__()
        n_readout_layer = n_layers if concat_hidden else 1
        n_message_layer = 1 if weight_tying else n_layers
        self.embed = EmbedID(out_size=hidden_dim, in_size=n_atom_types)
        self.update_layers = nn.ModuleList([GGNNUpdate(hidden_dim, num_edge_type) for _ in range(n_message_layer)])
        self.readout_layers = nn.ModuleList([
            GGNNReadout(out_dim=out_dim, hidden_dim=hidden_dim, activation=activation, activation_agg=activation,
                        in_dim=2 * hidden_dim) for _ in range(n_readout_layer)])
        self.out_dim, self.hidden_dim, self.n_layers = out_dim, hidden_dim, n_layers
        self.concat_hidden, self.weight_tying = concat_hidden, weight_tying
        self.atoms = None

    def reset_state(self):
        for u in self.update_layers:
            u.reset_state()

    def forward(self, atom_array, adj=None, is_real_node=None):
        pb = as_packed(atom_array, adj, self.embed.W.device)
        row_w = None
        if is_real_node is not None:
            # mask (mb, A) -> per-row weight: a virtual row carries the sum of its positions' masks
            if pb.dense_map is None:
                raise NotImplementedError("is_real_node needs the dense input form")
            m = torch.as_tensor(np.asarray(is_real_node), dtype=torch.float32, device=pb.device)
            row_w = torch.zeros(pb.n_rows, device=pb.device).index_add_(0, pb.dense_map.reshape(-1), m.reshape(-1))
        self.reset_state()                                                       # models/models/ggnn.py:87
        pb.check_atom_ids(self.embed.W.shape[0])
        h = Fn.EmbedFn.apply(self.embed.W, pb.atom_id)
        h0 = h
        g_list = []
        for step in range(self.n_layers):
            li = 0 if self.weight_tying else step
            h = self.update_layers[li](h, pb)
            if self.concat_hidden:
                g_list.append(self.readout_layers[step](h, pb, h0, row_w))
        self.atoms = PackedAtoms(h, pb, 0 if pb.dense_map is not None else None)
        if self.concat_hidden:
            return torch.cat(g_list, dim=1)
        return self.readout_layers[0](h, pb, h0, row_w)

    def get_atom_array(self):
        assert self.atoms is not None
        return self.atoms

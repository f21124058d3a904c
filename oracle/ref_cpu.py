"""CPU oracle for the GCN-BMP paired-molecule hot path.  TEST INFRASTRUCTURE ONLY.

This file is a dense, op-for-op torch-CPU restatement of the reference's
(Minys233/GCN-BMP) Chainer model code for the hot path named in BASELINE.json.
It exists to *check* the HIP path; nothing under ``gcn-bmp_amd/`` may import
it.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for
this path and cannot be imported here (chainer,
chainer_chemistry, rdkit are absent; the code is Python-2 only).  The oracle is
therefore pinned only by (a) line-by-line restatement of the repo's own model
files, cited below, (b) the third-party layer semantics restated in SURVEY.md
Appendix B (Chainer ``Linear``/``EmbedID``/``StatefulGRU``/``Bilinear``/
``softmax``; chainer-chemistry ``GraphLinear``/``concat_mols``) and (c) the
algebraic known-answer tests in ``tests/test_oracle.py``.

Everything keeps the reference's *dense* formulation on purpose: the
(mb, 4, A, A) adjacency batched matmul, the materialised (mb, A, 4d) message
tensor, and -- crucially -- the absence of any padding mask: zero-padded atoms
carry the id-0 embedding through the GRU, the readout sum and every
co-attention softmax (models/ggnn.py:340,603; nie_coattention.py:347-349).

All functions take a ``dict`` of torch tensors named after the reference's link
tree (``embed/W``, ``message_layers/0/W``, ``update_layer/W_r/W`` ...), so a
Chainer ``.npz`` snapshot key maps 1:1.  dtype follows the parameters
(float64 for golden vectors, float32 for the timed CPU baseline).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]

NUM_EDGE_TYPE = 4          # models/ggnn.py:37
MAX_ATOMIC_NUM = 117       # chainer_chemistry.config.MAX_ATOMIC_NUM (models/ggnn.py:14)


# --------------------------------------------------------------------------- #
# third-party layer semantics
# --------------------------------------------------------------------------- #
def linear(x: Tensor, W: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """chainer.links.Linear / chainer_chemistry GraphLinear: y = x W^T + b on the
    last axis (GraphLinear reshapes (s0*s1, s2) -> Linear -> back)."""
    y = x @ W.t()
    if b is not None:
        y = y + b
    return y


def bilinear(e1: Tensor, e2: Tensor, W: Tensor, V1: Tensor, V2: Tensor, b: Tensor) -> Tensor:
    """chainer.links.Bilinear(l, r, o): y = e1^T W e2 + e1 V1 + e2 V2 + b.
    W (l, r, o), V1 (l, o), V2 (r, o), b (o,).  e1 (n, l), e2 (n, r) -> (n, o)."""
    y = torch.einsum("ni,ijk,nj->nk", e1, W, e2)
    return y + e1 @ V1 + e2 @ V2 + b


def stateful_gru(p: Params, prefix: str, x: Tensor, s: Optional[Tensor]) -> Tensor:
    """chainer.links.GRU == StatefulGRU.__call__ (models/ggnn.py:132,260).

    First call after reset_state() (s is None): z = sigmoid(W_z x),
    h_bar = tanh(W x), h' = z * h_bar -- no U terms and no U biases.
    Later calls: r = sigmoid(W_r x + U_r s), z = sigmoid(W_z x + U_z s),
    h_bar = tanh(W x + U (r*s)), h' = z*h_bar + (1-z)*s  (linear_interpolate)."""
    g = lambda n: (p[f"{prefix}/{n}/W"], p[f"{prefix}/{n}/b"])
    z = linear(x, *g("W_z"))
    h_bar = linear(x, *g("W"))
    if s is not None:
        r = torch.sigmoid(linear(x, *g("W_r")) + linear(s, *g("U_r")))
        z = z + linear(s, *g("U_z"))
        h_bar = h_bar + linear(r * s, *g("U"))
    z = torch.sigmoid(z)
    h_bar = torch.tanh(h_bar)
    if s is not None:
        return z * h_bar + (1 - z) * s
    return z * h_bar


# --------------------------------------------------------------------------- #
# GGNN (models/ggnn.py, models/ggnn_att.py)
# --------------------------------------------------------------------------- #
def ggnn_message(h: Tensor, adj: Tensor, W: Tensor, b: Tensor) -> Tensor:
    """models/ggnn.py:215-243 (same body models/update/ggnn_update.py:31-50).

    m = GraphLinear(h) reshaped (mb, atom, ch, 4) -- edge type is the FASTEST
    axis of the 4d output, i.e. feature k = 4*c + e -- transposed to
    (mb, 4, atom, ch), batched matmul with adj (mb*4, atom, atom), summed over
    edge types."""
    mb, atom, ch = h.shape
    m = linear(h, W, b).reshape(mb, atom, ch, NUM_EDGE_TYPE)
    m = m.permute(0, 3, 1, 2).reshape(mb * NUM_EDGE_TYPE, atom, ch)
    a = adj.reshape(mb * NUM_EDGE_TYPE, atom, atom)
    m = torch.bmm(a, m).reshape(mb, NUM_EDGE_TYPE, atom, ch)
    return m.sum(dim=1)


def ggnn_readout(h: Tensor, h0: Tensor, Wi: Tensor, bi: Tensor, Wj: Tensor, bj: Tensor) -> Tensor:
    """models/ggnn.py:333-341: g = sum_atoms sigmoid(i([h,h0])) * j(h); the sum
    runs over ALL padded positions."""
    g = torch.sigmoid(linear(torch.cat((h, h0), dim=2), Wi, bi)) * linear(h, Wj, bj)
    return g.sum(dim=1)


def ggnn_forward(p: Params, atom_array: Tensor, adj: Tensor, n_layers: int,
                 weight_tying: bool = True, concat_hidden: bool = False,
                 prefix: str = "", dropout_masks: Optional[Sequence[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """models/ggnn.py:584-654 / models/ggnn_att.py:589-664 default path
    (message_function='matrix_multiply', readout_function='graph_level', no
    attention / aggregator / context / BN).  ``dropout_masks``: one (mb, atom, ch) multiplier per step in place of
    F.dropout's random draw (:626-627).

    Returns (g, atoms) where atoms = h_T is what ggnn_att's get_atom_array()
    hands to the co-attention (models/ggnn_att.py:651,662-664)."""
    P = lambda k: p[prefix + k]
    if atom_array.dtype in (torch.int32, torch.int64):
        h = P("embed/W")[atom_array.long()]            # :603 EmbedAtomID
    else:
        h = atom_array                                  # :605
    h0 = h.clone()                                      # :612
    mb, atom, ch = h.shape
    s = None                                            # :599 reset_state()
    g_list = []
    sp = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix + "update_layer/")}
    for step in range(n_layers):                        # :616
        li = 0 if weight_tying else step                # :220
        m = ggnn_message(h, adj, P(f"message_layers/{li}/W"), P(f"message_layers/{li}/b"))
        x = torch.cat((h.reshape(mb * atom, ch), m.reshape(mb * atom, ch)), dim=1)   # :254-260
        s = stateful_gru(sp, "update_layer", x, s)
        h = s.reshape(mb, atom, ch)                     # :262
        if dropout_masks is not None:                   # :626-627 F.dropout(h): h * mask, mask = (rand >= ratio) / (1 - ratio);
            h = h * dropout_masks[step]                 #          the GRU's state s stays un-dropped (stateful link)
        if concat_hidden:                               # :629-635
            g_list.append(ggnn_readout(h, h0, P(f"i_layers/{step}/W"), P(f"i_layers/{step}/b"),
                                       P(f"j_layers/{step}/W"), P(f"j_layers/{step}/b")))
    if concat_hidden:
        return torch.cat(g_list, dim=1), h              # :646-647
    g = ggnn_readout(h, h0, P("i_layers/0/W"), P("i_layers/0/b"), P("j_layers/0/W"), P("j_layers/0/b"))
    return g, h                                         # :653


# --------------------------------------------------------------------------- #
# modular GGNN (models/models/ggnn.py + update/ggnn_update.py + readout/ggnn_readout.py)
# --------------------------------------------------------------------------- #
ACT = {"identity": lambda x: x, "tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}


def ggnn_readout_block(p: Params, prefix: str, h: Tensor, h0: Optional[Tensor], nobias: bool,
                       activation: str = "identity", activation_agg: str = "identity",
                       is_real_node: Optional[Tensor] = None) -> Tensor:
    """models/readout/ggnn_readout.py:42-57: both i and j see [h,h0] (or h if h0
    is None); g2 = act(j(.)); optional mask; g = act_agg(sum)."""
    h1 = torch.cat((h, h0), dim=2) if h0 is not None else h
    bi = None if nobias else p[f"{prefix}/i_layer/b"]
    bj = None if nobias else p[f"{prefix}/j_layer/b"]
    g1 = torch.sigmoid(linear(h1, p[f"{prefix}/i_layer/W"], bi))
    g2 = ACT[activation](linear(h1, p[f"{prefix}/j_layer/W"], bj))
    g = g1 * g2
    if is_real_node is not None:
        g = g * is_real_node[:, :, None].to(g.dtype)
    return ACT[activation_agg](g.sum(dim=1))


def ggnn_modular_forward(p: Params, atom_array: Tensor, adj: Tensor, n_layers: int,
                         weight_tying: bool = True, concat_hidden: bool = False,
                         activation: str = "identity",
                         is_real_node: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """models/models/ggnn.py:72-108.  Each GGNNUpdate owns its GRU
    (update/ggnn_update.py:28) and all are reset at :87, so with
    weight_tying=False every GRU call takes the first-call branch."""
    h = p["embed/W"][atom_array.long()] if atom_array.dim() <= 2 else atom_array
    h0 = h.clone()
    mb, atom, ch = h.shape
    states: Dict[int, Optional[Tensor]] = {}
    g_list = []
    for step in range(n_layers):
        li = 0 if weight_tying else step
        pre = f"update_layers/{li}"
        m = ggnn_message(h, adj, p[f"{pre}/graph_linear/W"], p[f"{pre}/graph_linear/b"])
        x = torch.cat((h.reshape(mb * atom, ch), m.reshape(mb * atom, ch)), dim=1)
        sp = {k[len(pre) + 1:]: v for k, v in p.items() if k.startswith(pre + "/update_layer/")}
        s = stateful_gru(sp, "update_layer", x, states.get(li))
        states[li] = s
        h = s.reshape(mb, atom, ch)
        if concat_hidden:
            g_list.append(ggnn_readout_block(p, f"readout_layers/{step}", h, h0, False,
                                             activation, activation, is_real_node))
    if concat_hidden:
        return torch.cat(g_list, dim=1), h
    return ggnn_readout_block(p, "readout_layers/0", h, h0, False, activation, activation,
                              is_real_node), h


# --------------------------------------------------------------------------- #
# RelGCN (models/relgcn.py, models/update/relgcn_update.py)
# --------------------------------------------------------------------------- #
def rescale_adj(adj: Tensor) -> Tensor:
    """models/relgcn.py:20-28: column-degree normalisation; the degree sums over
    edge types (axis 1) and the row index (axis 2); 0 -> 1."""
    num_neighbor = adj.sum(dim=(1, 2))
    inv = 1.0 / torch.where(num_neighbor != 0, num_neighbor, torch.ones_like(num_neighbor))
    return adj * inv[:, None, None, :]


def relgcn_update(h: Tensor, adj: Tensor, Ws: Tensor, bs: Tensor, We: Tensor, be: Tensor) -> Tensor:
    """models/update/relgcn_update.py:24-44; edge type fastest in the 4*out axis."""
    mb, node, ch = h.shape
    out = Ws.shape[0]
    hs = linear(h, Ws, bs)
    m = linear(h, We, be).reshape(mb, node, out, NUM_EDGE_TYPE).permute(0, 3, 1, 2)
    m = torch.matmul(adj, m).sum(dim=1)
    return hs + m


def relgcn_forward(p: Params, atom_array: Tensor, adj: Tensor, n_convs: int,
                   scale_adj: bool = True) -> Tuple[Tensor, Tensor]:
    """models/relgcn.py:61-73.  Returns (g, atoms) with atoms = last-layer h
    (the reference has no get_atom_array(); SURVEY.md 8(a) R6 adds one)."""
    h = p["embed/W"][atom_array.long()]
    if scale_adj:
        adj = rescale_adj(adj)
    for i in range(n_convs):
        pre = f"rgcn_convs/{i}"
        h = torch.tanh(relgcn_update(h, adj, p[f"{pre}/graph_linear_self/W"], p[f"{pre}/graph_linear_self/b"],
                                     p[f"{pre}/graph_linear_edge/W"], p[f"{pre}/graph_linear_edge/b"]))
    g = ggnn_readout_block(p, "rgcn_readout", h, None, True, "tanh", "identity")
    return g, h


# --------------------------------------------------------------------------- #
# fine-grained co-attention family (models/coattention/*.py)
# --------------------------------------------------------------------------- #
def fine_energy(p: Params, prefix: str, atoms_1: Tensor, atoms_2: Tensor, activation: str) -> Tensor:
    """compute_attention(query=atoms_2, key=atoms_1) of nie_coattention.py:372-396
    (identical bodies in vqa/Pooling/lt).  Tiles both inputs to
    (mb*N2*N1, hid), calls Bilinear(e1=key=atoms_1, e2=query=atoms_2) and
    reshapes to C (mb, N2, N1)."""
    mb, n2, hid = atoms_2.shape
    n1 = atoms_1.shape[1]
    query = atoms_2[:, :, None, :].expand(mb, n2, n1, hid).reshape(mb * n2 * n1, hid)
    key = atoms_1[:, None, :, :].expand(mb, n2, n1, hid).reshape(mb * n2 * n1, hid)
    e = bilinear(key, query, p[f"{prefix}energy_layer/W"], p[f"{prefix}energy_layer/V1"],
                 p[f"{prefix}energy_layer/V2"], p[f"{prefix}energy_layer/b"])
    return ACT[activation](e).reshape(mb, n2, n1)


def fourier_energy(p: Params, prefix: str, atoms_1: Tensor, atoms_2: Tensor, activation: str) -> Tensor:
    """FourierFineCoattention.compute_attention(query=atoms_2, key=atoms_1), nie_coattention.py:460-505: functions.fft
    of (x, 0) along the last axis (:507-515), every (query, key) combination tiled out, then
    act(energy_layer(key_real, query_real) + energy_layer(key_imag, query_imag)) (:489) -- the same Bilinear link
    twice, so V1, V2 and b are applied twice."""
    mb, n2, hid = atoms_2.shape
    n1 = atoms_1.shape[1]
    fq, fk = torch.fft.fft(atoms_2, dim=-1), torch.fft.fft(atoms_1, dim=-1)
    til_q = lambda x: x[:, :, None, :].expand(mb, n2, n1, hid).reshape(mb * n2 * n1, hid)
    til_k = lambda x: x[:, None, :, :].expand(mb, n2, n1, hid).reshape(mb * n2 * n1, hid)
    E = lambda k, q: bilinear(k, q, p[f"{prefix}energy_layer/W"], p[f"{prefix}energy_layer/V1"],
                              p[f"{prefix}energy_layer/V2"], p[f"{prefix}energy_layer/b"])
    e = E(til_k(fk.real), til_q(fq.real)) + E(til_k(fk.imag), til_q(fq.imag))
    return ACT[activation](e).reshape(mb, n2, n1)


def nie_coattention(p: Params, atoms_1: Tensor, atoms_2: Tensor, activation: str = "tanh",
                    prefix: str = "", n_lt: int = 0, fourier: bool = False) -> Tuple[Tensor, Tensor]:
    """NieFineCoattention.__call__ nie_coattention.py:335-370 (VQAParallelCoattention
    vqa_parallel_coattention.py:42-77 is the same computation).  g_1/g_2 are
    ignored by the fine family.
    n_lt = 1, 2, 3: Deep / VeryDeep / ExtremeDeep NieFineCoattention (:38-75, :137-178, :241-282): after the energy
    C is taken from the ORIGINAL atoms, each side's atoms pass through its n_lt prev_lt GraphLinear layers (affine, no
    activation) before the head and j projections."""
    P = lambda k: p[prefix + k]
    C = (fourier_energy if fourier else fine_energy)(p, prefix, atoms_1, atoms_2, activation)        # (mb, N2, N1)
    L_2 = torch.softmax(C, dim=1)                                    # :347
    L_1 = torch.softmax(C.transpose(1, 2), dim=1)                    # :349  (mb, N1, N2)
    for k in range(n_lt):                                            # :54-59 / :155-163
        n1 = "prev_lt_layer_1" if n_lt == 1 else f"prev_lt_layers_1/{k}"
        n2 = "prev_lt_layer_2" if n_lt == 1 else f"prev_lt_layers_2/{k}"
        atoms_1 = linear(atoms_1, P(n1 + "/W"), P(n1 + "/b"))
        atoms_2 = linear(atoms_2, P(n2 + "/W"), P(n2 + "/b"))
    lt_1 = linear(atoms_1, P("lt_layer_1/W"))                        # (mb, N1, head)
    lt_2 = linear(atoms_2, P("lt_layer_2/W"))
    H_1 = torch.tanh(lt_1 + torch.bmm(L_1, lt_2))                    # :356-358
    H_2 = torch.tanh(lt_2 + torch.bmm(L_2, lt_1))                    # :361-362
    attn_1 = torch.softmax(linear(H_1, P("attention_layer_1/W")), dim=1)   # :364 default axis=1
    attn_2 = torch.softmax(linear(H_2, P("attention_layer_2/W")), dim=1)
    j1 = linear(atoms_1, P("j_layer/W"), P("j_layer/b"))
    j2 = linear(atoms_2, P("j_layer/W"), P("j_layer/b"))
    return (attn_1 * j1).sum(dim=1), (attn_2 * j2).sum(dim=1)        # :368-369


def pooling_coattention(p: Params, atoms_1: Tensor, atoms_2: Tensor, activation: str = "tanh",
                        prefix: str = "") -> Tuple[Tensor, Tensor]:
    """PoolingFineCoattention.__call__ PoolingFineCoattention.py:31-57."""
    P = lambda k: p[prefix + k]
    E = fine_energy(p, prefix, atoms_1, atoms_2, activation)        # (mb, N2, N1)
    attn_1 = torch.softmax(E.mean(dim=1), dim=1)[:, :, None]
    attn_2 = torch.softmax(E.mean(dim=2), dim=1)[:, :, None]
    j1 = linear(atoms_1, P("j_layer/W"), P("j_layer/b"))
    j2 = linear(atoms_2, P("j_layer/W"), P("j_layer/b"))
    return (attn_1 * j1).sum(dim=1), (attn_2 * j2).sum(dim=1)


def l2_normalize(x: Tensor, dim: int, eps: float = 1e-5) -> Tensor:
    """chainer.functions.normalize(x, eps=1e-5, axis): x / (||x||_2 + eps) along ``axis`` (the eps is added to the norm,
    not clamped; SURVEY.md Appendix B: third-party semantics restated from memory)."""
    return x / (torch.sqrt((x * x).sum(dim=dim, keepdim=True)) + eps)


def bimpm_coattention(p: Params, atoms_1: Tensor, atoms_2: Tensor, prefix: str = "", with_max_pool: bool = True,
                      with_att_mean: bool = True, with_att_max: bool = True) -> Tuple[Tensor, Tensor]:
    """BiMPM.__call__ models/coattention/bimpm.py:45-199 with aggr = F.sum (its only use, train_binary.py:255-256),
    op for op.  Returns (mol_1, mol_2) of shape (mb, n_match * head): the reference never applies an out layer (:35).
    Restated as written, including what looks unintended: mp_matching_func (:50-79) takes column 0 of the
    (head x head) product, so perspective k of v1 is compared with perspective 0 of v2."""
    mb, N_1, d = atoms_1.shape
    N_2 = atoms_2.shape[1]

    def mp_matching_func(v1, v2, w):                                   # :50-79
        head = w.shape[0]
        wt = w.t()[None, None]                                         # (1, 1, d, head)
        v1e = wt * v1[..., None]                                       # (mb, N, d, head)
        if v2.dim() == 3:
            v2e = wt * v2[..., None]
        else:
            v2e = wt * v2[:, None, :, None].expand(-1, v1.shape[1], -1, head)
        v1n, v2n = l2_normalize(v1e, 2), l2_normalize(v2e, 2)          # :73-74 normalised over hidden_dim
        sim = torch.matmul(v1n.transpose(2, 3), v2n)                   # (mb, N, head, head) :76
        return sim[:, :, :, 0]                                         # :78

    def mp_matching_func_pairwise(v1, v2, w):                          # :81-105
        we = w[None, :, None, :]                                       # (1, head, 1, d)
        v1n = l2_normalize(we * v1[:, None], 3)                        # (mb, head, N_1, d)
        v2n = l2_normalize(we * v2[:, None], 3)
        sim = torch.matmul(v1n, v2n.transpose(2, 3))                   # (mb, head, N_1, N_2)
        return sim.permute(0, 2, 3, 1)                                 # (mb, N_1, N_2, head)

    def div_with_small_value(n, dd, eps=1e-4):                         # :122-124
        return n / torch.maximum(dd, torch.full_like(dd, eps))

    mv1, mv2 = [], []
    if with_max_pool:                                                  # :132-142
        mv_max = mp_matching_func_pairwise(atoms_1, atoms_2, p[f"{prefix}max_pooling_W"])
        mv1.append(mv_max.max(dim=2).values)
        mv2.append(mv_max.max(dim=1).values)
    if with_att_mean or with_att_max:
        att = torch.matmul(l2_normalize(atoms_1, 2), l2_normalize(atoms_2, 2).transpose(1, 2))     # :107-120 (mb, N_1, N_2)
        att_atoms2 = atoms_2[:, None, :, :] * att[..., None]           # (mb, N_1, N_2, d) :150
        att_atoms1 = atoms_1[:, :, None, :] * att[..., None]           # :154
        if with_att_mean:                                              # :156-167
            mean2 = div_with_small_value(att_atoms2.sum(dim=2), att.sum(dim=2, keepdim=True))
            mean1 = div_with_small_value(att_atoms1.sum(dim=1), att.sum(dim=1, keepdim=True).transpose(1, 2))
            mv1.append(mp_matching_func(atoms_1, mean2, p[f"{prefix}att_mean_W"]))
            mv2.append(mp_matching_func(atoms_2, mean1, p[f"{prefix}att_mean_W"]))
        if with_att_max:                                               # :169-182
            max2 = att_atoms2.max(dim=2).values
            max1 = att_atoms1.max(dim=1).values
            mv1.append(mp_matching_func(atoms_1, max2, p[f"{prefix}att_max_W"]))
            mv2.append(mp_matching_func(atoms_2, max1, p[f"{prefix}att_max_W"]))
    mv1, mv2 = torch.cat(mv1, dim=2), torch.cat(mv2, dim=2)            # :185-187
    return mv1.sum(dim=1), mv2.sum(dim=1)                              # :190-192 aggr = F.sum over atoms


def init_bimpm(dr: "_Draw", prefix: str, hidden_dim: int, head: int) -> None:
    """bimpm.py:24-33: three (head, hidden_dim) parameters, HeNormal (std = sqrt(2 / fan_in), fan_in = hidden_dim)."""
    for name in ("max_pooling_W", "att_mean_W", "att_max_W"):
        dr.normal(f"{prefix}{name}", (head, hidden_dim), math.sqrt(2.0 / hidden_dim))


# --------------------------------------------------------------------------- #
# coarse co-attention family (atom x molecule-vector)
# --------------------------------------------------------------------------- #
def parallel_coattention(p: Params, atoms_1: Tensor, g_1: Tensor, atoms_2: Tensor, g_2: Tensor,
                         activation: str = "tanh", weight_tying: bool = True,
                         prefix: str = "") -> Tuple[Tensor, Tensor]:
    """ParallelCoattention parallel_coattention.py:34-84 (head must be 1 for the
    tile at :45 to match j_layer's out_dim).  No softmax."""
    P = lambda k: p[prefix + k]

    def attend(query, key, focus):
        li = 0 if weight_tying else focus - 1
        mb, n, hid = key.shape
        q = query[:, None, :].expand(mb, n, query.shape[1]).reshape(mb * n, -1)
        e = bilinear(key.reshape(mb * n, hid), q, P(f"energy_layers/{li}/W"), P(f"energy_layers/{li}/V1"),
                     P(f"energy_layers/{li}/V2"), P(f"energy_layers/{li}/b"))
        return ACT[activation](e).reshape(mb, n, -1)

    a1 = attend(g_2, atoms_1, 1)
    a2 = attend(g_1, atoms_2, 2)
    j1 = linear(atoms_1, P("j_layer/W"), P("j_layer/b"))
    j2 = linear(atoms_2, P("j_layer/W"), P("j_layer/b"))
    return (a1 * j1).sum(dim=1), (a2 * j2).sum(dim=1)


def circular_parallel_coattention(p: Params, atoms_1: Tensor, g_1: Tensor, atoms_2: Tensor, g_2: Tensor,
                                  activation: str = "tanh", prefix: str = "") -> Tuple[Tensor, Tensor]:
    """CircularParallelCoattention.__call__ parallel_coattention.py:108-160: atoms -> j_layer (:115, :129); the other
    molecule's vector tiled over the atoms; gate = act(circular_correlation(key=j atoms, query)) (:156, the same
    fft / conj-product / ifft op sequence as HolE's, :162-187); compact = sum over atoms of gate * j atoms."""
    P = lambda k: p[prefix + k]

    def attend(query, key):
        mb, n, o = key.shape
        q = query[:, None, :].expand(mb, n, o).reshape(mb * n, o)
        return ACT[activation](circular_correlation(key.reshape(mb * n, o), q)).reshape(mb, n, o)

    j1 = linear(atoms_1, P("j_layer/W"), P("j_layer/b"))
    j2 = linear(atoms_2, P("j_layer/W"), P("j_layer/b"))
    return (attend(g_2, j1) * j1).sum(dim=1), (attend(g_1, j2) * j2).sum(dim=1)


def alternating_coattention(p: Params, atoms_1: Tensor, g_1: Tensor, atoms_2: Tensor, g_2: Tensor,
                            prefix: str = "") -> Tuple[Tensor, Tensor]:
    """AlternatingCoattention alternating_coattention.py:36-86 with
    weight_tying=True (with False the reference indexes energy_layers_2[1] of a
    1-element ChainList, :69, and raises)."""
    P = lambda k: p[prefix + k]

    def attend(query, key):
        mb, n, _ = key.shape
        q = query[:, None, :].expand(mb, n, query.shape[1])
        e = torch.tanh(linear(torch.cat((q, key), dim=2), P("energy_layers_1/0/W"), P("energy_layers_1/0/b")))
        e = linear(e, P("energy_layers_2/0/W"), P("energy_layers_2/0/b"))
        return torch.softmax(e, dim=1)

    j1 = linear(atoms_1, P("j_layer/W"), P("j_layer/b"))
    j2 = linear(atoms_2, P("j_layer/W"), P("j_layer/b"))
    c1 = (attend(g_2, atoms_1) * j1).sum(dim=1)
    c2 = (attend(c1, atoms_2) * j2).sum(dim=1)          # :56 query=compact_1
    return c1, c2


def global_coattention(p: Params, atoms_1: Tensor, atoms_2: Tensor, weight_tying: bool = True,
                       prefix: str = "") -> Tuple[Tensor, Tensor]:
    """GlobalCoattention global_coattention.py:27-73: sigmoid(Linear([atom, mean(other)]))."""
    P = lambda k: p[prefix + k]
    g1 = atoms_1.mean(dim=1)
    g2 = atoms_2.mean(dim=1)

    def attend(query, key, focus):
        li = 0 if weight_tying else focus - 1
        mb, n, _ = key.shape
        q = query[:, None, :].expand(mb, n, query.shape[1])
        return torch.sigmoid(linear(torch.cat((key, q), dim=2), P(f"att_layers/{li}/W"), P(f"att_layers/{li}/b")))

    c1 = (attend(g2, atoms_1, 1) * linear(atoms_1, P("lt_layer/W"), P("lt_layer/b"))).sum(dim=1)
    c2 = (attend(g1, atoms_2, 2) * linear(atoms_2, P("lt_layer/W"), P("lt_layer/b"))).sum(dim=1)
    return c1, c2


def neural_coattention(p: Params, atoms_1: Tensor, atoms_2: Tensor, activation: str = "tanh",
                       weight_tying: bool = True, prefix: str = "") -> Tuple[Tensor, Tensor]:
    """NeuralCoattention neural_coattention.py:27-71."""
    P = lambda k: p[prefix + k]

    def attend(query, key, focus):
        li = 0 if weight_tying else focus - 1
        W, b = P(f"att_layers/{li}/W"), P(f"att_layers/{li}/b")
        context = ACT[activation](linear(query[:, None, :], W, b))       # (mb,1,o)
        doc = ACT[activation](linear(key, W, b))                          # (mb,N,o)
        return torch.sigmoid(torch.bmm(doc, context.transpose(1, 2))), doc

    a1, d1 = attend(atoms_2.mean(dim=1), atoms_1, 1)
    a2, d2 = attend(atoms_1.mean(dim=1), atoms_2, 2)
    return (a1 * d1).sum(dim=1), (a2 * d2).sum(dim=1)


# --------------------------------------------------------------------------- #
# link predictor, pair glue, loss
# --------------------------------------------------------------------------- #
def mlp_forward(p: Params, x: Tensor, n_hidden: int, prefix: str = "mlp/") -> Tensor:
    """MLP.__call__ models/mlp.py:40-45 (relu between layers, linear l_out)."""
    h = x
    for i in range(n_hidden):
        h = torch.relu(linear(h, p[f"{prefix}layers/{i}/W"], p[f"{prefix}layers/{i}/b"]))
    return linear(h, p[f"{prefix}l_out/W"], p[f"{prefix}l_out/b"])


def _mlp_tail(p: Params, h: Tensor, n_hidden: int, prefix: str, layers: str) -> Tensor:
    for i in range(n_hidden):
        h = torch.relu(linear(h, p[f"{prefix}{layers}/{i}/W"], p[f"{prefix}{layers}/{i}/b"]))
    return linear(h, p[f"{prefix}l_out/W"], p[f"{prefix}l_out/b"])


def ntn_forward(p: Params, left_x: Tensor, right_x: Tensor, n_hidden: int, prefix: str = "mlp/") -> Tensor:
    """NTN.__call__ models/mlp.py:65-72: links.Bilinear(left, right, ntn_out_dim) (:52), relu-MLP, l_out."""
    h = bilinear(left_x, right_x, p[prefix + "ntn_layer/W"], p[prefix + "ntn_layer/V1"], p[prefix + "ntn_layer/V2"],
                 p[prefix + "ntn_layer/b"])
    return _mlp_tail(p, h, n_hidden, prefix, "mlp_layers")


def distmult_forward(p: Params, left_x: Tensor, right_x: Tensor, n_hidden: int, prefix: str = "mlp/") -> Tensor:
    """DistMult.__call__ models/mlp.py:87-93 over BilinearDiag (:153-193): bilinear(e1, e2, W_mat) with
    W_mat[:, :, o] = diag(W[o]) and no V1/V2/b terms (:181-182), i.e. y[o] = sum_p W[o,p] e1[p] e2[p]."""
    W = p[prefix + "dm_layer/W"]                                   # (out, left)
    W_mat = torch.stack([torch.diag(v) for v in W]).permute(1, 2, 0)   # (left, right, out)  :186-192
    h = torch.einsum("ni,ijk,nj->nk", left_x, W_mat, right_x)
    return _mlp_tail(p, h, n_hidden, prefix, "mlp_layers")


def symmlp_forward(p: Params, left_x: Tensor, right_x: Tensor, n_hidden: int, prefix: str = "mlp/") -> Tensor:
    """SymMLP.__call__ models/mlp.py:104-110."""
    h = torch.cat((left_x + right_x, left_x * right_x), dim=1)
    return _mlp_tail(p, h, n_hidden, prefix, "layers")


def circular_correlation(left_x: Tensor, right_x: Tensor) -> Tensor:
    """HolE.circular_correlation models/mlp.py:126-151, op for op: fft of both (zero imaginary parts),
    conj(fft(a)) * fft(b) as real/imaginary pairs, real part of the ifft."""
    fa = torch.fft.fft(torch.complex(left_x, torch.zeros_like(left_x)), dim=-1)
    fb = torch.fft.fft(torch.complex(right_x, torch.zeros_like(right_x)), dim=-1)
    prod_real = fa.real * fb.real + fa.imag * fb.imag           # :144
    prod_imag = fa.real * fb.imag - fa.imag * fb.real           # :145
    return torch.fft.ifft(torch.complex(prod_real, prod_imag), dim=-1).real


def hole_forward(p: Params, left_x: Tensor, right_x: Tensor, n_hidden: int, prefix: str = "mlp/") -> Tensor:
    """HolE.__call__ models/mlp.py:119-124."""
    return _mlp_tail(p, circular_correlation(left_x, right_x), n_hidden, prefix, "layers")


def sigmoid_cross_entropy(y: Tensor, t: Tensor) -> Tensor:
    """chainer.functions.sigmoid_cross_entropy(normalize=True): mean over
    elements with t != -1 of softplus(y) - t*y (train_ddi_modify.py:285)."""
    t = t.to(y.dtype)
    mask = (t != -1)
    loss = torch.nn.functional.softplus(y) - t * y
    loss = torch.where(mask, loss, torch.zeros_like(loss))
    return loss.sum() / mask.sum().clamp(min=1).to(y.dtype)


def pair_forward(p: Params, atoms_1: Tensor, adjs_1: Tensor, atoms_2: Tensor, adjs_2: Tensor, *,
                 encoder: str = "ggnn", n_layers: int = 4, weight_tying: bool = True,
                 attn: Optional[str] = "nie", attn_activation: str = "tanh",
                 mlp_hidden: int = 2, scale_adj: bool = True, sim_method: str = "mlp") -> Tuple[Tensor, Tensor, Tensor]:
    """GraphConvPredictorForPair.__call__: with co-attention train_binary.py:84-118
    (= eval_coattention.py:66-100); without train_ddi_modify.py:66-77.  ``sim_method``: the link predictor selected by
    set_up_predictor (train_binary.py:165-187) and dispatched on by class name in __call__ (:98-116).
    Returns (logits, g1, g2) with g1/g2 the vectors fed to the link predictor."""
    def enc(a, adj):
        if encoder == "ggnn":
            return ggnn_forward(p, a, adj, n_layers, weight_tying, prefix="graph_conv/")
        if encoder == "relgcn":
            sub = {k[len("graph_conv/"):]: v for k, v in p.items() if k.startswith("graph_conv/")}
            return relgcn_forward(sub, a, adj, n_layers, scale_adj)
        raise ValueError(encoder)

    g1, at1 = enc(atoms_1, adjs_1)
    g2, at2 = enc(atoms_2, adjs_2)
    if attn in ("nie", "vqa"):
        g1, g2 = nie_coattention(p, at1, at2, attn_activation, prefix="attn/")
    elif attn == "pool":
        g1, g2 = pooling_coattention(p, at1, at2, attn_activation, prefix="attn/")
    elif attn == "parallel":
        g1, g2 = parallel_coattention(p, at1, g1, at2, g2, attn_activation, prefix="attn/")
    elif attn == "circ":
        g1, g2 = circular_parallel_coattention(p, at1, g1, at2, g2, attn_activation, prefix="attn/")
    elif attn in ("deep", "very-deep", "extreme-deep"):
        g1, g2 = nie_coattention(p, at1, at2, attn_activation, prefix="attn/",
                                 n_lt={"deep": 1, "very-deep": 2, "extreme-deep": 3}[attn])
    elif attn == "fourier":
        g1, g2 = nie_coattention(p, at1, at2, attn_activation, prefix="attn/", fourier=True)
    elif attn == "alternating":
        g1, g2 = alternating_coattention(p, at1, g1, at2, g2, prefix="attn/")
    elif attn == "global":
        g1, g2 = global_coattention(p, at1, at2, prefix="attn/")
    elif attn == "neural":
        g1, g2 = neural_coattention(p, at1, at2, attn_activation, prefix="attn/")
    elif attn == "bimpm":
        g1, g2 = bimpm_coattention(p, at1, at2, prefix="attn/")
    elif attn is not None:
        raise ValueError(attn)
    if sim_method == "mlp":                                          # train_binary.py:98-101
        y = mlp_forward(p, torch.cat((g1, g2), dim=-1), mlp_hidden)
    else:                                                            # :102-113: self.mlp(g1, g2)
        fwd = {"ntn": ntn_forward, "hole": hole_forward, "symmlp": symmlp_forward, "dist-mult": distmult_forward}[sim_method]
        y = fwd(p, g1, g2, mlp_hidden, prefix="mlp/")
    return y, g1, g2


# --------------------------------------------------------------------------- #
# parameter construction (Chainer default initialisers)
# --------------------------------------------------------------------------- #
class _Draw:
    """Fixed-order parameter draws from numpy RandomState(seed)."""

    def __init__(self, seed: int, dtype: torch.dtype, bias_scale: float):
        self.rs = np.random.RandomState(seed)
        self.dtype = dtype
        self.bias_scale = bias_scale
        self.p: Params = {}

    def normal(self, name: str, shape: Sequence[int], std: float) -> None:
        self.p[name] = torch.from_numpy(self.rs.normal(0.0, std, size=tuple(shape))).to(self.dtype)

    def lin(self, name: str, n_in: int, n_out: int, bias: bool = True) -> None:
        # chainer Linear: W ~ LeCunNormal (std 1/sqrt(in)); b = 0 by default.  Tests use
        # bias_scale > 0 so that bias handling is actually exercised.
        self.normal(f"{name}/W", (n_out, n_in), 1.0 / math.sqrt(n_in))
        if bias:
            self.normal(f"{name}/b", (n_out,), self.bias_scale if self.bias_scale > 0 else 1.0)
            if self.bias_scale == 0:
                self.p[f"{name}/b"].zero_()

    def bil(self, name: str, l: int, r: int, o: int) -> None:
        self.normal(f"{name}/W", (l, r, o), 1.0 / math.sqrt(l))
        self.normal(f"{name}/V1", (l, o), 1.0 / math.sqrt(l))
        self.normal(f"{name}/V2", (r, o), 1.0 / math.sqrt(r))
        self.normal(f"{name}/b", (o,), self.bias_scale if self.bias_scale > 0 else 1.0)
        if self.bias_scale == 0:
            self.p[f"{name}/b"].zero_()


def init_ggnn(dr: _Draw, prefix: str, out_dim: int, hidden_dim: int, n_layers: int,
              weight_tying: bool = True, concat_hidden: bool = False, n_atom_types: int = MAX_ATOMIC_NUM) -> None:
    """Link tree of models/ggnn.py:83-141."""
    d = hidden_dim
    dr.normal(f"{prefix}embed/W", (n_atom_types, d), 1.0)
    for i in range(1 if weight_tying else n_layers):
        dr.lin(f"{prefix}message_layers/{i}", d, NUM_EDGE_TYPE * d)
    for n in ("W_r", "W_z", "W"):
        dr.lin(f"{prefix}update_layer/{n}", 2 * d, d)
    for n in ("U_r", "U_z", "U"):
        dr.lin(f"{prefix}update_layer/{n}", d, d)
    for i in range(n_layers if concat_hidden else 1):
        dr.lin(f"{prefix}i_layers/{i}", 2 * d, out_dim)
        dr.lin(f"{prefix}j_layers/{i}", d, out_dim)


def init_ggnn_modular(dr: _Draw, out_dim: int, hidden_dim: int, n_layers: int, weight_tying: bool = True,
                      concat_hidden: bool = False) -> None:
    """Link tree of models/models/ggnn.py:54-64."""
    d = hidden_dim
    dr.normal("embed/W", (MAX_ATOMIC_NUM, d), 1.0)
    for i in range(1 if weight_tying else n_layers):
        dr.lin(f"update_layers/{i}/graph_linear", d, NUM_EDGE_TYPE * d)
        for n in ("W_r", "W_z", "W"):
            dr.lin(f"update_layers/{i}/update_layer/{n}", 2 * d, d)
        for n in ("U_r", "U_z", "U"):
            dr.lin(f"update_layers/{i}/update_layer/{n}", d, d)
    for i in range(n_layers if concat_hidden else 1):
        dr.lin(f"readout_layers/{i}/i_layer", 2 * d, out_dim)
        dr.lin(f"readout_layers/{i}/j_layer", 2 * d, out_dim)


def init_relgcn(dr: _Draw, prefix: str, out_channels: int, ch_list: Sequence[int]) -> None:
    """Link tree of models/relgcn.py:38-55."""
    dr.normal(f"{prefix}embed/W", (MAX_ATOMIC_NUM, ch_list[0]), 1.0)
    for i in range(len(ch_list) - 1):
        dr.lin(f"{prefix}rgcn_convs/{i}/graph_linear_self", ch_list[i], ch_list[i + 1])
        dr.lin(f"{prefix}rgcn_convs/{i}/graph_linear_edge", ch_list[i], NUM_EDGE_TYPE * ch_list[i + 1])
    dr.lin(f"{prefix}rgcn_readout/i_layer", ch_list[-1], out_channels, bias=False)
    dr.lin(f"{prefix}rgcn_readout/j_layer", ch_list[-1], out_channels, bias=False)


def init_nie(dr: _Draw, prefix: str, hidden_dim: int, out_dim: int, head: int, n_lt: int = 0) -> None:
    """Link tree of nie_coattention.py:323-330 (VQA identical); n_lt > 0: the Deep variants' trees (:24-33, :118-131)."""
    dr.bil(f"{prefix}energy_layer", hidden_dim, hidden_dim, 1)
    dr.lin(f"{prefix}attention_layer_1", head, 1, bias=False)
    dr.lin(f"{prefix}attention_layer_2", head, 1, bias=False)
    for side in (1, 2):
        for k in range(n_lt):
            dr.lin(f"{prefix}prev_lt_layer_{side}" if n_lt == 1 else f"{prefix}prev_lt_layers_{side}/{k}", hidden_dim, hidden_dim)
    dr.lin(f"{prefix}lt_layer_1", hidden_dim, head, bias=False)
    dr.lin(f"{prefix}lt_layer_2", hidden_dim, head, bias=False)
    dr.lin(f"{prefix}j_layer", hidden_dim, out_dim)


def init_pooling(dr: _Draw, prefix: str, hidden_dim: int, out_dim: int) -> None:
    dr.bil(f"{prefix}energy_layer", hidden_dim, hidden_dim, 1)
    dr.lin(f"{prefix}j_layer", hidden_dim, out_dim)


def init_parallel(dr: _Draw, prefix: str, hidden_dim: int, out_dim: int, head: int = 1,
                  weight_tying: bool = True) -> None:
    for i in range(1 if weight_tying else 2):
        dr.bil(f"{prefix}energy_layers/{i}", hidden_dim, out_dim, head)
    dr.lin(f"{prefix}j_layer", hidden_dim, out_dim)


def init_alternating(dr: _Draw, prefix: str, hidden_dim: int, out_dim: int, head: int) -> None:
    dr.lin(f"{prefix}energy_layers_1/0", hidden_dim + out_dim, head)
    dr.lin(f"{prefix}energy_layers_2/0", head, 1)
    dr.lin(f"{prefix}j_layer", hidden_dim, out_dim)


def init_global(dr: _Draw, prefix: str, hidden_dim: int, out_dim: int, weight_tying: bool = True) -> None:
    for i in range(1 if weight_tying else 2):
        dr.lin(f"{prefix}att_layers/{i}", 2 * hidden_dim, out_dim)
    dr.lin(f"{prefix}lt_layer", hidden_dim, out_dim)


def init_neural(dr: _Draw, prefix: str, hidden_dim: int, out_dim: int, weight_tying: bool = True) -> None:
    for i in range(1 if weight_tying else 2):
        dr.lin(f"{prefix}att_layers/{i}", hidden_dim, out_dim)


def init_mlp(dr: _Draw, prefix: str, in_dim: int, out_dim: int, hidden_dims: Sequence[int] = (32, 16)) -> None:
    """models/mlp.py:32-38."""
    n = in_dim
    for i, hd in enumerate(hidden_dims):
        dr.lin(f"{prefix}layers/{i}", n, hd)
        n = hd
    dr.lin(f"{prefix}l_out", n, out_dim)


def init_link(dr: _Draw, prefix: str, kind: str, fp_dim: int, out_dim: int, hidden_dims: Sequence[int] = (32, 16),
              feat_dim: int = 8) -> None:
    """Link predictors of models/mlp.py:48-124 (kind: ntn / distmult / symmlp / hole); ``feat_dim`` =
    ntn_out_dim / dm_out_dim (train_binary.py:173,184)."""
    if kind == "ntn":
        dr.bil(prefix + "ntn_layer", fp_dim, fp_dim, feat_dim)
        n, layers = feat_dim, "mlp_layers"
    elif kind == "distmult":
        dr.normal(prefix + "dm_layer/W", (feat_dim, fp_dim), 1.0 / math.sqrt(fp_dim))
        n, layers = feat_dim, "mlp_layers"
    elif kind == "symmlp":
        n, layers = 2 * fp_dim, "layers"
    elif kind == "hole":
        n, layers = fp_dim, "layers"
    else:
        raise ValueError(kind)
    for i, hd in enumerate(hidden_dims):
        dr.lin(f"{prefix}{layers}/{i}", n, hd)
        n = hd
    dr.lin(f"{prefix}l_out", n, out_dim)


def make_pair_params(seed: int = 777, *, encoder: str = "ggnn", hidden_dim: int = 16, out_dim: int = 16,
                     n_layers: int = 2, weight_tying: bool = True, attn: Optional[str] = "nie", head: int = 8,
                     class_num: int = 1, mlp_hidden: Sequence[int] = (32, 16),
                     dtype: torch.dtype = torch.float64, bias_scale: float = 0.1, sim_method: str = "mlp") -> Params:
    """Whole GraphConvPredictorForPair parameter set in a fixed draw order
    (graph_conv, attn, mlp) from RandomState(seed) (seed default train_ddi_modify.py:227)."""
    dr = _Draw(seed, dtype, bias_scale)
    if encoder == "ggnn":
        init_ggnn(dr, "graph_conv/", out_dim, hidden_dim, n_layers, weight_tying)
    elif encoder == "relgcn":
        init_relgcn(dr, "graph_conv/", out_dim, [hidden_dim] * (n_layers + 1))
    else:
        raise ValueError(encoder)
    if attn in ("nie", "vqa"):
        init_nie(dr, "attn/", hidden_dim, out_dim, head)
    elif attn == "pool":
        init_pooling(dr, "attn/", hidden_dim, out_dim)
    elif attn == "parallel":
        init_parallel(dr, "attn/", hidden_dim, out_dim, 1)
    elif attn == "circ":
        dr.lin("attn/j_layer", hidden_dim, out_dim)
    elif attn in ("deep", "very-deep", "extreme-deep"):
        init_nie(dr, "attn/", hidden_dim, out_dim, head, n_lt={"deep": 1, "very-deep": 2, "extreme-deep": 3}[attn])
    elif attn == "fourier":
        init_nie(dr, "attn/", hidden_dim, out_dim, head)
    elif attn == "alternating":
        init_alternating(dr, "attn/", hidden_dim, out_dim, head)
    elif attn == "global":
        init_global(dr, "attn/", hidden_dim, out_dim)
    elif attn == "neural":
        init_neural(dr, "attn/", hidden_dim, out_dim)
    elif attn == "bimpm":
        init_bimpm(dr, "attn/", hidden_dim, head)                     # train_binary.py:255: head = fp_out_dim there
        init_mlp(dr, "mlp/", 2 * 3 * head, class_num, mlp_hidden)     # three matchings of `head` perspectives per side
        return dr.p
    if sim_method == "mlp":
        init_mlp(dr, "mlp/", 2 * out_dim, class_num, mlp_hidden)
    else:                                                            # train_binary.py:170-186
        init_link(dr, "mlp/", {"dist-mult": "distmult"}.get(sim_method, sim_method), out_dim, class_num, mlp_hidden)
    return dr.p


def chainer_adam_step(params: List[Tensor], grads: List[Tensor], state: List[Dict[str, Tensor]], t: int,
                      alpha: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                      weight_decay_rate: float = 0.0) -> None:
    """chainer.optimizers.Adam update (train_ddi_modify.py:289): alpha_t =
    alpha*sqrt(1-b2^t)/(1-b1^t); p -= alpha_t*m/(sqrt(v)+eps) + wd*p."""
    a_t = alpha * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for p_, g, st in zip(params, grads, state):
        st["m"].add_((g - st["m"]) * (1 - beta1))
        st["v"].add_((g * g - st["v"]) * (1 - beta2))
        p_.sub_(a_t * st["m"] / (st["v"].sqrt() + eps) + weight_decay_rate * p_)

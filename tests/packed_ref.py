"""Packed-form CPU restatement (test infrastructure, not product).

Same math as ``oracle/ref_cpu.py`` but on the packed row layout of
``bmp.packed`` (virtual pad rows with multiplicities, CSR bonds).  It proves on
the CPU -- in float64, against the dense oracle -- that the packed formulation
reproduces the reference's unmasked-padding semantics, and it gives the GPU
tests per-op intermediates on exactly the tensors the HIP kernels see.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor


def _edges(pb, transposed=False):
    ptr = (pb.csrT_ptr if transposed else pb.csr_ptr).cpu().long()
    col = (pb.csrT_col if transposed else pb.csr_col).cpu().long()
    val = (pb.csrT_val if transposed else pb.csr_val).cpu()
    N = pb.n_rows
    dst = torch.repeat_interleave(torch.arange(N), ptr[1:] - ptr[:-1])
    return dst, col >> 2, col & 3, val


def gather_agg(pb, h: Tensor) -> Tuple[Tensor, Tensor]:
    """AGG[i, e*d + k] = sum_{(j,e) in N(i)} val * h[j, k];  WDEG[i, e] = sum val."""
    N, d = h.shape
    dst, src, typ, val = _edges(pb)
    val = val.to(h.dtype)
    agg = torch.zeros(N * 4, d, dtype=h.dtype)
    agg = agg.index_add(0, dst * 4 + typ, h[src] * val[:, None]).reshape(N, 4 * d)
    wdeg = torch.zeros(N * 4, dtype=h.dtype).index_add(0, dst * 4 + typ, val).reshape(N, 4)
    return agg, wdeg


def msg_weights(W: Tensor, b: Tensor) -> Tuple[Tensor, Tensor]:
    """Reference layout W (4*o, d) with feature k = 4*c + e (models/ggnn.py:223-224) ->
    WT (4*d, o) with row e*d + k, col c; bE (4, o)."""
    o4, d = W.shape
    o = o4 // 4
    WT = W.reshape(o, 4, d).permute(1, 2, 0).reshape(4 * d, o)
    bE = b.reshape(o, 4).t()
    return WT, bE


def message(pb, h: Tensor, W: Tensor, b: Tensor) -> Tensor:
    agg, wdeg = gather_agg(pb, h)
    WT, bE = msg_weights(W, b)
    return agg @ WT + wdeg @ bE


def gru(p: Dict[str, Tensor], pre: str, h: Tensor, m: Tensor, first: bool):
    g = lambda n: (p[f"{pre}/{n}/W"], p[f"{pre}/{n}/b"])
    x = torch.cat((h, m), dim=1)
    Wz, bz = g("W_z"); W, b = g("W")
    if first:
        z = torch.sigmoid(x @ Wz.t() + bz)
        c = torch.tanh(x @ W.t() + b)
        return z * c, dict(z=z, c=c)
    Wr, br = g("W_r"); Ur, bur = g("U_r"); Uz, buz = g("U_z"); U, bu = g("U")
    r = torch.sigmoid(x @ Wr.t() + br + h @ Ur.t() + bur)
    z = torch.sigmoid(x @ Wz.t() + bz + h @ Uz.t() + buz)
    c = torch.tanh(x @ W.t() + b + (r * h) @ U.t() + bu)
    return z * c + (1 - z) * h, dict(r=r, z=z, c=c)


def segment_sum(pb, rows: Tensor) -> Tensor:
    """g[mol] = sum over the molecule's rows (rows already weighted)."""
    r0 = pb.mol_row0.cpu().long()
    nr = pb.mol_nrows.cpu().long()
    mol = torch.repeat_interleave(torch.arange(pb.n_mols), nr)
    idx = torch.repeat_interleave(r0, nr) + (torch.arange(int(nr.sum())) -
                                             torch.repeat_interleave(torch.cumsum(nr, 0) - nr, nr))
    out = torch.zeros(pb.n_mols, rows.shape[1], dtype=rows.dtype)
    return out.index_add(0, mol, rows[idx])


def ggnn_forward(p: Dict[str, Tensor], pb, n_layers: int, weight_tying=True, prefix="") -> Tuple[Tensor, Tensor]:
    P = lambda k: p[prefix + k]
    sp = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix + "update_layer/")}
    h = P("embed/W")[pb.atom_id.cpu().long()]
    h0 = h
    for step in range(n_layers):
        li = 0 if weight_tying else step
        m = message(pb, h, P(f"message_layers/{li}/W"), P(f"message_layers/{li}/b"))
        h, _ = gru(sp, "update_layer", h, m, first=(step == 0))
    w = pb.row_w.cpu().to(h.dtype)[:, None]
    gi = torch.sigmoid(torch.cat((h, h0), 1) @ P("i_layers/0/W").t() + P("i_layers/0/b"))
    gj = h @ P("j_layers/0/W").t() + P("j_layers/0/b")
    return segment_sum(pb, w * gi * gj), h


def nie_coattention(p: Dict[str, Tensor], pb, X: Tensor, pair_m1, pair_m2, activation="tanh", prefix="", n_lt=0,
                    fourier=False):
    """Packed NieFineCoattention with multiplicities (nie_coattention.py:335-396).  n_lt > 0: the Deep variants
    (:13-309) -- head and j projections act on the prev_lt chain of the atoms, layer by layer as the reference does."""
    P = lambda k: p[prefix + k]

    def chain(x, side):
        for k in range(n_lt):
            name = f"prev_lt_layer_{side}" if f"{prefix}prev_lt_layer_{side}/W" in p else f"prev_lt_layers_{side}/{k}"
            x = x @ P(name + "/W").t() + P(name + "/b")
        return x

    act = {"tanh": torch.tanh, "identity": lambda x: x}[activation]
    W = P("energy_layer/W")[:, :, 0]; V1 = P("energy_layer/V1")[:, 0]; V2 = P("energy_layer/V2")[:, 0]
    cb = P("energy_layer/b")[0]
    r0 = pb.mol_row0.cpu().long(); nr = pb.mol_nrows.cpu().long(); w = pb.row_w.cpu().to(X.dtype)
    out1, out2 = [], []
    for b in range(len(pair_m1)):
        a, c = int(pair_m1[b]), int(pair_m2[b])
        x1 = X[r0[a]:r0[a] + nr[a]]; w1 = w[r0[a]:r0[a] + nr[a]]
        x2 = X[r0[c]:r0[c] + nr[c]]; w2 = w[r0[c]:r0[c] + nr[c]]
        pre = lambda u1, u2: (u2 @ W.t()) @ u1.t() + (u1 @ V1)[None, :] + (u2 @ V2)[:, None] + cb
        if fourier:          # FourierFineCoattention (:460-505): energy between the DFTs over the feature axis
            f1, f2 = torch.fft.fft(x1, dim=-1), torch.fft.fft(x2, dim=-1)
            C = act(pre(f1.real, f2.real) + pre(f1.imag, f2.imag))
        else:
            C = act(pre(x1, x2))                                    # (n2, n1)
        E = torch.exp(C - C.max())
        L2 = E / (w2[:, None] * E).sum(dim=0, keepdim=True)        # softmax over i (side-2 atoms) per j
        L1 = E / (w1[None, :] * E).sum(dim=1, keepdim=True)        # softmax over j per i; L1[j,i] = this[i,j]
        y1, y2 = chain(x1, 1), chain(x2, 2)
        P1 = y1 @ P("lt_layer_1/W").t(); P2 = y2 @ P("lt_layer_2/W").t()
        H1 = torch.tanh(P1 + (L1 * w2[:, None]).t() @ P2)
        H2 = torch.tanh(P2 + (L2 * w1[None, :]) @ P1)
        s1 = (H1 @ P("attention_layer_1/W").t())[:, 0]; s2 = (H2 @ P("attention_layer_2/W").t())[:, 0]
        e1 = torch.exp(s1 - s1.max()); e2 = torch.exp(s2 - s2.max())
        a1 = e1 / (w1 * e1).sum(); a2 = e2 / (w2 * e2).sum()
        J1 = y1 @ P("j_layer/W").t() + P("j_layer/b"); J2 = y2 @ P("j_layer/W").t() + P("j_layer/b")
        out1.append(((w1 * a1)[:, None] * J1).sum(0)); out2.append(((w2 * a2)[:, None] * J2).sum(0))
    return torch.stack(out1), torch.stack(out2)

"""De-duplicated encoding of a drug-pair batch (SURVEY.md 8(d) "de-duplication caveat", 7 "the big non-kernel win").

A batch of B pairs holds 2B molecule instances but at most 544 distinct drugs (setting.py:30-31); an atom's state after the
propagation steps depends on its molecule alone -- the padded positions of a batch never exchange messages with a real atom
(models/ggnn.py:215-263,584-654).  So the encoder runs once per DISTINCT molecule of the batch, and only the co-attention,
whose softmaxes see a side's zero-padding through the pad row's multiplicity (nie_coattention.py:347-349), works on the
per-instance layout: ``bmp_molrows_expand`` copies every instance's rows from its molecule's rows, ``bmp_molrows_reduce`` sums
the atom-state gradients of a molecule's instances in a fixed order before the encoder's backward (csrc/bmp_dedup.hip).

Mathematically identical to the per-instance path (same sums, another association: 1e-6 relative in float32), and reported by
bench.py BESIDE the per-instance figure, never instead of it.  Supported with the fine co-attention family,
which does not read the encoder's molecule vectors (nie_coattention.py:335-370): the readout then runs over the distinct
molecules only and its output is not used.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr, stream
from .packed import DeviceMolStore, PackedMolBatch, pack_from_store_device


@dataclass
class DedupPairBatch:
    pb_u: PackedMolBatch          # one-sided batch of the distinct molecules (what the encoder sees)
    pb: PackedMolBatch            # the per-instance two-sided batch (row multiplicities, pair metadata: the co-attention's)
    uid: torch.Tensor             # [2B] int32   distinct molecule of every instance
    uptr: torch.Tensor            # [U + 1] int32
    uinst: torch.Tensor           # [2B] int32   instances grouped by distinct molecule, ascending
    n_distinct: int

    @property
    def device(self):
        return self.pb.device


def dedup_from_store_device(dstore: DeviceMolStore, sides: Sequence[np.ndarray], labels: Optional[np.ndarray] = None):
    """The de-duplicated form of ``pack_from_store_device(dstore, sides, labels=...)`` for a two-sided pair batch."""
    if len(sides) != 2 or len(sides[0]) != len(sides[1]):
        raise ValueError("de-duplication works on a two-sided pair batch")
    mids = np.concatenate([np.asarray(s, dtype=np.int64) for s in sides])
    uniq, inv = np.unique(mids, return_inverse=True)
    out = pack_from_store_device(dstore, sides, labels=labels)
    pb, t = (out, None) if labels is None else out
    pb_u = pack_from_store_device(dstore, [uniq.astype(np.int32)])
    U = len(uniq)
    order = np.argsort(inv, kind="stable")
    uptr = np.zeros(U + 1, dtype=np.int64)
    np.cumsum(np.bincount(inv, minlength=U), out=uptr[1:])
    tab = torch.from_numpy(np.concatenate((inv, uptr, order)).astype(np.int32)).to(pb.device)
    I = len(mids)
    dd = DedupPairBatch(pb_u=pb_u, pb=pb, uid=tab[:I], uptr=tab[I:I + U + 1], uinst=tab[I + U + 1:], n_distinct=U)
    return dd if labels is None else (dd, t)


class MolRowsFn(Function):
    """Atom states of the distinct molecules [N_U x d] -> the per-instance rows [N_inst x d] (bmp_molrows_expand); the
    backward sums a molecule's instances in a fixed order (bmp_molrows_reduce)."""

    @staticmethod
    def forward(ctx, hU, dd: DedupPairBatch):
        L = _lib.lib()
        hU = hU.contiguous()
        d = hU.shape[1]
        N = dd.pb.n_rows
        out = torch.empty(N, d, dtype=torch.float32, device=hU.device)
        check(L.bmp_molrows_expand(ptr(hU), d, ptr(dd.pb.row_mol), ptr(dd.pb.mol_row0), ptr(dd.uid), ptr(dd.pb_u.mol_row0), N,
                                   ptr(out), stream()), "bmp_molrows_expand")
        ctx.dd, ctx.nu = dd, hU.shape[0]
        return out

    @staticmethod
    def backward(ctx, dX):
        L = _lib.lib()
        dd = ctx.dd
        dX = dX.contiguous()
        d = dX.shape[1]
        dh = torch.empty(ctx.nu, d, dtype=torch.float32, device=dX.device)
        check(L.bmp_molrows_reduce(ptr(dX), d, ptr(dd.pb_u.row_mol), ptr(dd.pb_u.mol_row0), ptr(dd.uptr), ptr(dd.uinst),
                                   ptr(dd.pb.mol_row0), ctx.nu, ptr(dh), stream()), "bmp_molrows_reduce")
        return dh, None

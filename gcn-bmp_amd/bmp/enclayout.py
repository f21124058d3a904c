"""The encoder's own row layout (csrc/bmp_enc.hip, csrc/bmp_collate.hip: bmp_collate_plan_enc).

The reference zero-pads every molecule of a batch side to the side's largest atom count and masks nothing
(train_ddi_modify.py:296; models/ggnn.py:340,603; nie_coattention.py:347-349).  The packed layout (bmp/packed.py) keeps ONE
virtual pad row per molecule instance.  Inside the encoder all of those rows are one and the same row: atom id 0, no bonds,
hence one state per propagation step for the whole batch (models/ggnn.py:215-263), and their gradients reach the weights and
the embedding only as a sum.  The ENCODER LAYOUT therefore holds

* the REAL atoms of every encoded molecule and ONE pad row per tile (the first row behind the tile's last molecule),
* in tiles of 1..4 live 32-row blocks over dense rows, heights chosen so that the 256 CUs of the chip finish together: a
  1024-pair batch of the DDI set is 6.8 block-rounds of work, which whole 128-row tiles run as 8 (two rounds, the second
  three-quarters full) and tiles of 4 + 3 blocks per CU as 7;
* optionally every DISTINCT molecule of the batch once (``dedup=True``; SURVEY.md 8(d) caveat: a batch of 1024 pairs holds
  2048 instances of about 530 of the 544 drugs, setting.py:30-31) -- reported by bench.py beside the per-instance figure.

The consumers of the atom states -- readout, co-attention -- keep the per-instance layout and its per-instance pad rows
(whose multiplicity carries the side's padding into every softmax and sum): ``EncRowsFn`` copies the rows over
(bmp_encrows_expand) and brings the gradients back, summing what was copied from one row in a fixed order
(bmp_encrows_reduce; no atomics, bitwise reproducible).  Same mathematics as the per-instance encoder, another association
of a few sums (1e-6 relative in float32); tests/test_enclayout.py pins the layout against the dense oracle in float64.
"""
from __future__ import annotations

import time as _time
from ctypes import c_void_p as _c_void_p
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr, stream
from .packed import (DEFAULT_R, DeviceMolStore, MolStore, PackedMolBatch, _i32p, _ragged_arange, pack_from_store,
                     pack_from_store_device)

N_CU = 256                       # MI355X: 256 CUs (the tile-height budget is per CU)
_DEC = {1: (1,), 2: (2,), 3: (3,), 4: (4,), 5: (3, 2), 6: (3, 3), 7: (4, 3)}


@dataclass
class EncBatch:
    pb_enc: PackedMolBatch        # the encoder layout (tile table in mt_row0 / mt_nblk)
    pb: PackedMolBatch            # the per-instance layout: row multiplicities, molecule ranges, pair metadata
    uid: torch.Tensor             # [I]      encoded molecule of every instance
    uptr: torch.Tensor            # [U + 1]  instances of every encoded molecule: uinst[uptr[u] : uptr[u + 1]], ascending
    uinst: torch.Tensor           # [I]
    enc_row0: torch.Tensor        # [U]      first encoder row of every encoded molecule
    enc_n: torch.Tensor           # [U]      its real atoms
    enc_pad: torch.Tensor         # [U]      the pad row of its tile
    tptr: torch.Tensor            # [T + 1]  encoded molecules of every tile: tmols[tptr[t] : tptr[t + 1]], ascending
    tmols: torch.Tensor           # [U]
    n_encoded: int
    dedup: bool
    budget: int                   # block-rounds per CU the tile heights were chosen for (8: uniform 128-row bins)

    @property
    def device(self):
        return self.pb.device


# ---------------------------------------------------------------------------------------------------------
# The plan in numpy: the reference form of bmp_collate_plan_enc (tests pin the C++ plan against it)
# ---------------------------------------------------------------------------------------------------------
def plan_enc_numpy(n_real_store: np.ndarray, nedges_store: np.ndarray, mids: np.ndarray, dedup: bool, n_cu: int = N_CU,
                   R: int = DEFAULT_R):
    mids = np.asarray(mids, dtype=np.int64)
    if dedup:
        umid, uid = np.unique(mids, return_inverse=True)
    else:
        umid, uid = mids.copy(), np.arange(len(mids))
    U = len(umid)
    n = n_real_store[umid].astype(np.int64)
    if int(n.max()) + 1 > R:
        return None
    tot = int(n.sum())
    b = 8
    for k in range(1, 8):
        cap = sum(n_cu * (32 * h - 1) for h in _DEC[k])
        if 100 * tot <= 99 * cap:
            b = k
            break
    caps = [32 * h - 1 for h in (_DEC[b] if b < 8 else ()) for _ in range(n_cu)]
    n_list = len(caps)
    caps = np.array(caps + [R - 1] * U, dtype=np.int64)
    used = np.zeros(len(caps), dtype=np.int64)
    bin_of = np.zeros(U, dtype=np.int64); off_in = np.zeros(U, dtype=np.int64)
    for u in np.argsort(-n, kind="stable"):
        fit = np.nonzero(caps - used >= n[u])[0]
        bn = int(fit[0])
        bin_of[u], off_in[u] = bn, used[bn]
        if b < 8 and bn >= n_list and used[bn] == 0:        # a spare bin is as tall as its first molecule needs
            caps[bn] = 32 * ((n[u] + 1 + 31) // 32) - 1
        used[bn] += n[u]
    order_bins = [q for q in range(n_list, len(caps)) if used[q]] + [q for q in range(n_list) if used[q]]
    tile_of_bin = {q: t for t, q in enumerate(order_bins)}
    mt_nblk = [int((used[q] + 1 + 31) // 32) for q in order_bins]
    T_real = len(order_bins)
    rows = 32 * sum(mt_nblk)
    while rows % R:
        mt_nblk.append(1); rows += 32
    mt_nblk = np.array(mt_nblk, dtype=np.int64)
    mt_row0 = np.cumsum(32 * mt_nblk) - 32 * mt_nblk
    tile = np.array([tile_of_bin[int(q)] for q in bin_of], dtype=np.int64)
    row0 = mt_row0[tile] + off_in
    enc_pad = mt_row0[tile] + used[bin_of]
    ndead = np.zeros(U, dtype=np.int64); tile_last = np.full(U, -1, dtype=np.int64)
    for t in range(T_real):
        mem = np.nonzero(tile == t)[0]
        u = int(mem[np.argmax(off_in[mem])])
        tile_last[u] = t
        ndead[u] = mt_row0[t] + 32 * mt_nblk[t] - (row0[u] + n[u])
        if t == T_real - 1:
            ndead[u] += 32 * (len(mt_nblk) - T_real)
    by_row = np.argsort(row0, kind="stable")
    ne = nedges_store[umid].astype(np.int64)
    ebase = np.zeros(U, dtype=np.int64)
    ebase[by_row] = np.cumsum(ne[by_row]) - ne[by_row]
    uorder = np.argsort(uid, kind="stable")
    uptr = np.zeros(U + 1, dtype=np.int64); np.cumsum(np.bincount(uid, minlength=U), out=uptr[1:])
    torder = np.argsort(tile, kind="stable")
    tptr = np.zeros(len(mt_nblk) + 1, dtype=np.int64); np.cumsum(np.bincount(tile, minlength=len(mt_nblk)), out=tptr[1:])
    return dict(U=U, T=len(mt_nblk), N=int(rows), n_edges=int(ne.sum()), n_real=tot, budget=b, umid=umid, uid=uid, n=n, row0=row0,
                ebase=ebase, ndead=ndead, tile_last=tile_last, enc_pad=enc_pad, uptr=uptr, uinst=uorder, tptr=tptr, tmols=torder,
                mt_row0=mt_row0, mt_nblk=mt_nblk, tile=tile)


def plan_enc_host(st_nrows: np.ndarray, st_nedges: np.ndarray, mids: np.ndarray, dedup: bool, n_cu: int = N_CU, R: int = DEFAULT_R,
                  out: Optional[np.ndarray] = None):
    """bmp_collate_plan_enc on host arrays.  ``out``: int32 buffer of at least ``plan_enc_ints(I)`` entries (a pinned staging
    buffer); returns (views dict, totals) or None when a molecule does not fit a tile."""
    L = _lib.lib()
    mids = np.ascontiguousarray(mids, dtype=np.int32)
    I = len(mids)
    if out is None:
        out = np.empty(plan_enc_ints(I), dtype=np.int32)
    o = 0
    v = {}
    for name, size in _ENC_FIELDS(I):
        v[name] = out[o:o + size]; o += size
    totals = np.zeros(6, dtype=np.int64)
    rc = L.bmp_collate_plan_enc(_i32p(st_nrows), _i32p(st_nedges), len(st_nrows), _i32p(mids), I, int(bool(dedup)), n_cu, R,
                                _i32p(v["tab"]), _i32p(v["tile_last"]), _i32p(v["uid"]), _i32p(v["uptr"]), _i32p(v["uinst"]),
                                _i32p(v["enc_pad"]), _i32p(v["tptr"]), _i32p(v["tmols"]), _i32p(v["mt_row0"]), _i32p(v["mt_nblk"]),
                                _i32p(totals))
    if rc == -2000:
        return None
    check(rc, "bmp_collate_plan_enc")
    return v, [int(x) for x in totals]


def _ENC_FIELDS(I: int):
    return (("tab", 6 * I), ("tile_last", I), ("uid", I), ("uptr", I + 1), ("uinst", I), ("enc_pad", I), ("tptr", I + 4),
            ("tmols", I), ("mt_row0", I + 4), ("mt_nblk", I + 4))


def plan_enc_ints(I: int) -> int:
    return sum(sz for _n, sz in _ENC_FIELDS(I))


# ---------------------------------------------------------------------------------------------------------
# Host (numpy) form of the encoder layout: the CPU path of the tests
# ---------------------------------------------------------------------------------------------------------
def encode_from_store(store: MolStore, sides: Sequence[np.ndarray], dedup: bool = False, device="cpu", n_cu: int = N_CU,
                      R: int = DEFAULT_R, with_dense_map: bool = False) -> EncBatch:
    pb = pack_from_store(store, sides, R=R, device=device, with_dense_map=with_dense_map)
    mids = np.concatenate([np.asarray(s, dtype=np.int64) for s in sides])
    pl = plan_enc_numpy(store.n_atoms, store.nedges, mids, dedup, n_cu, R)
    if pl is None:
        raise ValueError("a molecule of the batch has more than R - 1 atoms: keep the per-instance layout")
    U, N = pl["U"], pl["N"]
    n, row0 = pl["n"], pl["row0"]
    rowmap = _ragged_arange(row0, n)                                   # real atoms of the encoded molecules -> encoder rows
    src_rows = _ragged_arange(store.row_off[pl["umid"]], n)
    atom_id = np.zeros(N, dtype=np.int32); atom_id[rowmap] = store.atom_flat[src_rows]
    row_w = np.zeros(N, dtype=np.float32); row_w[rowmap] = 1.0
    row_mol = np.full(N, -1, dtype=np.int32); row_mol[rowmap] = np.repeat(np.arange(U, dtype=np.int32), n)
    for u in np.nonzero(pl["tile_last"] >= 0)[0]:
        row_mol[row0[u] + n[u]] = -2 - pl["tile_last"][u]
    ne = store.nedges[pl["umid"]]
    eidx = _ragged_arange(store.edge_off[pl["umid"]], ne)
    shift = np.repeat(row0, ne)
    dst, src, typ = store.e_dst[eidx] + shift, store.e_src[eidx] + shift, store.e_typ[eidx]

    def build(major, minor):
        order = np.lexsort((typ, minor, major))
        p_ = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(np.bincount(major, minlength=N), out=p_[1:])
        return p_.astype(np.int32), ((minor[order] << 2) | typ[order]).astype(np.int32), np.ones(len(order), np.float32)

    cp, cc, cv = build(dst, src)
    tp, tc, tv = build(src, dst)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    i32 = lambda a: T(np.asarray(a).astype(np.int32))
    pb_enc = PackedMolBatch(
        R=R, n_tiles=N // R, n_mols=U, atom_id=T(atom_id), row_w=T(row_w), csr_ptr=T(cp), csr_col=T(cc), csr_val=T(cv),
        csrT_ptr=T(tp), csrT_col=T(tc), csrT_val=T(tv), mol_row0=i32(row0), mol_nrows=i32(n), side_tiles=(0, N // R),
        side_mols=(0, U), n_real_atoms=pl["n_real"], n_edges=pl["n_edges"], max_rows_per_mol=int(n.max()),
        mol_nrows_host=n.astype(np.int64), row_mol=T(row_mol),
        atom_id_range=(int(atom_id[rowmap].min()), int(atom_id[rowmap].max())), mt_row0=i32(pl["mt_row0"]), mt_nblk=i32(pl["mt_nblk"]),
        n_mtiles=pl["T"])
    return EncBatch(pb_enc=pb_enc, pb=pb, uid=i32(pl["uid"]), uptr=i32(pl["uptr"]), uinst=i32(pl["uinst"]), enc_row0=pb_enc.mol_row0,
                    enc_n=pb_enc.mol_nrows, enc_pad=i32(pl["enc_pad"]), tptr=i32(pl["tptr"]), tmols=i32(pl["tmols"]), n_encoded=U,
                    dedup=bool(dedup), budget=pl["budget"])


def expand_rows_host(h: torch.Tensor, eb: EncBatch) -> torch.Tensor:
    """Reference semantics of bmp_encrows_expand on host tensors (differentiable: its autograd is the reduce)."""
    pb = eb.pb
    rm = pb.row_mol.long()
    live = rm >= 0
    inst = rm.clamp(min=0)
    u = eb.uid.long()[inst]
    l = torch.arange(pb.n_rows, device=h.device) - pb.mol_row0.long()[inst]
    src = torch.where(l < eb.enc_n.long()[u], eb.enc_row0.long()[u] + l, eb.enc_pad.long()[u])
    out = h[src.clamp(0, h.shape[0] - 1)]
    return torch.where(live[:, None], out, torch.zeros_like(out))


# ---------------------------------------------------------------------------------------------------------
# Device form: the per-iteration collate of bench.py / the trainer
# ---------------------------------------------------------------------------------------------------------
def encode_from_store_device(dstore: DeviceMolStore, sides: Sequence[np.ndarray], labels: Optional[np.ndarray] = None,
                             dedup: bool = False, n_cu: int = N_CU, R: int = DEFAULT_R):
    """The batch in both layouts, collated on the device: the per-instance batch of ``pack_from_store_device`` (row
    multiplicities, pair metadata) and the encoder layout -- one more host plan (bmp_collate_plan_enc), one more pinned copy,
    one more launch of bmp_collate_emit.  Falls back to the per-instance batch alone when a molecule does not fit a tile.
    Returns an EncBatch (or the PackedMolBatch), with the labels on the device as second value when ``labels`` is given."""
    L = _lib.lib()
    dev = dstore.device
    out = pack_from_store_device(dstore, sides, R=R, labels=labels)
    pb, t = (out, None) if labels is None else out
    mids = np.ascontiguousarray(np.concatenate([np.asarray(s) for s in sides]), dtype=np.int32)
    I = len(mids)
    n_ints = plan_enc_ints(I)
    st = dstore._staging(n_ints)
    st_np = st.numpy()
    t0 = _time.perf_counter()
    res = plan_enc_host(dstore.st_nrows, dstore.st_nedges, mids, dedup, n_cu, R, out=st_np)
    dstore.plan_seconds += _time.perf_counter() - t0
    if res is None:
        dstore._staged_unused()
        return pb if labels is None else (pb, t)
    v, (U, T, N, E, n_real, budget) = res
    cur = torch.cuda.current_stream(dev)
    cs = dstore.stream if dstore.stream is not None else cur
    with torch.cuda.stream(cs):
        up = torch.empty(n_ints, dtype=torch.int32, device=dev)
        up.copy_(st[:n_ints], non_blocking=True)
        dstore._staged()
        ibuf = torch.empty(4 * N + 2 + 2 * E, dtype=torch.int32, device=dev)
        fbuf = torch.empty(N + 2 * E, dtype=torch.float32, device=dev)
    atom_id, csr_ptr, csr_col = ibuf[:N], ibuf[N:2 * N + 1], ibuf[2 * N + 1:2 * N + 1 + E]
    o = 2 * N + 1 + E
    csrT_ptr, csrT_col, row_mol = ibuf[o:o + N + 1], ibuf[o + N + 1:o + N + 1 + E], ibuf[o + N + 1 + E:o + 2 * N + 1 + E]
    row_w, csr_val, csrT_val = fbuf[:N], fbuf[N:N + E], fbuf[N + E:]
    dv, off = {}, 0
    for name, size in _ENC_FIELDS(I):
        dv[name] = up[off:off + size]; off += size
    # the plan table is written with stride U (the encoded molecules), the upload keeps the host's offsets
    tab = dv["tab"]
    d = dstore.dev
    check(L.bmp_collate_emit(ptr(tab), U, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), ptr(d[5]), ptr(d[6]),
                             ptr(atom_id), ptr(row_w), ptr(row_mol), ptr(csr_ptr), ptr(csr_col), ptr(csr_val), ptr(csrT_ptr),
                             ptr(csrT_col), ptr(csrT_val), ptr(dv["tile_last"]), _c_void_p(cs.cuda_stream)), "bmp_collate_emit")
    if cs is not cur:
        for x in (up, ibuf, fbuf):
            x.record_stream(cur)
        cur.wait_stream(cs)
    n_host = v["tab"][U:2 * U].astype(np.int64)
    pb_enc = PackedMolBatch(
        R=R, n_tiles=N // R, n_mols=U, atom_id=atom_id, row_w=row_w, csr_ptr=csr_ptr, csr_col=csr_col, csr_val=csr_val,
        csrT_ptr=csrT_ptr, csrT_col=csrT_col, csrT_val=csrT_val, mol_row0=tab[:U], mol_nrows=tab[U:2 * U], side_tiles=(0, N // R),
        side_mols=(0, U), n_real_atoms=n_real, n_edges=E, max_rows_per_mol=int(n_host.max()), mol_nrows_host=n_host, row_mol=row_mol,
        atom_id_range=dstore.atom_id_range, mt_row0=dv["mt_row0"][:T], mt_nblk=dv["mt_nblk"][:T], n_mtiles=T)
    eb = EncBatch(pb_enc=pb_enc, pb=pb, uid=dv["uid"], uptr=dv["uptr"][:U + 1], uinst=dv["uinst"], enc_row0=tab[:U],
                  enc_n=tab[U:2 * U], enc_pad=dv["enc_pad"][:U], tptr=dv["tptr"][:T + 1], tmols=dv["tmols"][:U], n_encoded=U,
                  dedup=bool(dedup), budget=budget)
    return eb if labels is None else (eb, t)


class EncRowsFn(Function):
    """Atom states in the encoder layout [N_enc x d] -> the per-instance rows [N_inst x d] (bmp_encrows_expand); the backward
    sums, per encoder row and in a fixed order, the gradients of the instance rows copied from it (bmp_encrows_reduce)."""

    @staticmethod
    def forward(ctx, h, eb: EncBatch):
        ctx.eb, ctx.n_enc = eb, h.shape[0]
        if not h.is_cuda:
            raise ValueError("EncRowsFn runs on the GPU (host tensors: enclayout.expand_rows_host)")
        L = _lib.lib()
        h = h.contiguous()
        d = h.shape[1]
        N = eb.pb.n_rows
        out = torch.empty(N, d, dtype=torch.float32, device=h.device)
        check(L.bmp_encrows_expand(ptr(h), d, ptr(eb.pb.row_mol), ptr(eb.pb.mol_row0), ptr(eb.uid), ptr(eb.enc_row0), ptr(eb.enc_n),
                                   ptr(eb.enc_pad), N, ptr(out), stream()), "bmp_encrows_expand")
        return out

    @staticmethod
    def backward(ctx, dX):
        L = _lib.lib()
        eb = ctx.eb
        dX = dX.contiguous()
        d = dX.shape[1]
        dh = torch.empty(ctx.n_enc, d, dtype=torch.float32, device=dX.device)
        check(L.bmp_encrows_reduce(ptr(dX), d, ptr(eb.pb_enc.row_mol), ptr(eb.enc_row0), ptr(eb.enc_n), ptr(eb.uptr), ptr(eb.uinst),
                                   ptr(eb.pb.mol_row0), ptr(eb.tptr), ptr(eb.tmols), ctx.n_enc, ptr(dh), stream()),
              "bmp_encrows_reduce")
        return dh, None

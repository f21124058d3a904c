"""Packed molecule batches: the HBM layout the HIP kernels work on.

The reference batches molecules by zero-padding every molecule of a batch side
to the side's max atom count A (``concat_mols``, train_ddi_modify.py:296) and
carries a dense (mb, 4, A, A) adjacency that is ~99 % zeros.  Padded atoms are
NOT masked anywhere (models/ggnn.py:340,603; nie_coattention.py:347-349), so
results depend on A.  The packed layout reproduces those semantics exactly
without doing the padded work:

* rows -- every molecule contributes its real atoms plus ONE *virtual pad row*
  (atom id 0, no bonds) whose multiplicity ``row_w`` = A - n stands for all of
  its zero-padded positions (they share one trajectory: same embedding row,
  message 0).  Real rows have ``row_w`` = 1.  Every sum over atoms in the
  reference (readout, softmax denominators, attention-weighted sums) becomes a
  ``row_w``-weighted sum.
* tiles -- rows are laid out in tiles of R rows (R multiple of 32, default 128);
  a molecule never straddles a tile, so all neighbour gathers are tile-local
  (LDS-resident in the fused kernels).  Unused rows at a tile's end are *dead
  rows* (``row_w`` = 0, id 0, no edges).  All row-indexed buffers are
  (n_tiles*R, .) so GEMM kernels need no bounds checks.
* bonds -- CSR over destination rows: ``csr_col`` = (src_row << 2) | bond_type,
  ``csr_val`` = adjacency value (1.0; 1/deg after RelGCN's rescale_adj,
  models/relgcn.py:20-28).  ``csrT_*`` is the transpose (by source row) used by
  the backward pass; for the reference's symmetric adjacency it equals the CSR
  but it is always built so asymmetric dense inputs stay exact.

Index work here is integer-exact and is checked bit-for-bit against the dense
form in tests/test_packed.py.
"""
from __future__ import annotations

import os
import time as _time
from ctypes import c_void_p as _c_void_p
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .synth import Molecule, NUM_EDGE_TYPE

DEFAULT_R = 128


def _ragged_arange(starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """concatenate([arange(s, s+l) for s, l in zip(starts, lens)]) without a Python loop."""
    lens = lens.astype(np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    excl = np.cumsum(lens) - lens
    return np.repeat(starts.astype(np.int64) - excl, lens) + np.arange(total, dtype=np.int64)


class MolStore:
    """Flat per-molecule local CSR of a drug store (host, numpy).  Row space per
    molecule: n real atoms followed by the virtual pad row."""

    def __init__(self, mols: Sequence[Molecule]):
        self.n_mols = len(mols)
        self.n_atoms = np.array([m.n for m in mols], dtype=np.int64)
        self.nrows = self.n_atoms + 1
        self.row_off = np.cumsum(self.nrows) - self.nrows
        self.atom_flat = np.zeros(int(self.nrows.sum()), dtype=np.int32)
        e_dst, e_src, e_typ = [], [], []
        self.nedges = np.zeros(self.n_mols, dtype=np.int64)
        for k, m in enumerate(mols):
            o = self.row_off[k]
            self.atom_flat[o:o + m.n] = m.atoms
            if len(m.bonds):
                i, j, t = m.bonds[:, 0].astype(np.int64), m.bonds[:, 1].astype(np.int64), m.bonds[:, 2]
                e_dst.append(np.concatenate((i, j)))          # local row ids
                e_src.append(np.concatenate((j, i)))
                e_typ.append(np.concatenate((t, t)))
                self.nedges[k] = 2 * len(i)
            else:
                e_dst.append(np.zeros(0, np.int64)); e_src.append(np.zeros(0, np.int64))
                e_typ.append(np.zeros(0, np.int32))
        self.edge_off = np.cumsum(self.nedges) - self.nedges
        self.e_dst = np.concatenate(e_dst).astype(np.int64)
        self.e_src = np.concatenate(e_src).astype(np.int64)
        self.e_typ = np.concatenate(e_typ).astype(np.int64)


def _bin_pack(sizes: np.ndarray, R: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """First-fit-decreasing of item sizes into bins of capacity R.
    Returns (bin index, offset inside bin) per item and the bin count.

    An item LARGER than a bin (a molecule of more than R - 1 atoms: the reference's preprocessor has no size limit,
    train_ddi_modify.py:256) takes ceil(size / R) whole consecutive bins of its own, from a bin boundary; such items come
    first, in the stable decreasing-size order, and what is left of their last bin stays empty (dead rows).  The batch
    then holds rows whose neighbours live in another tile, and the encoders take the row-wise operators for it (bmp/ggnn.py)."""
    order = np.argsort(-sizes, kind="stable")
    caps = np.full(len(sizes) + 1, R, dtype=np.int64)
    nb = 0
    bins = np.zeros(len(sizes), dtype=np.int64)
    offs = np.zeros(len(sizes), dtype=np.int64)
    first = 0                      # bins [0, first) belong to the oversized items
    for it in order:
        s = int(sizes[it])
        if s <= R:
            break
        bins[it] = first
        first += (s + R - 1) // R
    for it in order:
        s = int(sizes[it])
        if s > R:
            continue
        fit = caps[:nb] >= s
        b = int(np.argmax(fit)) if nb and fit.any() else nb
        if b == nb:
            nb += 1
        bins[it] = first + b
        offs[it] = R - caps[b]
        caps[b] -= s
    return bins, offs, first + nb


@dataclass
class PackedMolBatch:
    """Device-resident packed batch (see module docstring).  All index tensors int32."""
    R: int
    n_tiles: int
    n_mols: int
    atom_id: torch.Tensor          # (N,)
    row_w: torch.Tensor            # (N,) float32
    csr_ptr: torch.Tensor          # (N+1,)
    csr_col: torch.Tensor          # (E,)  src_row<<2 | type
    csr_val: torch.Tensor          # (E,) float32
    csrT_ptr: torch.Tensor
    csrT_col: torch.Tensor         # (E,)  dst_row<<2 | type
    csrT_val: torch.Tensor
    mol_row0: torch.Tensor         # (n_mols,)
    mol_nrows: torch.Tensor        # (n_mols,)  real atoms + 1
    side_tiles: Tuple[int, ...] = (0,)            # tile boundaries of the sides: (0, T1[, T1+T2])
    side_mols: Tuple[int, ...] = (0,)             # molecule boundaries of the sides
    dense_map: Optional[torch.Tensor] = None      # (n_mols, A) int64 row of every dense position (one side only)
    dense_maps: Optional[List[torch.Tensor]] = None   # per side
    n_real_atoms: int = 0
    n_edges: int = 0
    max_rows_per_mol: int = 0
    mol_nrows_host: Optional[np.ndarray] = None   # host copy of mol_nrows (pair metadata without a device sync)
    row_mol: Optional[torch.Tensor] = None        # (N,) molecule of every row, -1 for rows of no molecule
    atom_id_range: Tuple[int, int] = (0, 0)       # (min, max) atom id of the batch, known on the host at pack time
    # The encoder layout (bmp/enclayout.py): molecule tiles of 1..4 live 32-row blocks over dense rows -- tile t = rows
    # [mt_row0[t], mt_row0[t] + 32 * mt_nblk[t]).  None: tile t = rows [R t, R t + R) and n_mtiles == n_tiles.
    mt_row0: Optional[torch.Tensor] = None
    mt_nblk: Optional[torch.Tensor] = None
    n_mtiles: int = -1
    # 0: the tile table covers dense rows (tile t + 1 follows tile t).  R: tile t starts at row R t whatever its height (the
    # fixed-shape batch below); the fused step kernels then clear the rows of a tile's dead blocks in what they write.
    tile_stride: int = 0
    _cache: dict = field(default_factory=dict, repr=False)

    def __post_init__(self):
        if self.n_mtiles < 0:
            self.n_mtiles = self.n_tiles

    @property
    def n_rows(self) -> int:
        return self.n_tiles * self.R

    @property
    def device(self) -> torch.device:
        return self.atom_id.device

    @property
    def oversized(self) -> bool:
        """A molecule of this batch spans more than one tile (more than R - 1 atoms): neighbour gathers are not tile-local,
        the encoders take the row-wise operators (gather over global rows, row GEMMs, per-molecule segment sums) and the
        co-attention the global-memory class of its pair kernels."""
        return self.max_rows_per_mol > self.R

    def check_atom_ids(self, n_atom_types: int) -> None:
        """EmbedID's type check (chainer rejects ids outside the table; models/ggnn.py:85,603): raise before any kernel
        indexes the embedding table.  The range was taken on the host when the batch was packed: no device sync."""
        lo, hi = self.atom_id_range
        if lo < 0 or hi >= n_atom_types:
            raise ValueError(f"atom ids must lie in [0, {n_atom_types}); this batch has ids in [{lo}, {hi}]")

    def type_rows_F(self):
        """``type_rows_T`` for the FORWARD CSR: the rows whose gathered features agg_e are not zero (the unfused message
        operator's weight gradient dWT = agg^T dpre walks them)."""
        return self.type_rows_T(forward=True)

    def type_rows_T(self, forward: bool = False):
        """(idx [4 x N] int32, cnt [4] int32), device tensors: the rows of the TRANSPOSED CSR that hold an entry of bond type e
        (``bmp_type_rows``) -- the rows whose gathered gradient G_e is not zero, which is all the step's weight-gradient
        launches need to walk (73 / 19 / 2 / 52 % of a DDI batch's rows).  Built once per batch on the caller's stream (two small
        launches) and kept; host batches: None."""
        if not self.atom_id.is_cuda:
            return None
        key = "type_rows_F" if forward else "type_rows_T"
        tr = self._cache.get(key)
        if tr is None:
            from . import _lib
            from ._lib import check, ptr, stream
            L = _lib.lib()
            N, dev = self.n_rows, self.atom_id.device
            live = bool(self.tile_stride) and not forward and self.row_mol is not None
            nl = 5 if live else 4
            idx = torch.empty(nl * N, dtype=torch.int32, device=dev)
            cnt = torch.empty(nl, dtype=torch.int32, device=dev)
            ws = torch.empty(max(int(L.bmp_type_rows_ws_ints(N)), 8), dtype=torch.int32, device=dev)
            cp, cc = (self.csr_ptr, self.csr_col) if forward else (self.csrT_ptr, self.csrT_col)
            if live:        # tiles at a fixed stride: most rows belong to no molecule, and a fifth list names those that do
                check(L.bmp_type_rows_live(ptr(cp), ptr(cc), ptr(self.row_mol), N, ptr(idx), ptr(cnt), ptr(ws), stream()),
                      "bmp_type_rows_live")
                self._cache["live_rows"] = (idx[4 * N:], cnt[4:])
            else:
                check(L.bmp_type_rows(ptr(cp), ptr(cc), N, ptr(idx), ptr(cnt), ptr(ws), stream()), "bmp_type_rows")
            tr = (idx, cnt)
            self._cache[key] = tr
        return tr

    def with_edge_vals(self, csr_val: torch.Tensor, csrT_val: torch.Tensor) -> "PackedMolBatch":
        import dataclasses
        return dataclasses.replace(self, csr_val=csr_val, csrT_val=csrT_val, _cache={})

    def to_dense(self, rows: torch.Tensor, side: int = 0) -> torch.Tensor:
        """Expand a (N, c) row tensor to the reference's dense (mb, A, c) layout."""
        dm = self.dense_maps[side] if self.dense_maps is not None else self.dense_map
        if dm is None:
            raise ValueError("this batch has no dense position map")
        return rows[dm]


def _assemble(inst_nrows: np.ndarray, flat_atom: np.ndarray, flat_w: np.ndarray,
              e_dst: np.ndarray, e_src: np.ndarray, e_typ: np.ndarray, e_val: np.ndarray,
              side_of_inst: np.ndarray, n_sides: int, R: int, device,
              dense_maps_flat: Optional[List[np.ndarray]] = None, n_real_atoms: Optional[int] = None) -> PackedMolBatch:
    """Common placement step: bin-pack instances into tiles (per side), remap rows
    and edges, build CSR + transposed CSR, upload."""
    I = len(inst_nrows)
    flat_off = np.cumsum(inst_nrows) - inst_nrows
    inst_row0 = np.zeros(I, dtype=np.int64)
    side_tiles = [0]
    side_mols = [0]
    tile0 = 0
    for s in range(n_sides):
        sel = np.nonzero(side_of_inst == s)[0]
        bins, offs, nb = _bin_pack(inst_nrows[sel], R)
        inst_row0[sel] = (tile0 + bins) * R + offs
        tile0 += nb
        side_tiles.append(tile0)
        side_mols.append(side_mols[-1] + len(sel))
    n_tiles = tile0
    N = n_tiles * R
    rowmap = _ragged_arange(inst_row0, inst_nrows)              # flat row -> packed row
    atom_id = np.zeros(N, dtype=np.int32)
    row_w = np.zeros(N, dtype=np.float32)
    atom_id[rowmap] = flat_atom
    row_w[rowmap] = flat_w
    dst = rowmap[e_dst] if len(e_dst) else np.zeros(0, np.int64)
    src = rowmap[e_src] if len(e_src) else np.zeros(0, np.int64)

    def build(major: np.ndarray, minor: np.ndarray):
        order = np.lexsort((e_typ, minor, major))
        ptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(np.bincount(major, minlength=N), out=ptr[1:])
        col = ((minor[order] << 2) | e_typ[order]).astype(np.int32)
        return ptr.astype(np.int32), col, e_val[order].astype(np.float32)

    csr_ptr, csr_col, csr_val = build(dst, src)
    csrT_ptr, csrT_col, csrT_val = build(src, dst)

    row_mol = np.full(N, -1, dtype=np.int32)
    row_mol[rowmap] = np.repeat(np.arange(I, dtype=np.int32), inst_nrows)
    ints = [atom_id, csr_ptr, csr_col, csrT_ptr, csrT_col, inst_row0.astype(np.int32), inst_nrows.astype(np.int32), row_mol]
    flts = [row_w, csr_val, csrT_val]
    ibuf = torch.from_numpy(np.concatenate(ints)).to(device)
    fbuf = torch.from_numpy(np.concatenate(flts)).to(device)
    iv, o = [], 0
    for a in ints:
        iv.append(ibuf[o:o + len(a)]); o += len(a)
    fv, o = [], 0
    for a in flts:
        fv.append(fbuf[o:o + len(a)]); o += len(a)
    dmaps = None
    if dense_maps_flat is not None:
        dmaps = [torch.from_numpy(rowmap[dm]).to(device) for dm in dense_maps_flat]
    return PackedMolBatch(
        R=R, n_tiles=n_tiles, n_mols=I,
        atom_id=iv[0], row_w=fv[0],
        csr_ptr=iv[1], csr_col=iv[2], csr_val=fv[1],
        csrT_ptr=iv[3], csrT_col=iv[4], csrT_val=fv[2],
        mol_row0=iv[5], mol_nrows=iv[6],
        side_tiles=tuple(side_tiles), side_mols=tuple(side_mols),
        dense_map=dmaps[0] if dmaps is not None and len(dmaps) == 1 else None,
        dense_maps=dmaps,
        # every instance is its real rows + ONE virtual pad row (whose weight may happen to be 1: not a real atom)
        n_real_atoms=int((inst_nrows - 1).sum()) if n_real_atoms is None else int(n_real_atoms), n_edges=int(len(e_dst)),
        max_rows_per_mol=int(inst_nrows.max()) if I else 0,
        mol_nrows_host=inst_nrows.astype(np.int64), row_mol=iv[7],
        atom_id_range=(int(flat_atom.min()), int(flat_atom.max())) if len(flat_atom) else (0, 0),
    )


def pack_from_store(store: MolStore, sides: Sequence[np.ndarray], R: int = DEFAULT_R, device="cpu",
                    with_dense_map: bool = False, pad_to: Optional[Sequence[int]] = None) -> PackedMolBatch:
    """Pack molecule instances ``sides[s][b]`` (indices into the store).  Each side is
    zero-padded to its own max atom count (concat_mols pads every field of the
    batch independently), or to ``pad_to[s]`` when given."""
    mids = np.concatenate([np.asarray(s, dtype=np.int64) for s in sides])
    side_of = np.concatenate([np.full(len(s), k, dtype=np.int64) for k, s in enumerate(sides)])
    n = store.n_atoms[mids]
    A = np.zeros(len(sides), dtype=np.int64)
    for k in range(len(sides)):
        A[k] = int(n[side_of == k].max()) if pad_to is None else int(pad_to[k])
    inst_nrows = n + 1
    flat_off = np.cumsum(inst_nrows) - inst_nrows
    src_rows = _ragged_arange(store.row_off[mids], inst_nrows)
    flat_atom = store.atom_flat[src_rows]
    flat_w = np.ones(len(flat_atom), dtype=np.float32)
    flat_w[flat_off + n] = (A[side_of] - n).astype(np.float32)      # virtual pad rows
    ne = store.nedges[mids]
    eidx = _ragged_arange(store.edge_off[mids], ne)
    eshift = np.repeat(flat_off, ne)
    e_dst = store.e_dst[eidx] + eshift
    e_src = store.e_src[eidx] + eshift
    e_typ = store.e_typ[eidx]
    e_val = np.ones(len(eidx), dtype=np.float32)
    dmf = None
    if with_dense_map:
        dmf = []
        for k in range(len(sides)):
            sel = np.nonzero(side_of == k)[0]
            a = np.arange(A[k])[None, :]
            dm = np.where(a < n[sel][:, None], flat_off[sel][:, None] + a, (flat_off[sel] + n[sel])[:, None])
            dmf.append(dm)
    return _assemble(inst_nrows, flat_atom, flat_w, e_dst, e_src, e_typ, e_val, side_of, len(sides), R, device, dmf)


def pack_from_dense(atom_arrays: Sequence[np.ndarray], adjs: Sequence[np.ndarray], R: int = DEFAULT_R,
                    device="cpu") -> PackedMolBatch:
    """Pack the reference's dense batch form: per side ``atom_array`` (mb, A) int32 and
    ``adj`` (mb, 4, A, A) float32.  A position is merged into the
    molecule's virtual pad row iff its atom id is 0 and it has no incoming bond --
    exactly the positions that follow the shared pad trajectory.  Integer-exact."""
    inst_nrows_l, flat_atom_l, flat_w_l = [], [], []
    e_dst_l, e_src_l, e_typ_l, e_val_l, side_l, dmf = [], [], [], [], [], []
    flat_base = 0
    for k, (atoms, adj) in enumerate(zip(atom_arrays, adjs)):
        atoms = np.asarray(atoms)
        adj = np.asarray(adj, dtype=np.float32)
        mb, A = atoms.shape
        if adj.shape != (mb, NUM_EDGE_TYPE, A, A):
            raise ValueError(f"adj shape {adj.shape} does not match atoms {atoms.shape}")
        nzb, nze, nzi, nzj = np.nonzero(adj)
        indeg = np.zeros((mb, A), dtype=np.int64)
        np.add.at(indeg, (nzb, nzi), 1)
        padlike = (atoms == 0) & (indeg == 0)
        real = ~padlike
        n = real.sum(axis=1).astype(np.int64)
        local = np.cumsum(real, axis=1) - 1
        nrows = n + 1
        off = flat_base + np.cumsum(nrows) - nrows
        dm = np.where(real, off[:, None] + local, (off + n)[:, None])          # (mb, A) flat row
        fa = np.zeros(int(nrows.sum()), dtype=np.int32)
        fw = np.ones(int(nrows.sum()), dtype=np.float32)
        fa[dm[real] - flat_base] = atoms[real]
        fw[off + n - flat_base] = padlike.sum(axis=1).astype(np.float32)
        inst_nrows_l.append(nrows); flat_atom_l.append(fa); flat_w_l.append(fw)
        e_dst_l.append(dm[nzb, nzi]); e_src_l.append(dm[nzb, nzj]); e_typ_l.append(nze.astype(np.int64))
        e_val_l.append(adj[nzb, nze, nzi, nzj])
        side_l.append(np.full(mb, k, dtype=np.int64))
        dmf.append(dm)
        flat_base += int(nrows.sum())
    return _assemble(np.concatenate(inst_nrows_l), np.concatenate(flat_atom_l), np.concatenate(flat_w_l),
                     np.concatenate(e_dst_l), np.concatenate(e_src_l), np.concatenate(e_typ_l),
                     np.concatenate(e_val_l), np.concatenate(side_l), len(atom_arrays), R, device, dmf)


def unpack_to_dense(pb: PackedMolBatch, side: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Inverse of pack_from_dense for one side (host, for tests): rebuild
    (atoms (mb, A) int32, adj (mb, 4, A, A) float32) from the packed arrays."""
    dm = pb.dense_maps[side].cpu().numpy()
    mb, A = dm.shape
    atom_id = pb.atom_id.cpu().numpy()
    atoms = atom_id[dm].astype(np.int32)
    adj = np.zeros((mb, NUM_EDGE_TYPE, A, A), dtype=np.float32)
    ptr = pb.csr_ptr.cpu().numpy().astype(np.int64)
    col = pb.csr_col.cpu().numpy().astype(np.int64)
    val = pb.csr_val.cpu().numpy()
    N = pb.n_rows
    row_of_edge = np.repeat(np.arange(N), np.diff(ptr))
    # first dense position of every packed row (virtual rows map to their first pad position)
    pos_b = np.full(N, -1, dtype=np.int64)
    pos_a = np.full(N, -1, dtype=np.int64)
    bb, aa = np.meshgrid(np.arange(mb), np.arange(A), indexing="ij")
    pos_b[dm[::-1, ::-1].ravel()] = bb[::-1, ::-1].ravel()
    pos_a[dm[::-1, ::-1].ravel()] = aa[::-1, ::-1].ravel()
    src = col >> 2
    keep = (pos_b[row_of_edge] >= 0) & (pos_b[src] >= 0)
    adj[pos_b[row_of_edge][keep], (col & 3)[keep], pos_a[row_of_edge][keep], pos_a[src][keep]] = val[keep]
    return atoms, adj


def pack_from_dense_device(atom_arrays: Sequence, adjs: Sequence[torch.Tensor], R: int = DEFAULT_R) -> PackedMolBatch:
    """pack_from_dense for adjacency tensors that already live on the GPU (the reference's call form hands the
    encoders dense device arrays, train_binary.py:85-89): the dense (mb, 4, A, A) array -- 260 MB for 1024 pairs --
    never crosses PCIe.  The device counts the bonds per position (bmp_dense_count); the host, which only needs
    those counts and the atom ids (mb x A integers), places the molecules into tiles exactly as pack_from_dense does;
    the device then writes the CSR entries in the host packer's order (bmp_dense_to_csr).  Bit-identical to
    pack_from_dense (tests/test_gpu_dense.py).  Inputs with bonds leaving a padded position (atom id 0, no incoming
    bond: only possible with an asymmetric adjacency) fall back to the host packer."""
    from . import _lib
    from ._lib import check, ptr, stream
    L = _lib.lib()
    dev = adjs[0].device
    atoms_np, counts = [], []
    for atoms, adj in zip(atom_arrays, adjs):
        a = atoms.detach().cpu().numpy() if isinstance(atoms, torch.Tensor) else np.asarray(atoms)
        if adj.dtype != torch.float32 or not adj.is_cuda or not adj.is_contiguous():
            raise ValueError("adjacency must be a contiguous float32 CUDA tensor")
        mb, A = a.shape
        if tuple(adj.shape) != (mb, NUM_EDGE_TYPE, A, A):
            raise ValueError(f"adj shape {tuple(adj.shape)} does not match atoms {a.shape}")
        cnt = torch.empty(2, mb, A, dtype=torch.int32, device=dev)
        check(L.bmp_dense_count(ptr(adj), mb, A, ptr(cnt[0]), ptr(cnt[1]), stream()), "bmp_dense_count")
        atoms_np.append(a.astype(np.int32)); counts.append(cnt)
    counts = [c.cpu().numpy().astype(np.int64) for c in counts]              # one small D2H per side
    inst_nrows_l, flat_atom_l, flat_w_l, side_l, dmf, rown_l, coln_l = [], [], [], [], [], [], []
    flat_base = 0
    for k, (atoms, (indeg, outdeg)) in enumerate(zip(atoms_np, counts)):
        mb, A = atoms.shape
        padlike = (atoms == 0) & (indeg == 0)
        if (outdeg[padlike] != 0).any():
            return pack_from_dense(atoms_np, [a.cpu().numpy() for a in adjs], R=R, device=dev)
        real = ~padlike
        n = real.sum(axis=1).astype(np.int64)
        local = np.cumsum(real, axis=1) - 1
        nrows = n + 1
        off = flat_base + np.cumsum(nrows) - nrows
        dm = np.where(real, off[:, None] + local, (off + n)[:, None])          # (mb, A) flat row
        fa = np.zeros(int(nrows.sum()), dtype=np.int32)
        fw = np.ones(int(nrows.sum()), dtype=np.float32)
        fa[dm[real] - flat_base] = atoms[real]
        fw[off + n - flat_base] = padlike.sum(axis=1).astype(np.float32)
        inst_nrows_l.append(nrows); flat_atom_l.append(fa); flat_w_l.append(fw)
        side_l.append(np.full(mb, k, dtype=np.int64)); dmf.append(dm)
        rown_l.append(indeg); coln_l.append(outdeg)
        flat_base += int(nrows.sum())
    inst_nrows = np.concatenate(inst_nrows_l)
    side_of = np.concatenate(side_l)
    I = len(inst_nrows)
    inst_row0 = np.zeros(I, dtype=np.int64)
    side_tiles, side_mols, tile0 = [0], [0], 0
    for s_ in range(len(atoms_np)):
        sel = np.nonzero(side_of == s_)[0]
        bins, offs, nb = _bin_pack(inst_nrows[sel], R)
        inst_row0[sel] = (tile0 + bins) * R + offs
        tile0 += nb
        side_tiles.append(tile0); side_mols.append(side_mols[-1] + len(sel))
    n_tiles = tile0
    N = n_tiles * R
    rowmap = _ragged_arange(inst_row0, inst_nrows)                              # flat row -> packed row
    flat_atom, flat_w = np.concatenate(flat_atom_l), np.concatenate(flat_w_l)
    atom_id = np.zeros(N, dtype=np.int32); row_w = np.zeros(N, dtype=np.float32)
    atom_id[rowmap] = flat_atom; row_w[rowmap] = flat_w
    deg_in = np.zeros(N, dtype=np.int64); deg_out = np.zeros(N, dtype=np.int64)
    dense_rows = [rowmap[dm] for dm in dmf]                                     # per side: packed row of every dense position
    for dr, rn, cn in zip(dense_rows, rown_l, coln_l):
        np.add.at(deg_in, dr.ravel(), rn.ravel()); np.add.at(deg_out, dr.ravel(), cn.ravel())
    csr_ptr = np.zeros(N + 1, dtype=np.int64); np.cumsum(deg_in, out=csr_ptr[1:])
    csrT_ptr = np.zeros(N + 1, dtype=np.int64); np.cumsum(deg_out, out=csrT_ptr[1:])
    E = int(csr_ptr[-1])
    row_mol = np.full(N, -1, dtype=np.int32)
    row_mol[rowmap] = np.repeat(np.arange(I, dtype=np.int32), inst_nrows)
    ints = [atom_id, csr_ptr.astype(np.int32), csrT_ptr.astype(np.int32), inst_row0.astype(np.int32),
            inst_nrows.astype(np.int32)] + [dr.astype(np.int32).ravel() for dr in dense_rows] + [row_mol]
    ibuf = torch.from_numpy(np.concatenate(ints)).to(dev)
    iv, o = [], 0
    for a in ints:
        iv.append(ibuf[o:o + len(a)]); o += len(a)
    row_w_d = torch.from_numpy(row_w).to(dev)
    col = torch.empty(max(E, 1), dtype=torch.int32, device=dev); val = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
    colT = torch.empty_like(col); valT = torch.empty_like(val)
    for k, adj in enumerate(adjs):
        mb, A = atoms_np[k].shape
        check(L.bmp_dense_to_csr(ptr(adj), mb, A, ptr(iv[5 + k]), ptr(iv[1]), 0, ptr(col), ptr(val), stream()), "bmp_dense_to_csr")
        check(L.bmp_dense_to_csr(ptr(adj), mb, A, ptr(iv[5 + k]), ptr(iv[2]), 1, ptr(colT), ptr(valT), stream()), "bmp_dense_to_csr")
    dmaps = [torch.from_numpy(dr).to(dev) for dr in dense_rows]
    return PackedMolBatch(
        R=R, n_tiles=n_tiles, n_mols=I, atom_id=iv[0], row_w=row_w_d,
        csr_ptr=iv[1], csr_col=col[:E], csr_val=val[:E], csrT_ptr=iv[2], csrT_col=colT[:E], csrT_val=valT[:E],
        mol_row0=iv[3], mol_nrows=iv[4], side_tiles=tuple(side_tiles), side_mols=tuple(side_mols),
        dense_map=dmaps[0] if len(dmaps) == 1 else None, dense_maps=dmaps,
        n_real_atoms=int((inst_nrows - 1).sum()), n_edges=E, max_rows_per_mol=int(inst_nrows.max()) if I else 0,
        mol_nrows_host=inst_nrows.astype(np.int64), row_mol=iv[-1],
        atom_id_range=(int(flat_atom.min()), int(flat_atom.max())) if len(flat_atom) else (0, 0))


# ---------------------------------------------------------------------------------------------------------
# The per-iteration collate on the device (train_ddi_modify.py:280 SerialIterator + :295-296 concat_mols):
# the drug store lives in HBM, a batch is (idx1, idx2) -> one host planning call (size arithmetic only,
# bmp_collate_plan), one small H2D copy, one kernel (bmp_collate_emit).  csrc/bmp_collate.hip.
# ---------------------------------------------------------------------------------------------------------
def _i32p(a: np.ndarray):
    return _c_void_p(a.__array_interface__["data"][0])       # (ndarray.ctypes.data_as costs ~60 us per call)


def collate_plan_host(st_nrows: np.ndarray, st_nedges: np.ndarray, sides: Sequence[np.ndarray], R: int = DEFAULT_R,
                      pad_to: Optional[Sequence[int]] = None, tab: Optional[np.ndarray] = None):
    """bmp_collate_plan on host arrays (no device work): returns (tab int32 [6*I], side_tiles, side_mols,
    n_tiles, n_edges, n_real_atoms, max_rows)."""
    from . import _lib
    L = _lib.lib()
    mids = np.ascontiguousarray(np.concatenate([np.asarray(s) for s in sides]), dtype=np.int32)
    side_ptr = np.zeros(len(sides) + 1, dtype=np.int32)
    np.cumsum([len(s) for s in sides], out=side_ptr[1:])
    I = int(side_ptr[-1])
    if tab is None:
        tab = np.empty(6 * I, dtype=np.int32)
    side_tiles = np.zeros(len(sides) + 1, dtype=np.int32)
    totals = np.zeros(4, dtype=np.int64)
    pt = None if pad_to is None else np.ascontiguousarray(pad_to, dtype=np.int32)
    _lib.check(L.bmp_collate_plan(_i32p(st_nrows), _i32p(st_nedges), len(st_nrows), _i32p(mids), _i32p(side_ptr), len(sides),
                                  None if pt is None else _i32p(pt), R, _i32p(tab), _i32p(side_tiles), _i32p(totals)),
               "bmp_collate_plan")
    return (tab, tuple(int(x) for x in side_tiles), tuple(int(x) for x in side_ptr), int(totals[0]), int(totals[1]),
            int(totals[2]), int(totals[3]))


class DeviceMolStore:
    """The drug store resident in HBM: per molecule its atom ids and local CSR / transposed CSR, sorted as a packed
    batch sorts them (destination row, source row, bond type).  ~0.3 MB for the 544 drugs of the binary DDI set."""

    N_STAGE = 8        # pinned staging buffers in flight (a buffer is reused only after its copy has completed)

    def __init__(self, store: MolStore, device):
        self.host = store
        self.device = torch.device(device)
        M = store.n_mols
        self.st_nrows = np.ascontiguousarray(store.nrows, dtype=np.int32)
        self.st_nedges = np.ascontiguousarray(store.nedges, dtype=np.int32)
        rowoff = np.zeros(M + 1, dtype=np.int64); np.cumsum(store.nrows, out=rowoff[1:])
        eoff = np.zeros(M + 1, dtype=np.int64); np.cumsum(store.nedges, out=eoff[1:])
        mol_of_edge = np.repeat(np.arange(M, dtype=np.int64), store.nedges)

        def local_csr(major, minor):
            order = np.lexsort((store.e_typ, minor, major, mol_of_edge))
            col = ((minor[order] << 2) | store.e_typ[order]).astype(np.int32)
            deg = np.bincount(rowoff[mol_of_edge] + major, minlength=int(rowoff[-1]))
            rend = np.cumsum(deg) - np.repeat(eoff[:-1], store.nrows)            # end of the row's entries, local
            return col, rend.astype(np.int32)

        col, rend = local_csr(store.e_dst, store.e_src)
        colT, rendT = local_csr(store.e_src, store.e_dst)
        host = [rowoff.astype(np.int32), eoff.astype(np.int32), store.atom_flat.astype(np.int32), rend, rendT, col, colT]
        self.host_arrays = host
        self.atom_id_range = (int(store.atom_flat.min()), int(store.atom_flat.max()))      # bounds every batch's range
        self.plan_seconds, self.plan_calls = 0.0, 0
        if self.device.type == "cuda":
            buf = torch.from_numpy(np.concatenate(host)).to(self.device)
            self.dev, o = [], 0
            for a in host:
                self.dev.append(buf[o:o + len(a)]); o += len(a)
            self._stage = [None] * self.N_STAGE
            self._events = [None] * self.N_STAGE
            self._k = 0
            # the batch's copy + emit kernel run on a stream of their own, so that batch i+1 is written while step i still
            # computes (the host runs ahead); the caller's stream picks up behind them.  BMP_COLLATE_STREAM=0: in line
            self.stream = torch.cuda.Stream(device=self.device) if os.environ.get("BMP_COLLATE_STREAM", "1") != "0" else None
            if self.stream is not None:
                self.stream.wait_stream(torch.cuda.current_stream(self.device))          # the store's upload above

    def _staging(self, n_ints: int) -> torch.Tensor:
        k = self._k
        self._k = (k + 1) % self.N_STAGE
        if self._events[k] is not None:
            self._events[k].synchronize()
        if self._stage[k] is None or self._stage[k].numel() < n_ints:
            self._stage[k] = torch.empty(max(n_ints, 1 << 15), dtype=torch.int32).pin_memory()
        self._cur = k
        return self._stage[k]

    def _staged(self) -> None:
        ev = self._events[self._cur]
        if ev is None:
            ev = self._events[self._cur] = torch.cuda.Event()
        ev.record()

    def _staged_unused(self) -> None:
        """The buffer handed out by ``_staging`` was not sent after all: it is free again at once."""
        self._k = self._cur


def pack_from_store_device(dstore: DeviceMolStore, sides: Sequence[np.ndarray], R: int = DEFAULT_R,
                           pad_to: Optional[Sequence[int]] = None, labels: Optional[np.ndarray] = None):
    """pack_from_store with the store resident on the GPU: the host only does the size arithmetic (zero-padding width
    per side, tile placement, entry bases: ``bmp_collate_plan``), one pinned H2D copy carries the plan (and the
    batch's labels and co-attention metadata), and one kernel writes the packed arrays (``bmp_collate_emit``).
    Bit-identical to ``pack_from_store`` (tests/test_gpu_collate.py).  Returns the batch, or (batch, labels on the
    device) when ``labels`` (int array, one row per pair) is given."""
    from . import _lib
    from ._lib import check, ptr, stream
    L = _lib.lib()
    dev = dstore.device
    if dev.type != "cuda":
        raise ValueError("pack_from_store_device needs the store on a GPU (no CPU fallback: use pack_from_store)")
    n_sides = len(sides)
    I = int(sum(len(s) for s in sides))
    B = len(sides[0])
    paired = n_sides == 2 and len(sides[1]) == B
    n_tab = 6 * I + (6 * I) % 2                       # keeps the int64 block 8-byte aligned
    n_meta = 8 * B if paired else 0
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32).reshape(-1)
    n_lab = 0 if lab is None else lab.size
    st = dstore._staging(n_tab + n_meta + n_lab)          # may wait for the copy that last used this buffer
    st_np = st.numpy()
    t_plan = _time.perf_counter()
    tab, side_tiles, side_mols, n_tiles, E, n_real, max_rows = collate_plan_host(
        dstore.st_nrows, dstore.st_nedges, sides, R, pad_to, tab=st_np[:6 * I])
    counts, ctotal = None, 0
    if paired:
        cnt = np.zeros(6, dtype=np.int32); ct = np.zeros(1, dtype=np.int64)
        check(L.bmp_collate_pair_meta(_i32p(st_np), I, B, side_tiles[1], R, _i32p(st_np[n_tab:]), _i32p(cnt), _i32p(ct)),
              "bmp_collate_pair_meta")
        counts, ctotal, np_big = [int(c) for c in cnt[:5]], int(ct[0]), int(cnt[5])
    if lab is not None:
        st_np[n_tab + n_meta:n_tab + n_meta + n_lab] = lab
    dstore.plan_seconds += _time.perf_counter() - t_plan          # the host's share of the collate (size arithmetic)
    dstore.plan_calls += 1
    n_up = n_tab + n_meta + n_lab
    N = n_tiles * R
    cur = torch.cuda.current_stream(dev)
    cs = dstore.stream if dstore.stream is not None else cur
    with torch.cuda.stream(cs):
        up = torch.empty(n_up, dtype=torch.int32, device=dev)
        up.copy_(st[:n_up], non_blocking=True)
        dstore._staged()
        ibuf = torch.empty(4 * N + 2 + 2 * E, dtype=torch.int32, device=dev)
        fbuf = torch.empty(N + 2 * E, dtype=torch.float32, device=dev)
    atom_id, csr_ptr, csr_col = ibuf[:N], ibuf[N:2 * N + 1], ibuf[2 * N + 1:2 * N + 1 + E]
    o = 2 * N + 1 + E
    csrT_ptr, csrT_col, row_mol = ibuf[o:o + N + 1], ibuf[o + N + 1:o + N + 1 + E], ibuf[o + N + 1 + E:o + 2 * N + 1 + E]
    row_w, csr_val, csrT_val = fbuf[:N], fbuf[N:N + E], fbuf[N + E:]
    assert row_mol.numel() == N and csrT_val.numel() == E and csrT_ptr.numel() == N + 1
    d = dstore.dev
    check(L.bmp_collate_emit(ptr(up), I, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), ptr(d[5]), ptr(d[6]),
                             ptr(atom_id), ptr(row_w), ptr(row_mol), ptr(csr_ptr), ptr(csr_col), ptr(csr_val), ptr(csrT_ptr),
                             ptr(csrT_col), ptr(csrT_val), None, _c_void_p(cs.cuda_stream)), "bmp_collate_emit")
    if cs is not cur:
        for t in (up, ibuf, fbuf):          # allocated on the collate stream, read by the caller's: no reuse before that work
            t.record_stream(cur)
        cur.wait_stream(cs)
    nrows_host = st_np[I:2 * I].astype(np.int64)
    pb = PackedMolBatch(
        R=R, n_tiles=n_tiles, n_mols=I, atom_id=atom_id, row_w=row_w, csr_ptr=csr_ptr, csr_col=csr_col, csr_val=csr_val,
        csrT_ptr=csrT_ptr, csrT_col=csrT_col, csrT_val=csrT_val, mol_row0=up[:I], mol_nrows=up[I:2 * I],
        side_tiles=side_tiles, side_mols=side_mols, n_real_atoms=n_real, n_edges=E, max_rows_per_mol=max_rows,
        mol_nrows_host=nrows_host, row_mol=row_mol, atom_id_range=dstore.atom_id_range)
    if paired:
        m = up[n_tab:n_tab + n_meta]
        # forward: the pairs of the classes 0..3 as ONE launch sized by the largest of them, the oversized pairs (class 4) apart
        counts_f = [0, 0, 0, 0, counts[4]]
        if B > counts[4]:
            counts_f[max(k for k in range(4) if counts[k])] = B - counts[4]
        pb._cache["pair_meta"] = dict(
            B=B, T1=side_tiles[1], T2=side_tiles[2] - side_tiles[1], coff=m[:2 * B].view(torch.int64), r1=m[2 * B:3 * B],
            n1=m[3 * B:4 * B], r2=m[4 * B:5 * B], n2=m[5 * B:6 * B], order=m[6 * B:7 * B], order_f=m[7 * B:8 * B],
            counts=counts, counts_f=counts_f, ctotal=ctotal, np_big=np_big)
    if lab is None:
        return pb
    return pb, up[n_tab + n_meta:].view(np.asarray(labels).shape if np.asarray(labels).ndim > 1 else (-1,))


# ---------------------------------------------------------------------------------------------------------
# A batch of a FIXED shape at FIXED device addresses (round 4): what a HIP graph of the training step needs.
# ---------------------------------------------------------------------------------------------------------
def plan_one_per_tile(st_nrows: np.ndarray, st_nedges: np.ndarray, sides: Sequence[np.ndarray], R: int, out: np.ndarray,
                      mt_out: np.ndarray):
    """The plan table of ``bmp_collate_plan`` (row0 | nrows | mid | ebase | padw | ndead, stride I; ``out``, int32, >= 6 I
    entries) for the placement "every molecule instance owns one R-row tile", and the tile table (``mt_out``, int32, 2 I
    entries: mt_row0 [I] | mt_nblk [I], the tile's live 32-row blocks -- they hold the real atoms and the pad row).  Host
    arithmetic on a few dozen integers; returns (I, n_edges, n_real_atoms)."""
    mids = np.concatenate([np.asarray(s) for s in sides]).astype(np.int64)
    I = len(mids)
    if mids.min() < 0 or mids.max() >= len(st_nrows):
        raise ValueError("molecule index outside the store")
    nrows = st_nrows[mids].astype(np.int64)
    if int(nrows.max()) > R:
        raise ValueError(f"a molecule of more than {R - 1} atoms does not fit the one-tile-per-molecule placement")
    ne = st_nedges[mids].astype(np.int64)
    tab = out[:6 * I].reshape(6, I)
    tab[0] = np.arange(I) * R
    tab[1] = nrows
    tab[2] = mids
    tab[3] = np.cumsum(ne) - ne
    lo = 0
    for s in sides:                     # concat_mols pads every side to its largest molecule (train_ddi_modify.py:296)
        hi = lo + len(s)
        tab[4, lo:hi] = int(nrows[lo:hi].max()) - nrows[lo:hi]
        lo = hi
    tab[5] = R - nrows
    mt_out[:I] = tab[0]
    mt_out[I:2 * I] = (nrows + 31) // 32
    return I, int(ne.sum()), int(nrows.sum()) - I


class StaticPairBatch:
    """A two-sided batch of ``B`` drug pairs whose arrays live at fixed device addresses and have a fixed shape, so that a
    training step on it can be recorded ONCE as a HIP graph (bmp.dp.GraphedTrainStep) and replayed for every batch of the
    epoch: the reference's default of 32 pairs (train_ddi_modify.py:196) is a chain of ~55 launches of one workgroup
    round each, and the framework's per-launch host work (1.2-1.6 ms per step, by the box) was what the 32-pair leg of
    bench.py measured.

    Placement: every molecule instance owns one R-row tile; the tile table names its live 32-row blocks, which is all the
    fused step kernels work on (a step lasts as long as the batch's tallest molecule, as in the encoder layout of
    bmp/enclayout.py) -- hence N = 2 B R rows, 2 B tiles on 2 B CUs and per-pair row ranges that depend on nothing but B.
    ``tile_stride = R`` tells those kernels to clear the rows of a tile's other blocks in what they write: the GEMMs that
    walk all rows (weight gradients, projections) would otherwise read what an earlier batch left there.  The pair kernels take every pair in
    their 128-row class (a class is an upper bound on the rows of its pairs).  Only the CONTENTS change from batch to batch:
    ``load`` plans on the host (numpy on 2 B integers + bmp_collate_pair_meta) into a pinned buffer and copies it over;
    ``emit`` (one launch of bmp_collate_emit, part of the recorded step) writes the packed arrays from the store in HBM."""

    N_STAGE = 4

    def __init__(self, dstore: DeviceMolStore, B: int, label_cols: int = 1, R: int = DEFAULT_R):
        if dstore.device.type != "cuda":
            raise ValueError("StaticPairBatch needs the store on a GPU")
        if int(dstore.st_nrows.max()) > R:
            raise ValueError(f"the store holds a molecule of more than {R - 1} atoms: no fixed-shape batch for it")
        self.dstore, self.B, self.R, self.label_cols = dstore, int(B), R, int(label_cols)
        dev = dstore.device
        I = 2 * self.B
        N = I * R
        e_cap = I * int(dstore.st_nedges.max())
        self.n_tab, self.n_meta, self.n_mt, self.n_lab = 6 * I, 8 * self.B, 2 * I, self.B * self.label_cols
        self.n_up = self.n_tab + self.n_meta + self.n_mt + self.n_lab
        self.up = torch.zeros(self.n_up, dtype=torch.int32, device=dev)
        ibuf = torch.zeros(4 * N + 2 + 2 * e_cap, dtype=torch.int32, device=dev)
        fbuf = torch.zeros(N + 2 * e_cap, dtype=torch.float32, device=dev)
        o = 2 * N + 1 + e_cap
        up, nt = self.up, self.n_tab
        m = up[nt:nt + self.n_meta]
        mt = up[nt + self.n_meta:nt + self.n_meta + self.n_mt]
        Bp = self.B
        self.pb = PackedMolBatch(
            R=R, n_tiles=I, n_mols=I, atom_id=ibuf[:N], row_w=fbuf[:N], csr_ptr=ibuf[N:2 * N + 1], csr_col=ibuf[2 * N + 1:o],
            csr_val=fbuf[N:N + e_cap], csrT_ptr=ibuf[o:o + N + 1], csrT_col=ibuf[o + N + 1:o + N + 1 + e_cap],
            csrT_val=fbuf[N + e_cap:], mol_row0=up[:I], mol_nrows=up[I:2 * I], side_tiles=(0, Bp, I), side_mols=(0, Bp, I),
            n_real_atoms=0, n_edges=e_cap, max_rows_per_mol=R, mol_nrows_host=None, row_mol=ibuf[o + N + 1 + e_cap:o + 2 * N + 1 + e_cap],
            atom_id_range=dstore.atom_id_range, mt_row0=mt[:I], mt_nblk=mt[I:], n_mtiles=I, tile_stride=R)
        # every pair in the 128-row class, whatever its size; C blocks at their largest
        counts = [0, 0, 0, Bp, 0]
        self.pair_meta = dict(
            B=Bp, T1=Bp, T2=Bp, coff=m[:2 * Bp].view(torch.int64), r1=m[2 * Bp:3 * Bp], n1=m[3 * Bp:4 * Bp], r2=m[4 * Bp:5 * Bp],
            n2=m[5 * Bp:6 * Bp], order=m[6 * Bp:7 * Bp], order_f=m[7 * Bp:8 * Bp], counts=counts, counts_f=list(counts),
            ctotal=Bp * (R * R + 4 * R), np_big=0)
        lab = up[nt + self.n_meta + self.n_mt:]
        self.t = lab.view(Bp, self.label_cols) if self.label_cols > 1 else lab.view(Bp, 1)
        self._stage = [torch.empty(self.n_up, dtype=torch.int32).pin_memory() for _ in range(self.N_STAGE)]
        self._events = [None] * self.N_STAGE
        self._k = 0
        self.n_real_atoms = self.n_edges = 0
        self.wait_seconds = 0.0            # time ``load`` spent waiting for the GPU to catch up (not host work)
        self.reset_derived()

    def reset_derived(self) -> None:
        """Forget what earlier steps derived from the batch's contents (row lists per bond type, rescaled adjacency): the
        arrays are the same tensors, their contents are not."""
        self.pb._cache = {"pair_meta": self.pair_meta}

    def load(self, sides: Sequence[np.ndarray], labels: np.ndarray) -> None:
        """Plan the batch (idx1, idx2) on the host and send the plan, the pair metadata, the tile table and the labels to their
        fixed place on the device (one pinned copy on the caller's stream).  ``emit`` then writes the packed arrays."""
        L = _lib_mod().lib()
        if len(sides) != 2 or len(sides[0]) != self.B or len(sides[1]) != self.B:
            raise ValueError(f"a StaticPairBatch of {self.B} pairs takes two sides of {self.B} molecules")
        lab = np.ascontiguousarray(labels, dtype=np.int32).reshape(-1)
        if lab.size != self.n_lab:
            raise ValueError(f"labels: expected {self.B} x {self.label_cols} entries")
        k = self._k
        self._k = (k + 1) % self.N_STAGE
        if self._events[k] is not None:
            tw = _time.perf_counter()
            self._events[k].synchronize()          # the copy that last used this pinned buffer: the host is N_STAGE steps ahead
            self.wait_seconds += _time.perf_counter() - tw
        st = self._stage[k].numpy()
        t0 = _time.perf_counter()
        o_mt = self.n_tab + self.n_meta
        I, self.n_edges, self.n_real_atoms = plan_one_per_tile(self.dstore.st_nrows, self.dstore.st_nedges, sides, self.R, st,
                                                               st[o_mt:o_mt + self.n_mt])
        cnt = np.zeros(6, dtype=np.int32); ct = np.zeros(1, dtype=np.int64)
        rc = L.bmp_collate_pair_meta(_i32p(st), I, self.B, self.B, self.R, _i32p(st[self.n_tab:]), _i32p(cnt), _i32p(ct))
        if rc:
            raise RuntimeError(f"bmp_collate_pair_meta failed ({rc})")
        st[self.n_tab + self.n_meta + self.n_mt:] = lab
        # (host copy of the row counts: what the operators outside the recorded path read -- the coarse co-attention family)
        self.pb.mol_nrows_host = st[I:2 * I].astype(np.int64)
        self.dstore.plan_seconds += _time.perf_counter() - t0
        self.dstore.plan_calls += 1
        self.up.copy_(self._stage[k], non_blocking=True)
        if self._events[k] is None:
            self._events[k] = torch.cuda.Event()
        self._events[k].record()

    def emit(self) -> None:
        """bmp_collate_emit on the caller's stream, into the fixed arrays (recorded with the step when a graph is captured)."""
        from ._lib import check, ptr, stream
        L = _lib_mod().lib()
        d, pb = self.dstore.dev, self.pb
        check(L.bmp_collate_emit(ptr(self.up), 2 * self.B, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), ptr(d[5]), ptr(d[6]),
                                 ptr(pb.atom_id), ptr(pb.row_w), ptr(pb.row_mol), ptr(pb.csr_ptr), ptr(pb.csr_col), ptr(pb.csr_val),
                                 ptr(pb.csrT_ptr), ptr(pb.csrT_col), ptr(pb.csrT_val), None, stream()), "bmp_collate_emit")


def _lib_mod():
    from . import _lib
    return _lib

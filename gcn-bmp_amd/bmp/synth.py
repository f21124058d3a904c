"""Synthetic drug store with the statistics of the binary DDI set.

The reference's datasets are external downloads (dataset/README.md:1) and RDKit
is absent, so the only available input is synthetic data that honours the batch
contract of the reference's data path: per molecule an
int32 atomic-number vector and a 4-channel symmetric 0/1 adjacency
(single, double, triple, aromatic; no self loops), as produced by
``CSVFileParserForPair.parse`` (parsers.py:156-335) with the stock GGNN
preprocessor (train_ddi_modify.py:256).

Cardinalities follow the reference: 544 drugs, all C(544,2)=147 696 unordered
pairs (setting.py:30-31, RECORD.txt:56-60), 32.32 % positives (RECORD.txt:58),
store seed 2018 (setting.py:28), pair seed 777 (train_ddi_modify.py:227).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

NUM_EDGE_TYPE = 4
_ATOM_Z = np.array([6, 7, 8, 9, 15, 16, 17], dtype=np.int32)          # C N O F P S Cl
_ATOM_P = np.array([0.72, 0.10, 0.12, 0.015, 0.005, 0.02, 0.02])
_BOND_P = np.array([0.55, 0.10, 0.01, 0.34])                           # single double triple aromatic


@dataclass
class Molecule:
    atoms: np.ndarray        # (n,) int32 atomic numbers, never 0
    bonds: np.ndarray        # (nb, 3) int32: i, j, type  (i != j, undirected, listed once)

    @property
    def n(self) -> int:
        return int(self.atoms.shape[0])

    def dense_adj(self, A: int | None = None) -> np.ndarray:
        """(4, A, A) float32, symmetric 0/1 (GGNN preprocessor contract)."""
        A = self.n if A is None else A
        adj = np.zeros((NUM_EDGE_TYPE, A, A), dtype=np.float32)
        if len(self.bonds):
            i, j, t = self.bonds[:, 0], self.bonds[:, 1], self.bonds[:, 2]
            adj[t, i, j] = 1.0
            adj[t, j, i] = 1.0
        return adj


def _make_molecule(rs: np.random.RandomState, n_lo: int, n_hi: int, n_mean: float) -> Molecule:
    n = int(np.clip(np.round(np.exp(rs.normal(np.log(n_mean), 0.45))), n_lo, n_hi))
    atoms = _ATOM_Z[rs.choice(len(_ATOM_Z), size=n, p=_ATOM_P)].astype(np.int32)
    deg = np.zeros(n, dtype=np.int64)
    nbr = [set() for _ in range(n)]
    bonds: List[Tuple[int, int, int]] = []

    def add(i: int, j: int) -> None:
        t = int(rs.choice(NUM_EDGE_TYPE, p=_BOND_P))
        bonds.append((i, j, t))
        deg[i] += 1
        deg[j] += 1
        nbr[i].add(j)
        nbr[j].add(i)

    for k in range(1, n):                       # random tree, max degree 4
        cand = np.nonzero(deg[:k] < 4)[0]
        add(int(cand[rs.randint(len(cand))]), k)
    for _ in range(int(round(0.12 * n))):       # ring closures between non-adjacent atoms
        for _try in range(16):
            i, j = int(rs.randint(n)), int(rs.randint(n))
            if i != j and deg[i] < 4 and deg[j] < 4 and j not in nbr[i]:
                add(i, j)
                break
    return Molecule(atoms, np.asarray(bonds, dtype=np.int32).reshape(-1, 3))


def make_store(n_mols: int = 544, seed: int = 2018, n_lo: int = 4, n_hi: int = 96,
               n_mean: float = 24.0) -> List[Molecule]:
    rs = np.random.RandomState(seed)
    return [_make_molecule(rs, n_lo, n_hi, n_mean) for _ in range(n_mols)]


def make_pairs(n_mols: int = 544, seed: int = 777, pos_rate: float = 0.3232,
               limit: int | None = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """All unordered pairs in a seeded permutation + Bernoulli labels.
    Returns (idx1, idx2, label) int32 arrays."""
    rs = np.random.RandomState(seed)
    iu, ju = np.triu_indices(n_mols, k=1)
    perm = rs.permutation(len(iu))
    label = (rs.uniform(size=len(iu)) < pos_rate).astype(np.int32)
    if limit is not None:
        perm = perm[:limit]
        label = label[:limit]
    return iu[perm].astype(np.int32), ju[perm].astype(np.int32), label


def make_multilabel_pairs(n_mols: int = 1704, n_pairs: int = 192000, n_class: int = 37, seed: int = 777
                          ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Multi-label store pairs (setting.py:33; train_ggnn_hole_multi_class_x37.py:274-310):
    one class ~ Zipf(1.2) over n_class, 1.5 % of pairs get a second class."""
    rs = np.random.RandomState(seed)
    iu, ju = np.triu_indices(n_mols, k=1)
    sel = rs.choice(len(iu), size=n_pairs, replace=False)
    w = 1.0 / np.arange(1, n_class + 1) ** 1.2
    w /= w.sum()
    lab = np.zeros((n_pairs, n_class), dtype=np.int32)
    lab[np.arange(n_pairs), rs.choice(n_class, size=n_pairs, p=w)] = 1
    second = rs.uniform(size=n_pairs) < 0.015
    lab[second, rs.choice(n_class, size=int(second.sum()), p=w)] = 1
    return iu[sel].astype(np.int32), ju[sel].astype(np.int32), lab


def concat_mols(mols: List[Molecule]) -> Tuple[np.ndarray, np.ndarray]:
    """chainer_chemistry ``concat_mols`` semantics (train_ddi_modify.py:296): zero-pad
    every field to the batch max shape and stack.  Returns atoms (B, A) int32,
    adj (B, 4, A, A) float32."""
    A = max(m.n for m in mols)
    atoms = np.zeros((len(mols), A), dtype=np.int32)
    adj = np.zeros((len(mols), NUM_EDGE_TYPE, A, A), dtype=np.float32)
    for b, m in enumerate(mols):
        atoms[b, :m.n] = m.atoms
        adj[b, :, :m.n, :m.n] = m.dense_adj()
    return atoms, adj

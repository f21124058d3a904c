"""SMILES -> (atomic numbers, 4-channel adjacency): the step in front of the hot path.

The reference gets these arrays from RDKit (``Chem.MolFromSmiles``, parsers.py:209-210, no canonicalisation:
parsers.py:229-235) and chainer-chemistry's GGNN preprocessor (atomic-number vector in atom order; adjacency
channels SINGLE, DOUBLE, TRIPLE, AROMATIC; hydrogens not added).  Neither library is available here, so this is a
small stand-alone reader of the SMILES grammar that keeps the atom order of the string:

* organic-subset atoms, bracket atoms (isotope, chirality, H count, charge and class are read and dropped --
  the model only sees atomic numbers), two-letter elements, aromatic lower-case atoms;
* bonds ``- = # :`` and the stereo marks ``/ \\`` (single), branches, ring closures (digits, ``%nn``, with an optional
  bond symbol on either end), ``.`` for disconnected parts;
* an unmarked bond between two aromatic atoms is AROMATIC if it lies in a ring and SINGLE otherwise (biphenyl);
* explicit hydrogens ``[H]`` attached to one heavy atom are dropped, as RDKit's default sanitisation does.

* Kekule input is aromatised the way RDKit's sanitisation would in the common cases (``C1=CC=CC=C1`` -> six AROMATIC
  bonds): ``perceive_aromaticity`` counts pi electrons per ring of the smallest-ring set (sizes 5-7) -- 1 for an atom with
  a double bond to another ring atom of the molecule's ring systems, 0 for a carbon whose double bond leaves the rings
  towards O / S / N (pyridone, caffeine), 2 for N / O / S / Se / P-H without a double bond (pyrrole, furan, thiophene),
  sp3 atoms disqualify the ring -- and marks the ring's bonds AROMATIC when the count is 4n + 2.

What it does NOT do: RDKit's full aromaticity model (envelopes of fused systems such as azulene, charged rings,
exotic elements), valence checks, ``$`` bonds and ``*`` atoms (rejected: atom id 0 is the padding id).
PARITY UNPINNED: without RDKit nothing here can be checked against the reference's own output; tests/test_smiles.py
holds hand-derived cases only.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .synth import Molecule

SINGLE, DOUBLE, TRIPLE, AROMATIC = 0, 1, 2, 3

_ELEMENTS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr "
             "Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir "
             "Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl "
             "Mc Lv").split()
ATOMIC_NUMBER: Dict[str, int] = {s: i + 1 for i, s in enumerate(_ELEMENTS)}
_ORGANIC = ("Cl", "Br", "B", "C", "N", "O", "P", "S", "F", "I")
_AROMATIC = {"b": 5, "c": 6, "n": 7, "o": 8, "p": 15, "s": 16, "se": 34, "as": 33, "te": 52}
_BOND_SYMBOL = {"-": SINGLE, "/": SINGLE, "\\": SINGLE, "=": DOUBLE, "#": TRIPLE, ":": AROMATIC}


class SmilesError(ValueError):
    pass


def _bracket_atom(body: str) -> Tuple[int, bool]:
    """'[13CH3+]' body -> (atomic number, aromatic).  Isotope, chirality, H count, charge, class are skipped."""
    i = 0
    while i < len(body) and body[i].isdigit():
        i += 1
    rest = body[i:]
    for sym in sorted(_AROMATIC, key=len, reverse=True):
        if rest.startswith(sym) and not (len(rest) > len(sym) and rest[len(sym)].islower()):
            return _AROMATIC[sym], True
    for n in (2, 1):
        sym = rest[:n]
        if sym in ATOMIC_NUMBER and sym[0].isupper():
            return ATOMIC_NUMBER[sym], False
    raise SmilesError(f"unknown element in bracket atom [{body}]")


def parse_smiles(smiles: str) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (atoms (n,) int32 atomic numbers, bonds (nb, 3) int32 rows i, j, type) in the atom order of the string."""
    atoms: List[int] = []
    arom: List[bool] = []
    bonds: List[List[int]] = []                    # i, j, type or -1 (unmarked, decided at the end)
    stack: List[int] = []
    rings: Dict[int, Tuple[int, int]] = {}         # ring number -> (atom, bond type or -1)
    prev: Optional[int] = None
    pending = -1
    i, n = 0, len(smiles)

    def add_atom(z: int, aromatic: bool) -> None:
        nonlocal prev, pending
        atoms.append(z); arom.append(aromatic)
        k = len(atoms) - 1
        if prev is not None:
            bonds.append([prev, k, pending])
        prev, pending = k, -1

    while i < n:
        c = smiles[i]
        if c == "[":
            j = smiles.find("]", i)
            if j < 0:
                raise SmilesError("unclosed '['")
            z, ar = _bracket_atom(smiles[i + 1:j])
            add_atom(z, ar)
            i = j + 1
        elif c in _BOND_SYMBOL:
            pending = _BOND_SYMBOL[c]
            i += 1
        elif c == "$":
            raise SmilesError("quadruple bonds are not supported")
        elif c == "(":
            if prev is None:
                raise SmilesError("branch before any atom")
            stack.append(prev); i += 1
        elif c == ")":
            if not stack:
                raise SmilesError("unbalanced ')'")
            prev = stack.pop(); i += 1
        elif c == ".":
            prev, pending = None, -1
            i += 1
        elif c.isdigit() or c == "%":
            if c == "%":
                if not smiles[i + 1:i + 3].isdigit() or len(smiles[i + 1:i + 3]) < 2:
                    raise SmilesError("'%' needs two digits")
                num = int(smiles[i + 1:i + 3]); i += 3
            else:
                num = int(c); i += 1
            if prev is None:
                raise SmilesError("ring closure before any atom")
            if num in rings:
                other, btype = rings.pop(num)
                if btype >= 0 and pending >= 0 and btype != pending:
                    raise SmilesError(f"ring closure {num}: conflicting bond symbols")
                if other == prev:
                    raise SmilesError(f"ring closure {num} on one atom")
                bonds.append([other, prev, btype if btype >= 0 else pending])
            else:
                rings[num] = (prev, pending)
            pending = -1
        elif c == "*":
            raise SmilesError("wildcard atoms are not supported (atom id 0 is the padding id)")
        else:
            two = smiles[i:i + 2]
            if two in _ORGANIC:
                add_atom(ATOMIC_NUMBER[two], False); i += 2
            elif c in _ORGANIC:
                add_atom(ATOMIC_NUMBER[c], False); i += 1
            elif c in _AROMATIC:
                add_atom(_AROMATIC[c], True); i += 1
            else:
                raise SmilesError(f"unexpected character {c!r} at {i}")
    if rings:
        raise SmilesError(f"unclosed ring bond(s) {sorted(rings)}")
    if stack:
        raise SmilesError("unbalanced '('")
    if not atoms:
        raise SmilesError("empty SMILES")

    # unmarked bonds: aromatic iff both ends aromatic and the bond is in a ring (not a bridge)
    bridge = _bridges(len(atoms), bonds)
    for k, b in enumerate(bonds):
        if b[2] < 0:
            b[2] = AROMATIC if (arom[b[0]] and arom[b[1]] and not bridge[k]) else SINGLE
    seen = set()
    for a, b, _t in bonds:
        key = (min(a, b), max(a, b))
        if key in seen:
            raise SmilesError("two bonds between the same pair of atoms")
        seen.add(key)
    atoms_a, bonds_a = _drop_hydrogens(np.asarray(atoms, np.int32), np.asarray(bonds, np.int32).reshape(-1, 3))
    return atoms_a, perceive_aromaticity(atoms_a, bonds_a)


_LONE_PAIR = {7, 8, 15, 16, 34}            # N O P S Se: contribute 2 pi electrons when they carry no double bond
_AROMATIC_Z = {5, 6, 7, 8, 15, 16, 33, 34}


def _smallest_rings(n: int, bonds: np.ndarray, max_size: int = 7) -> List[List[int]]:
    """For every ring bond the smallest ring through it (BFS from one end to the other without the bond); the distinct
    rings of size <= max_size.  For the fused ring systems of drug-like molecules this is the set RDKit's SSSR gives."""
    adj: List[List[int]] = [[] for _ in range(n)]
    for a, b, _t in bonds:
        adj[a].append(int(b)); adj[b].append(int(a))
    rings, seen = [], set()
    for a, b, _t in bonds:
        a, b = int(a), int(b)
        prev = {a: -1}
        frontier = [a]
        found = False
        while frontier and not found:
            nxt = []
            for v in frontier:
                for w in adj[v]:
                    if (v == a and w == b) or w in prev:
                        continue
                    prev[w] = v
                    if w == b:
                        found = True
                        break
                    nxt.append(w)
                if found:
                    break
            frontier = nxt
        if not found:
            continue
        path = [b]
        while path[-1] != a:
            path.append(prev[path[-1]])
        if len(path) > max_size:
            continue
        key = frozenset(path)
        if key not in seen:
            seen.add(key)
            rings.append(path)
    return rings


def perceive_aromaticity(atoms: np.ndarray, bonds: np.ndarray) -> np.ndarray:
    """Kekule -> aromatic bond types by pi-electron counting per ring (see the module docstring).  Bonds that are already
    AROMATIC (aromatic SMILES) stay; rings mixing aromatic and Kekule notation are left alone."""
    if len(bonds) == 0:
        return bonds
    n = len(atoms)
    bonds = bonds.copy()
    rings = [r for r in _smallest_rings(n, bonds) if 5 <= len(r) <= 7]
    if not rings:
        return bonds
    in_ring = np.zeros(n, dtype=bool)
    for r in rings:
        in_ring[r] = True
    btype: Dict[Tuple[int, int], int] = {}
    nbrs: List[List[Tuple[int, int]]] = [[] for _ in range(n)]
    for k, (a, b, t) in enumerate(bonds.tolist()):
        btype[(min(a, b), max(a, b))] = k
        nbrs[a].append((b, t)); nbrs[b].append((a, t))
    bridge = _bridges(n, bonds.tolist())
    ring_bond = {(min(a, b), max(a, b)) for k, (a, b, _t) in enumerate(bonds.tolist()) if not bridge[k]}

    def electrons(v: int) -> int:
        """pi electrons atom v brings to a ring it is part of; -1: the atom cannot be aromatic."""
        z = int(atoms[v])
        if z not in _AROMATIC_Z:
            return -1
        doubles = [(w, t) for w, t in nbrs[v] if t == DOUBLE]
        if any(t == TRIPLE for _w, t in nbrs[v]) or len(doubles) > 1:
            return -1
        if doubles:
            w = doubles[0][0]
            if (min(v, w), max(v, w)) in ring_bond:
                return 1                                   # a ring double bond (this ring's or a fused neighbour's)
            if z == 6 and int(atoms[w]) in (7, 8, 16):
                return 0                                   # exocyclic C=O / C=S / C=N: sp2 carbon with an empty p orbital
            return -1                                      # exocyclic C=C (fulvene, an ylidene link between two rings): the
                                                           # bond itself lies in no ring, whatever its far end belongs to
        if any(t == AROMATIC for _w, t in nbrs[v]):
            return -1                                      # already aromatic notation: leave such rings alone
        if z in _LONE_PAIR and len(nbrs[v]) <= 3:
            return 2
        return -1                                          # sp3 carbon (cyclopentadiene's CH2, cyclohexene's CH2)

    for r in rings:
        es = [electrons(v) for v in r]
        if min(es) < 0 or (sum(es) - 2) % 4 != 0:
            continue
        ring_pairs = [(r[k], r[(k + 1) % len(r)]) for k in range(len(r))]
        if any(bonds[btype[(min(a, b), max(a, b))], 2] == TRIPLE for a, b in ring_pairs):
            continue
        for a, b in ring_pairs:
            bonds[btype[(min(a, b), max(a, b))], 2] = AROMATIC
    return bonds


def _bridges(n: int, bonds: Sequence[Sequence[int]]) -> List[bool]:
    """bridge[k] is True iff bond k lies in no ring (iterative low-link DFS)."""
    adj: List[List[Tuple[int, int]]] = [[] for _ in range(n)]
    for k, (a, b, _t) in enumerate(bonds):
        adj[a].append((b, k)); adj[b].append((a, k))
    disc, low = [-1] * n, [0] * n
    bridge = [False] * len(bonds)
    t = 0
    for root in range(n):
        if disc[root] >= 0:
            continue
        disc[root] = low[root] = t; t += 1
        st = [(root, -1, 0)]
        while st:
            v, pe, idx = st.pop()
            if idx < len(adj[v]):
                st.append((v, pe, idx + 1))
                w, e = adj[v][idx]
                if e == pe:
                    continue
                if disc[w] < 0:
                    disc[w] = low[w] = t; t += 1
                    st.append((w, e, 0))
                else:
                    low[v] = min(low[v], disc[w])
            elif pe >= 0:
                a, b, _t = bonds[pe]
                parent = a if b == v else b
                low[parent] = min(low[parent], low[v])
                if low[v] > disc[parent]:
                    bridge[pe] = True
    return bridge


def _drop_hydrogens(atoms: np.ndarray, bonds: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Explicit H atoms with exactly one neighbour, and that neighbour heavy, disappear (RDKit RemoveHs default)."""
    deg = np.zeros(len(atoms), dtype=np.int64)
    for a, b in bonds[:, :2]:
        deg[a] += 1; deg[b] += 1
    drop = np.zeros(len(atoms), dtype=bool)
    for a, b in bonds[:, :2]:
        if atoms[a] == 1 and deg[a] == 1 and atoms[b] != 1:
            drop[a] = True
        if atoms[b] == 1 and deg[b] == 1 and atoms[a] != 1:
            drop[b] = True
    if not drop.any():
        return atoms, bonds
    new_id = np.cumsum(~drop) - 1
    keep = ~(drop[bonds[:, 0]] | drop[bonds[:, 1]])
    nb = bonds[keep].copy()
    nb[:, 0] = new_id[nb[:, 0]]; nb[:, 1] = new_id[nb[:, 1]]
    return atoms[~drop], nb


def smiles_to_molecule(smiles: str, max_atoms: int = -1) -> Molecule:
    """The GGNN preprocessor's output for one SMILES as a bmp.synth.Molecule (``dense_adj()`` gives the (4, n, n)
    array of the reference).  ``max_atoms`` >= 0 rejects larger molecules like the preprocessor's type check."""
    atoms, bonds = parse_smiles(smiles)
    if 0 <= max_atoms < len(atoms):
        raise SmilesError(f"{len(atoms)} atoms > max_atoms = {max_atoms}")
    return Molecule(atoms=atoms, bonds=bonds)


def parse_pair_csv(path: str, smiles_cols: Sequence[str] = ("smiles_1", "smiles_2"), labels: Optional[Sequence[str]] = None,
                   max_atoms: int = -1):
    """CSVFileParserForPair.parse (parsers.py:156-335) into the form the packed path wants: a store of the distinct
    molecules plus index pairs.  Rows whose SMILES cannot be read are skipped and counted, as the reference does.
    Returns dict(store=[Molecule], idx1, idx2, labels (n, len(labels)) int32 or None, smiles=[...], n_failed)."""
    import csv
    store: List[Molecule] = []
    index: Dict[str, int] = {}
    i1: List[int] = []
    i2: List[int] = []
    labs: List[List[int]] = []
    failed = 0

    def mol_id(s: str) -> int:
        if s not in index:
            store.append(smiles_to_molecule(s, max_atoms))
            index[s] = len(store) - 1
        return index[s]

    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            try:
                a, b = mol_id(row[smiles_cols[0]]), mol_id(row[smiles_cols[1]])
            except SmilesError:
                failed += 1
                continue
            i1.append(a); i2.append(b)
            if labels is not None:
                labs.append([int(row[c]) for c in labels])
    return dict(store=store, idx1=np.asarray(i1, np.int32), idx2=np.asarray(i2, np.int32),
                labels=np.asarray(labs, np.int32) if labels is not None else None,
                smiles=sorted(index, key=index.get), n_failed=failed)

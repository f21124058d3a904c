"""CPU tests of the packed batch layout (integer-exact) and of the packed formulation
(virtual pad rows + multiplicities) against the dense oracle in float64."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from bmp import synth, packed
import packed_ref as PR

T = torch.from_numpy


@pytest.fixture(scope="module")
def store():
    return synth.make_store(60, seed=5, n_lo=2, n_hi=24, n_mean=9)


def test_store_statistics():
    st = synth.make_store()
    n = np.array([m.n for m in st])
    assert len(st) == 544 and n.min() >= 4 and n.max() <= 96
    assert 24 < n.mean() < 29
    for m in st[:50]:
        adj = m.dense_adj()
        assert np.array_equal(adj, adj.transpose(0, 2, 1))          # symmetric
        assert adj.sum(axis=0).max() <= 1                            # one bond type per atom pair
        assert np.trace(adj.sum(axis=0)) == 0                        # no self loops
        assert adj.sum(axis=(0, 2)).max() <= 4                       # valence cap
    i1, i2, lab = synth.make_pairs()
    assert len(i1) == 147696 and (i1 < i2).all() and abs(lab.mean() - 0.3232) < 0.01


@pytest.mark.parametrize("R", [32, 64, 128])
def test_pack_store_roundtrip_bit_exact(store, R):
    ms = packed.MolStore(store)
    rs = np.random.RandomState(1)
    i1, i2 = rs.randint(0, len(store), 37), rs.randint(0, len(store), 37)
    pb = packed.pack_from_store(ms, [i1, i2], R=R, with_dense_map=True)
    assert pb.n_rows == pb.n_tiles * R and pb.side_mols == (0, 37, 74)
    for s, idx in enumerate((i1, i2)):
        atoms, adj = synth.concat_mols([store[k] for k in idx])
        a2, adj2 = packed.unpack_to_dense(pb, s)
        assert np.array_equal(atoms, a2) and np.array_equal(adj, adj2)
    # molecules never straddle a tile; sides live in disjoint tile ranges
    r0 = pb.mol_row0.numpy(); nr = pb.mol_nrows.numpy()
    assert ((r0 // R) == ((r0 + nr - 1) // R)).all()
    t = r0 // R
    assert t[:37].max() < pb.side_tiles[1] <= t[37:].min()
    # every bond is tile-local
    ptr = pb.csr_ptr.numpy(); col = pb.csr_col.numpy()
    dst = np.repeat(np.arange(pb.n_rows), np.diff(ptr))
    assert ((col >> 2) // R == dst // R).all()
    # multiplicities: sum of row_w per molecule == side's padded atom count
    w = pb.row_w.numpy()
    for s, idx in enumerate((i1, i2)):
        A = max(store[k].n for k in idx)
        for b in range(37):
            m = s * 37 + b
            assert w[r0[m]:r0[m] + nr[m]].sum() == A


def test_pack_dense_equals_pack_store(store):
    ms = packed.MolStore(store)
    idx = np.arange(20)
    atoms, adj = synth.concat_mols([store[k] for k in idx])
    pa = packed.pack_from_store(ms, [idx], R=64, with_dense_map=True)
    pd = packed.pack_from_dense([atoms], [adj], R=64)
    for f in ("atom_id", "row_w", "csr_ptr", "csr_col", "csr_val", "csrT_ptr", "csrT_col", "mol_row0", "mol_nrows"):
        assert torch.equal(getattr(pa, f), getattr(pd, f)), f
    assert torch.equal(pa.dense_map, pd.dense_map)


def test_pack_dense_asymmetric_weighted_and_isolated():
    """Arbitrary dense input: asymmetric float adjacency, an isolated REAL atom (id != 0, no
    bonds: must stay its own row), an id-0 position WITH an incoming bond (not pad-like)."""
    atoms = np.array([[6, 0, 11, 0, 0]], np.int32)
    adj = np.zeros((1, 4, 5, 5), np.float32)
    adj[0, 2, 0, 1] = 0.5          # row 0 receives from position 1 (id 0, but it has an in-edge below)
    adj[0, 1, 1, 0] = 2.0          # position 1 has an incoming bond -> real row
    adj[0, 3, 0, 3] = 0.25         # position 3 is pad-like (id 0, no in-edge) but is a SOURCE
    pb = packed.pack_from_dense([atoms], [adj], R=32)
    a2, adj2 = packed.unpack_to_dense(pb, 0)
    assert np.array_equal(atoms, a2)
    # position 4 is pad-like too and shares the virtual row with 3: the edge from 3 is
    # reproduced at the virtual row's first position (3)
    assert np.array_equal(adj, adj2)
    assert int(pb.mol_nrows[0]) == 4                     # 3 real rows (0,1,2) + virtual
    assert float(pb.row_w[int(pb.mol_row0[0]) + 3]) == 2.0
    # transpose CSR is the exact transpose
    ptr, col, val = pb.csr_ptr.numpy(), pb.csr_col.numpy(), pb.csr_val.numpy()
    ptrT, colT, valT = pb.csrT_ptr.numpy(), pb.csrT_col.numpy(), pb.csrT_val.numpy()
    E = {(d, c >> 2, c & 3, v) for d in range(pb.n_rows) for c, v in zip(col[ptr[d]:ptr[d + 1]], val[ptr[d]:ptr[d + 1]])}
    ET = {(c >> 2, s, c & 3, v) for s in range(pb.n_rows) for c, v in zip(colT[ptrT[s]:ptrT[s + 1]], valT[ptrT[s]:ptrT[s + 1]])}
    assert E == ET and len(E) == 3


def test_empty_and_single():
    ms = packed.MolStore([synth.Molecule(np.array([11], np.int32), np.zeros((0, 3), np.int32))])
    pb = packed.pack_from_store(ms, [np.array([0])], R=32, with_dense_map=True)
    assert pb.n_tiles == 1 and pb.n_edges == 0 and int(pb.csr_ptr[-1]) == 0
    # a molecule larger than a tile (the reference's preprocessor has no size limit, train_ddi_modify.py:256): whole
    # consecutive tiles of its own at the head of the side, the smaller ones behind it
    big = synth.Molecule(np.full(40, 6, np.int32), np.zeros((0, 3), np.int32))
    pbb = packed.pack_from_store(packed.MolStore([ms_mol := synth.Molecule(np.array([11], np.int32), np.zeros((0, 3), np.int32)), big]),
                                 [np.array([0, 1, 0])], R=32, with_dense_map=True)
    assert pbb.oversized and pbb.max_rows_per_mol == 41 and pbb.n_tiles == 3
    assert pbb.mol_row0.tolist() == [64, 0, 66] and pbb.mol_nrows.tolist() == [2, 41, 2]
    assert float(pbb.row_w.sum()) == 3 * 40 and int((pbb.row_mol >= 0).sum()) == 45


def test_bin_pack_fill():
    st = synth.make_store()
    ms = packed.MolStore(st)
    i1, i2, _ = synth.make_pairs(limit=1024)
    pb = packed.pack_from_store(ms, [i1, i2], R=128)
    used = int(pb.mol_nrows.sum())
    assert used / pb.n_rows > 0.97


# ------------------------------------------------- packed formulation == dense oracle (float64)
def _setup(store, nl=3, attn="nie"):
    ms = packed.MolStore(store)
    rs = np.random.RandomState(3)
    i1, i2 = rs.randint(0, len(store), 7), rs.randint(0, len(store), 7)
    p = O.make_pair_params(777, hidden_dim=8, out_dim=8, n_layers=nl, attn=attn, dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    d1 = synth.concat_mols([store[k] for k in i1]); d2 = synth.concat_mols([store[k] for k in i2])
    pb = packed.pack_from_store(ms, [i1, i2], R=32, with_dense_map=True)
    return p, pb, d1, d2


def test_packed_ggnn_matches_dense_oracle_with_grads(store):
    p, pb, (a1, j1), (a2, j2) = _setup(store)
    g, h = PR.ggnn_forward(p, pb, 3, prefix="graph_conv/")
    ge1, at1 = O.ggnn_forward(p, T(a1), T(j1).double(), 3, prefix="graph_conv/")
    ge2, at2 = O.ggnn_forward(p, T(a2), T(j2).double(), 3, prefix="graph_conv/")
    assert torch.allclose(g[:7], ge1, atol=1e-12) and torch.allclose(g[7:], ge2, atol=1e-12)
    assert torch.allclose(pb.to_dense(h, 0), at1, atol=1e-13)
    names = [n for n in sorted(p) if n.startswith("graph_conv/")]
    c = torch.randn(14, 8, dtype=torch.float64)
    gp = torch.autograd.grad((g * c).sum(), [p[n] for n in names])
    gd = torch.autograd.grad((torch.cat((ge1, ge2)) * c).sum(), [p[n] for n in names])
    for n, x, y in zip(names, gp, gd):
        assert torch.allclose(x, y, rtol=1e-9, atol=1e-11), n


def test_packed_nie_matches_dense_oracle_with_grads(store):
    p, pb, (a1, j1), (a2, j2) = _setup(store, nl=2)
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn="nie")
    _, h = PR.ggnn_forward(p, pb, 2, prefix="graph_conv/")
    c1, c2 = PR.nie_coattention(p, pb, h, np.arange(7), 7 + np.arange(7), prefix="attn/")
    assert torch.allclose(c1, g1, atol=1e-13) and torch.allclose(c2, g2, atol=1e-13)
    names = [n for n in sorted(p) if not n.startswith("mlp/")]
    w1 = torch.randn(7, 8, dtype=torch.float64); w2 = torch.randn(7, 8, dtype=torch.float64)
    gp = torch.autograd.grad((c1 * w1).sum() + (c2 * w2).sum(), [p[n] for n in names], allow_unused=True)
    gd = torch.autograd.grad((g1 * w1).sum() + (g2 * w2).sum(), [p[n] for n in names], allow_unused=True)
    for n, x, y_ in zip(names, gp, gd):
        assert (x is None) == (y_ is None), n        # the fine family ignores g_1/g_2 (readout unused)
        if x is not None:
            assert torch.allclose(x, y_, rtol=1e-9, atol=1e-11), n


@pytest.mark.parametrize("n_lt", [1, 2, 3])
def test_packed_deep_nie_matches_dense_oracle(store, n_lt):
    """Deep / VeryDeep / ExtremeDeep NieFineCoattention (nie_coattention.py:13-309): the packed restatement (pad
    positions as one weighted row) against the dense one on the padded atom states of a real encoder."""
    p, pb, (a1, j1), (a2, j2) = _setup(store, nl=2)
    dr = O._Draw(777, torch.float64, 0.2)
    O.init_nie(dr, "deep/", 8, 8, 8, n_lt=n_lt)
    p.update(dr.p)
    _, at1 = O.ggnn_forward(p, T(a1), T(j1).double(), 2, prefix="graph_conv/")
    _, at2 = O.ggnn_forward(p, T(a2), T(j2).double(), 2, prefix="graph_conv/")
    g1, g2 = O.nie_coattention(p, at1, at2, prefix="deep/", n_lt=n_lt)
    _, h = PR.ggnn_forward(p, pb, 2, prefix="graph_conv/")
    c1, c2 = PR.nie_coattention(p, pb, h, np.arange(7), 7 + np.arange(7), prefix="deep/", n_lt=n_lt)
    assert torch.allclose(c1, g1, atol=1e-12) and torch.allclose(c2, g2, atol=1e-12)


def test_packed_fourier_nie_matches_dense_oracle(store):
    """FourierFineCoattention (nie_coattention.py:399-513)."""
    p, pb, (a1, j1), (a2, j2) = _setup(store, nl=2)
    _, at1 = O.ggnn_forward(p, T(a1), T(j1).double(), 2, prefix="graph_conv/")
    _, at2 = O.ggnn_forward(p, T(a2), T(j2).double(), 2, prefix="graph_conv/")
    g1, g2 = O.nie_coattention(p, at1, at2, prefix="attn/", fourier=True)
    _, h = PR.ggnn_forward(p, pb, 2, prefix="graph_conv/")
    c1, c2 = PR.nie_coattention(p, pb, h, np.arange(7), 7 + np.arange(7), prefix="attn/", fourier=True)
    assert torch.allclose(c1, g1, atol=1e-12) and torch.allclose(c2, g2, atol=1e-12)
    n1, n2 = O.nie_coattention(p, at1, at2, prefix="attn/")
    assert not torch.allclose(g1, n1, atol=1e-3)                   # and it is not the plain Nie energy


def test_oversized_molecules_match_dense_oracle_with_grads():
    """Molecules larger than a tile (here R = 32: 33..80 atoms; SURVEY.md 8(a) R0, train_ddi_modify.py:256 sets no limit):
    the packed restatement -- gathers over absolute rows, per-molecule segment sums, pair blocks of any size -- against
    the dense float64 oracle: molecule vectors, atom states, co-attention outputs and every gradient."""
    store = synth.make_store(6, seed=21, n_lo=2, n_hi=20, n_mean=9) + synth.make_store(3, seed=22, n_lo=33, n_hi=80, n_mean=50)
    assert max(m.n for m in store) > 32
    ms = packed.MolStore(store)
    i1 = np.array([0, 6, 7, 3, 8, 1, 6]); i2 = np.array([7, 2, 6, 8, 8, 4, 0])
    p = O.make_pair_params(777, hidden_dim=8, out_dim=8, n_layers=2, attn="nie", dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    pb = packed.pack_from_store(ms, [i1, i2], R=32, with_dense_map=True)
    assert pb.oversized and pb.n_tiles > 10
    # dense <-> packed round trip of the index work stays exact
    for side, (a, j) in enumerate(((a1, j1), (a2, j2))):
        ad, jd = packed.unpack_to_dense(pb, side)
        assert np.array_equal(ad, a) and np.array_equal(jd, j)
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn="nie")
    ge1, at1 = O.ggnn_forward(p, T(a1), T(j1).double(), 2, prefix="graph_conv/")
    g, h = PR.ggnn_forward(p, pb, 2, prefix="graph_conv/")
    assert torch.allclose(g[:7], ge1, atol=1e-11) and torch.allclose(pb.to_dense(h, 0), at1, atol=1e-12)
    c1, c2 = PR.nie_coattention(p, pb, h, np.arange(7), 7 + np.arange(7), prefix="attn/")
    assert torch.allclose(c1, g1, atol=1e-12) and torch.allclose(c2, g2, atol=1e-12)
    names = [n for n in sorted(p) if not n.startswith("mlp/")]
    w1 = torch.randn(7, 8, dtype=torch.float64); w2 = torch.randn(7, 8, dtype=torch.float64)
    gp = torch.autograd.grad((c1 * w1).sum() + (c2 * w2).sum(), [p[n] for n in names], allow_unused=True)
    gd = torch.autograd.grad((g1 * w1).sum() + (g2 * w2).sum(), [p[n] for n in names], allow_unused=True)
    for n, x, y_ in zip(names, gp, gd):
        assert (x is None) == (y_ is None), n
        if x is not None:
            assert torch.allclose(x, y_, rtol=1e-9, atol=1e-11), n

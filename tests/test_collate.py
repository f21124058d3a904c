"""The per-iteration collate (train_ddi_modify.py:280,295-296) as index work: bmp_collate_plan / bmp_collate_pair_meta are
HOST functions of the C ABI, so they run here without a GPU.  Checked bit for bit against the numpy packer
(bmp.packed.pack_from_store, itself checked against the dense form in test_packed.py): the plan's placement equals
``_bin_pack``, and the plan + the store's local CSR reproduce every array of the packed batch (the device kernel
bmp_collate_emit is restated in numpy below; its GPU run is tests/test_gpu_collate.py)."""
import numpy as np
import pytest
import torch

from bmp import packed, synth


def emit_numpy(ds: "packed.DeviceMolStore", tab, I, n_tiles, E, R):
    """What k_collate_emit writes, one instance at a time."""
    rowoff, eoff, atom, rend, rendT, col, colT = ds.host_arrays
    N = n_tiles * R
    out = dict(atom_id=np.full(N, -7, np.int32), row_w=np.full(N, -7, np.float32), row_mol=np.full(N, -7, np.int32),
               csr_ptr=np.full(N + 1, -7, np.int32), csrT_ptr=np.full(N + 1, -7, np.int32), csr_col=np.full(E, -7, np.int32),
               csrT_col=np.full(E, -7, np.int32), csr_val=np.full(E, -7, np.float32), csrT_val=np.full(E, -7, np.float32))
    row0, nrows, mid, ebase, padw, ndead = (tab[k * I:(k + 1) * I] for k in range(6))
    for i in range(I):
        ro, eo = rowoff[mid[i]], eoff[mid[i]]
        ne = eoff[mid[i] + 1] - eo
        r = row0[i] + np.arange(nrows[i])
        out["atom_id"][r] = atom[ro:ro + nrows[i]]
        out["row_w"][r] = 1.0
        out["row_w"][r[-1]] = padw[i]
        out["row_mol"][r] = i
        out["csr_ptr"][r + 1] = ebase[i] + rend[ro:ro + nrows[i]]
        out["csrT_ptr"][r + 1] = ebase[i] + rendT[ro:ro + nrows[i]]
        dr = row0[i] + nrows[i] + np.arange(ndead[i])
        out["atom_id"][dr] = 0; out["row_w"][dr] = 0; out["row_mol"][dr] = -1
        out["csr_ptr"][dr + 1] = ebase[i] + ne; out["csrT_ptr"][dr + 1] = ebase[i] + ne
        out["csr_col"][ebase[i]:ebase[i] + ne] = col[eo:eo + ne] + (row0[i] << 2)
        out["csrT_col"][ebase[i]:ebase[i] + ne] = colT[eo:eo + ne] + (row0[i] << 2)
        out["csr_val"][ebase[i]:ebase[i] + ne] = 1.0
        out["csrT_val"][ebase[i]:ebase[i] + ne] = 1.0
        if row0[i] == 0:
            out["csr_ptr"][0] = 0; out["csrT_ptr"][0] = 0
    return out


@pytest.mark.parametrize("n_mols,B,seed,pad_to", [(60, 37, 0, None), (544, 1024, 1, None), (30, 8, 2, (70, 64)),
                                                   (12, 1, 3, None), (300, 256, 4, None)])
def test_plan_and_emit_equal_host_packer(n_mols, B, seed, pad_to):
    store = synth.make_store(n_mols, seed=10 + seed, n_lo=1, n_hi=60 if pad_to else 96, n_mean=22)
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, "cpu")
    rs = np.random.RandomState(seed)
    sides = [rs.randint(0, n_mols, B), rs.randint(0, n_mols, B)]
    ref = packed.pack_from_store(ms, sides, pad_to=pad_to)
    tab, side_tiles, side_mols, n_tiles, E, n_real, max_rows = packed.collate_plan_host(ds.st_nrows, ds.st_nedges, sides, pad_to=pad_to)
    I = 2 * B
    assert (side_tiles, side_mols, n_tiles, E, n_real, max_rows) == (
        ref.side_tiles, ref.side_mols, ref.n_tiles, ref.n_edges, ref.n_real_atoms, ref.max_rows_per_mol)
    assert np.array_equal(tab[:I], ref.mol_row0.numpy()) and np.array_equal(tab[I:2 * I], ref.mol_nrows.numpy())
    got = emit_numpy(ds, tab, I, n_tiles, E, ref.R)
    for k, v in got.items():
        assert np.array_equal(v, getattr(ref, k).numpy()), k          # every element written, every element equal


def test_plan_placement_is_first_fit_decreasing():
    rs = np.random.RandomState(5)
    for trial in range(24):
        n = rs.randint(1, 400)
        sizes = rs.randint(1, 129, size=n).astype(np.int64)
        if trial % 3 == 2:          # a few molecules larger than a tile: whole consecutive tiles at the head of the side
            big = rs.choice(n, min(n, rs.randint(1, 5)), replace=False)
            sizes[big] = rs.randint(129, 700, size=len(big))
        bins, offs, nb = packed._bin_pack(sizes, 128)
        nedges = np.zeros(n, np.int32)
        tab, side_tiles, _sm, n_tiles, *_ = packed.collate_plan_host(sizes.astype(np.int32), nedges, [np.arange(n)])
        assert n_tiles == nb and np.array_equal(tab[:n], bins * 128 + offs)
        # dead rows: exactly the rows no instance covers
        cover = np.zeros(nb * 128, np.int32)
        for i in range(n):
            cover[tab[i]:tab[i] + sizes[i] + tab[5 * n + i]] += 1
        assert (cover == 1).all()


def test_pair_meta_equals_numpy_form():
    from bmp import _lib
    from bmp.coattention import _cbuf_floats, _size_classes
    store = synth.make_store(200, seed=3, n_lo=1, n_hi=120, n_mean=30)
    store[7] = synth.make_store(1, seed=8, n_lo=140, n_hi=141, n_mean=140)[0]        # two molecules larger than a tile
    store[90] = synth.make_store(1, seed=9, n_lo=300, n_hi=301, n_mean=300)[0]
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, "cpu")
    rs = np.random.RandomState(1)
    B = 333
    sides = [rs.randint(0, 200, B), rs.randint(0, 200, B)]
    tab, side_tiles, *_ = packed.collate_plan_host(ds.st_nrows, ds.st_nedges, sides)
    meta = np.zeros(8 * B, np.int32); cnt = np.zeros(6, np.int32); ct = np.zeros(1, np.int64)
    _lib.check(_lib.lib().bmp_collate_pair_meta(packed._i32p(tab), 2 * B, B, side_tiles[1], 128, packed._i32p(meta),
                                                packed._i32p(cnt), packed._i32p(ct)), "pair_meta")
    nr = tab[2 * B:4 * B].astype(np.int64)
    nr1, nr2 = nr[:B], nr[B:]
    coff = np.concatenate(([0], np.cumsum(_cbuf_floats(nr1, nr2))))
    order, counts, order_f, counts_f, np_big = _size_classes(nr1, nr2, "cpu")
    assert np.array_equal(meta[:2 * B].view(np.int64), coff[:-1]) and ct[0] == coff[-1]
    assert np.array_equal(meta[2 * B:3 * B], tab[:B]) and np.array_equal(meta[3 * B:4 * B], nr1)
    assert np.array_equal(meta[4 * B:5 * B], tab[B:2 * B] - side_tiles[1] * 128) and np.array_equal(meta[5 * B:6 * B], nr2)
    assert np.array_equal(meta[6 * B:7 * B], order.numpy()) and np.array_equal(meta[7 * B:], order_f.numpy())
    assert list(cnt[:5]) == counts and cnt[5] == np_big and counts[4] > 0 and np_big >= 300


def test_plan_rejects_bad_input():
    nrows = np.array([5, 200], np.int32); ne = np.zeros(2, np.int32)
    tab, st, *_ = packed.collate_plan_host(nrows, ne, [np.array([1, 0])])     # larger than a tile: two whole tiles, then the rest
    assert st == (0, 3) and list(tab[:2]) == [0, 256] and list(tab[10:12]) == [56, 123]
    with pytest.raises(ValueError):
        packed.collate_plan_host(nrows, ne, [np.array([2])])              # molecule index out of range
    with pytest.raises(ValueError):
        packed.collate_plan_host(nrows, ne, [np.array([0])], pad_to=[2])  # pad_to below the side's largest molecule


def test_plan_equals_host_packer_on_random_stores():
    """Property test (hypothesis): for arbitrary molecule sizes, batch compositions and paddings the C++ plan + the emit
    rule reproduce the numpy packer exactly -- including one-molecule tiles, molecules that fill a tile to the last row and
    repeated molecules."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 60), st.integers(1, 90), st.integers(0, 10 ** 6), st.booleans(), st.integers(0, 30))
    def check(n_mols, B, seed, two_sided, extra_pad):
        rs = np.random.RandomState(seed)
        mols = []
        for _ in range(n_mols):
            n = int(rs.randint(1, 128))                      # up to 127 atoms: with the pad row exactly a full tile
            if rs.uniform() < 0.04:
                n = int(rs.randint(128, 420))                # ... and now and then a molecule larger than a tile
            atoms = rs.randint(1, 100, size=n).astype(np.int32)
            nb = int(rs.randint(0, 2 * n))
            i, j = rs.randint(0, n, nb), rs.randint(0, n, nb)
            keep = i != j
            key = {}
            for a_, b_, t_ in zip(i[keep], j[keep], rs.randint(0, 4, keep.sum())):
                key[(min(a_, b_), max(a_, b_))] = t_          # one bond per atom pair
            bonds = np.array([(a_, b_, t_) for (a_, b_), t_ in key.items()], dtype=np.int32).reshape(-1, 3)
            mols.append(synth.Molecule(atoms, bonds))
        ms = packed.MolStore(mols)
        ds = packed.DeviceMolStore(ms, "cpu")
        sides = [rs.randint(0, n_mols, B)] + ([rs.randint(0, n_mols, B)] if two_sided else [])
        pad_to = None
        if extra_pad:
            pad_to = [int(ms.n_atoms[s].max()) + extra_pad for s in sides]
        ref = packed.pack_from_store(ms, sides, pad_to=pad_to)
        tab, side_tiles, side_mols, n_tiles, E, n_real, max_rows = packed.collate_plan_host(ds.st_nrows, ds.st_nedges, sides, pad_to=pad_to)
        assert (side_tiles, side_mols, n_tiles, E, n_real, max_rows) == (
            ref.side_tiles, ref.side_mols, ref.n_tiles, ref.n_edges, ref.n_real_atoms, ref.max_rows_per_mol)
        got = emit_numpy(ds, tab, len(tab) // 6, n_tiles, E, ref.R)
        for k, v in got.items():
            assert np.array_equal(v, getattr(ref, k).numpy()), k

    check()


def test_one_molecule_per_tile_plan_holds_the_same_molecules():
    """bmp.packed.plan_one_per_tile (the fixed-shape batch a recorded step replays on): the emit kernel's restatement, run on
    that plan, writes every row of 2 B tiles, and every instance's rows, pad multiplicity and bonds are those of the usual
    packed batch, shifted to the instance's own tile."""
    R, B = 128, 24
    store = synth.make_store(70, seed=21, n_lo=1, n_hi=100, n_mean=25)
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, "cpu")
    rs = np.random.RandomState(4)
    sides = [rs.randint(0, 70, B), rs.randint(0, 70, B)]
    I = 2 * B
    tab = np.zeros(6 * I, dtype=np.int32); mt = np.zeros(2 * I, dtype=np.int32)
    I_, E, n_real = packed.plan_one_per_tile(ds.st_nrows, ds.st_nedges, sides, R, tab, mt)
    ref = packed.pack_from_store(ms, sides)
    assert (I_, E, n_real) == (I, ref.n_edges, ref.n_real_atoms)
    assert np.array_equal(tab[I:2 * I], ref.mol_nrows.numpy())
    assert np.array_equal(mt[:I], np.arange(I) * R) and np.array_equal(mt[I:], (tab[I:2 * I] + 31) // 32)
    got = emit_numpy(ds, tab, I, I, E, R)
    for k, v in got.items():
        assert not (v == -7).any(), k                      # every element written (the dead rows through ndead)
    r0, rr0, nr = tab[:I], ref.mol_row0.numpy(), ref.mol_nrows.numpy()
    rw, rp, rc = ref.row_w.numpy(), ref.csr_ptr.numpy(), ref.csr_col.numpy()
    for i in range(I):
        a, b = slice(r0[i], r0[i] + nr[i]), slice(rr0[i], rr0[i] + nr[i])
        assert np.array_equal(got["atom_id"][a], ref.atom_id.numpy()[b]) and np.array_equal(got["row_w"][a], rw[b])
        assert (got["row_mol"][a] == i).all()
        dead = slice(r0[i] + nr[i], r0[i] + R)
        assert (got["row_w"][dead] == 0).all() and (got["row_mol"][dead] == -1).all()
        assert (np.diff(got["csr_ptr"][r0[i] + nr[i]:r0[i] + R + 1]) == 0).all()          # dead rows: no bonds
        for l in range(nr[i]):                                                             # the same bonds, other row numbers
            mine = got["csr_col"][got["csr_ptr"][r0[i] + l]:got["csr_ptr"][r0[i] + l + 1]]
            theirs = rc[rp[rr0[i] + l]:rp[rr0[i] + l + 1]]
            assert np.array_equal(mine - (r0[i] << 2), theirs - (rr0[i] << 2))
    with pytest.raises(ValueError):
        packed.plan_one_per_tile(ds.st_nrows, ds.st_nedges, sides, 64, tab, mt)            # a 100-atom molecule, 64-row tiles

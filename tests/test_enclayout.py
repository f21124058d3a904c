"""The encoder layout (bmp/enclayout.py; csrc/bmp_collate.hip bmp_collate_plan_enc): real atoms + one pad row per tile,
tiles of 1..4 live 32-row blocks, optional de-duplication.  CPU tests: the C++ plan equals its numpy statement; the layout
covers every row exactly once; and -- in float64 against the dense oracle -- the encoder run on this layout, expanded to the
per-instance rows, gives the reference's padded atom states, co-attention outputs and EVERY gradient (the pad rows' gradients
come back as one sum per tile)."""
import numpy as np
import pytest
import torch

import packed_ref as PR
from bmp import enclayout, packed, synth
from oracle import ref_cpu as O

T = torch.from_numpy


@pytest.mark.parametrize("dedup", [False, True])
def test_cpp_plan_equals_numpy_plan(dedup):
    rs = np.random.RandomState(7)
    for trial, (n_mols, I, n_cu) in enumerate([(544, 2048, 256), (544, 64, 256), (60, 512, 256), (30, 7, 4), (200, 900, 16),
                                                (544, 2048, 40), (5, 1, 256), (300, 4096, 256)]):
        store = synth.make_store(n_mols, seed=40 + trial, n_lo=1, n_hi=127 if trial % 2 else 96, n_mean=24)
        ms = packed.MolStore(store)
        ds = packed.DeviceMolStore(ms, "cpu")
        mids = rs.randint(0, n_mols, I)
        ref = enclayout.plan_enc_numpy(ms.n_atoms, ms.nedges, mids, dedup, n_cu)
        v, (U, Tn, N, E, n_real, budget) = enclayout.plan_enc_host(ds.st_nrows, ds.st_nedges, mids, dedup, n_cu)
        assert (U, Tn, N, E, n_real, budget) == (ref["U"], ref["T"], ref["N"], ref["n_edges"], ref["n_real"], ref["budget"])
        tab = v["tab"]
        for k, name in enumerate(("row0", "n", "umid", "ebase")):
            assert np.array_equal(tab[k * U:(k + 1) * U], ref[name]), name
        assert (tab[4 * U:5 * U] == 1).all() and np.array_equal(tab[5 * U:6 * U], ref["ndead"])
        for name in ("tile_last", "enc_pad", "tmols"):
            assert np.array_equal(v[name][:U], ref[name]), name
        assert np.array_equal(v["uid"], ref["uid"]) and np.array_equal(v["uinst"], ref["uinst"])
        assert np.array_equal(v["uptr"][:U + 1], ref["uptr"]) and np.array_equal(v["tptr"][:Tn + 1], ref["tptr"])
        assert np.array_equal(v["mt_row0"][:Tn], ref["mt_row0"]) and np.array_equal(v["mt_nblk"][:Tn], ref["mt_nblk"])
        # every row is covered exactly once: real atoms, then (for a tile's last molecule) the dead rows behind it
        cover = np.zeros(N, np.int32)
        for u in range(U):
            cover[ref["row0"][u]:ref["row0"][u] + ref["n"][u] + ref["ndead"][u]] += 1
        assert (cover == 1).all() and N % 128 == 0
        assert (ref["mt_nblk"] >= 1).all() and (ref["mt_nblk"] <= 4).all()
        # a tile holds its molecules and its pad row
        for t in range(Tn):
            mem = ref["tmols"][ref["tptr"][t]:ref["tptr"][t + 1]]
            assert ref["n"][mem].sum() + (1 if len(mem) else 0) <= 32 * ref["mt_nblk"][t]


def test_headline_batches_need_seven_block_rounds():
    """The point of the tile heights: a 1024-pair batch of the 544-drug store is scheduled as 4 + 3 blocks per CU."""
    store = synth.make_store(); ms = packed.MolStore(store)
    i1, i2, _ = synth.make_pairs()
    import heapq
    worst = []
    for k in (0, 1, 2, 50, 143):
        mids = np.concatenate((i1[k * 1024:(k + 1) * 1024], i2[k * 1024:(k + 1) * 1024]))
        pl = enclayout.plan_enc_numpy(ms.n_atoms, ms.nedges, mids, False)
        h = [0] * 256; heapq.heapify(h)
        for x in pl["mt_nblk"]:
            heapq.heappush(h, heapq.heappop(h) + int(x))
        worst.append(max(h))
        assert pl["budget"] == 7 and pl["N"] <= 57344 + 128
    assert max(worst) == 7
    pl = enclayout.plan_enc_numpy(ms.n_atoms, ms.nedges, np.concatenate((i1[:32], i2[:32])), False)      # the reference's batch of 32
    assert pl["budget"] == 1 and pl["T"] >= 40 and int(pl["mt_nblk"].max()) <= 3


def test_oversized_molecule_keeps_the_instance_layout():
    store = synth.make_store(4, seed=2, n_lo=3, n_hi=9, n_mean=5) + [synth._make_molecule(np.random.RandomState(0), 130, 130, 130.0)]
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, "cpu")
    assert enclayout.plan_enc_numpy(ms.n_atoms, ms.nedges, np.array([0, 4, 1]), False) is None
    assert enclayout.plan_enc_host(ds.st_nrows, ds.st_nedges, np.array([0, 4, 1]), False) is None


@pytest.mark.parametrize("dedup,n_cu", [(False, 256), (True, 256), (False, 3), (True, 2)])
def test_encoder_layout_matches_dense_oracle_with_grads(dedup, n_cu):
    store = synth.make_store(10, seed=9, n_lo=2, n_hi=40, n_mean=12)
    ms = packed.MolStore(store)
    i1 = np.array([0, 1, 2, 0, 3, 3, 8, 1, 0, 5, 5, 2, 9]); i2 = np.array([1, 0, 0, 0, 4, 3, 1, 8, 7, 5, 2, 2, 6])
    B = len(i1)
    p = O.make_pair_params(777, hidden_dim=8, out_dim=8, n_layers=3, attn="nie", dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    eb = enclayout.encode_from_store(ms, [i1, i2], dedup=dedup, n_cu=n_cu, with_dense_map=True)
    assert eb.n_encoded == (len(np.unique(np.concatenate((i1, i2)))) if dedup else 2 * B)
    assert eb.pb_enc.n_rows % 128 == 0 and eb.pb_enc.n_mtiles == len(eb.pb_enc.mt_nblk)
    # encoder on the encoder layout, rows copied to the instance layout
    _, h_enc = PR.ggnn_forward(p, eb.pb_enc, 3, prefix="graph_conv/")
    h0_enc = p["graph_conv/embed/W"][eb.pb_enc.atom_id.long()]
    X = enclayout.expand_rows_host(h_enc, eb)
    X0 = enclayout.expand_rows_host(h0_enc, eb)
    ge1, at1 = O.ggnn_forward(p, T(a1), T(j1).double(), 3, prefix="graph_conv/")
    ge2, at2 = O.ggnn_forward(p, T(a2), T(j2).double(), 3, prefix="graph_conv/")
    assert torch.allclose(eb.pb.to_dense(X, 0), at1, atol=1e-12) and torch.allclose(eb.pb.to_dense(X, 1), at2, atol=1e-12)
    # readout on the instance rows (models/ggnn.py:333-341: the sum runs over the padded positions too)
    w = eb.pb.row_w.double()[:, None]
    gi = torch.sigmoid(torch.cat((X, X0), 1) @ p["graph_conv/i_layers/0/W"].t() + p["graph_conv/i_layers/0/b"])
    gj = X @ p["graph_conv/j_layers/0/W"].t() + p["graph_conv/j_layers/0/b"]
    g = PR.segment_sum(eb.pb, w * gi * gj)
    assert torch.allclose(g[:B], ge1, atol=1e-11) and torch.allclose(g[B:], ge2, atol=1e-11)
    # co-attention on the instance rows; gradients of everything through the expand (whose autograd is the reduce)
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=3, attn="nie")
    c1, c2 = PR.nie_coattention(p, eb.pb, X, np.arange(B), B + np.arange(B), prefix="attn/")
    assert torch.allclose(c1, g1, atol=1e-12) and torch.allclose(c2, g2, atol=1e-12)
    names = [n for n in sorted(p) if not n.startswith("mlp/")]
    w1 = torch.randn(B, 8, dtype=torch.float64); w2 = torch.randn(B, 8, dtype=torch.float64); wg = torch.randn(2 * B, 8, dtype=torch.float64)
    gp = torch.autograd.grad((c1 * w1).sum() + (c2 * w2).sum() + (g * wg).sum(), [p[n] for n in names], allow_unused=True)
    gd = torch.autograd.grad((g1 * w1).sum() + (g2 * w2).sum() + (torch.cat((ge1, ge2)) * wg).sum(), [p[n] for n in names], allow_unused=True)
    for n, x, y_ in zip(names, gp, gd):
        assert (x is None) == (y_ is None), n
        if x is not None:
            assert torch.allclose(x, y_, rtol=1e-9, atol=1e-11), n

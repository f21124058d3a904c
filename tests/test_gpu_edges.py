"""Edge cases of the packed path against the dense oracle: the largest molecule a tile can hold (127 atoms + the
virtual pad row = 128 rows, which also puts its pair in the 128-row co-attention size class the benchmark data never
reaches), single-atom molecules without bonds, a one-pair batch, equal-size pairs (pad rows of weight zero) and
molecules that do not fit a tile at all (130, 200, 300 atoms: several tiles each)."""
import numpy as np
import pytest
import torch

from bmp import synth
from oracle import ref_cpu as O

T = torch.from_numpy


def _mol(rs, n):
    return synth._make_molecule(rs, n, n, float(n))


def _run(dev, store, i1, i2, d=64, nl=2, attn="nie", tol=1e-4, encoder="ggnn"):
    from bmp import packed
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import grad_dict, load_param_dict
    B = len(i1)
    p = O.make_pair_params(777, encoder=encoder, hidden_dim=d, out_dim=d, n_layers=nl, attn=attn, head=8, dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    label = T((np.arange(B).reshape(-1, 1) % 2).astype(np.int32))
    y, _, _ = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), encoder=encoder, n_layers=nl, attn=attn)
    O.sigmoid_cross_entropy(y, label).backward()
    model = build_pair_predictor(hidden_dim=d, out_dim=d, n_layers=nl, attn=attn, head=8, encoder=encoder).to(dev)
    load_param_dict(model, p)
    pb = packed.pack_from_store(packed.MolStore(store), [np.asarray(i1), np.asarray(i2)], device=dev)
    yd = model(pb)
    model.loss(yd, label.to(dev)).backward()

    def close(got, want, name):
        from parity_util import close as _c
        _c(got, want, name, tol)
    close(yd, y, "logits")
    for name, gr in grad_dict(model).items():
        if p[name].grad is not None:
            close(gr, p[name].grad, f"grad {name}")
    return pb


@pytest.mark.gpu
def test_largest_molecule_fills_a_tile_and_the_128_row_pair_class():
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    store = [_mol(rs, 127), _mol(rs, 127), _mol(rs, 5), _mol(rs, 100), _mol(rs, 33), _mol(rs, 64)]
    pb = _run(dev, store, [0, 2, 3, 5], [1, 0, 4, 3])
    assert pb.max_rows_per_mol == 128
    from bmp.coattention import _size_classes
    _, counts, _, counts_f, _ = _size_classes(pb.mol_nrows_host[:4], pb.mol_nrows_host[4:], "cpu")
    assert counts[3] >= 2 and counts_f[3] == 4          # 128-row class present; the forward runs everything in it


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["nie", "pool", None])
def test_single_atoms_one_pair_and_equal_sizes(attn):
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(1)
    lone = synth.Molecule(atoms=np.array([8], np.int32), bonds=np.zeros((0, 3), np.int32))       # one atom, no bond
    store = [lone, _mol(rs, 7), _mol(rs, 7), lone, _mol(rs, 2)]
    _run(dev, store, [0], [1], d=64, nl=2, attn=attn)                 # one pair; side 1 is a single atom
    _run(dev, store, [1, 2], [2, 1], d=64, nl=3, attn=attn)           # equal sizes everywhere: every pad row has weight 0
    _run(dev, store, [0, 3, 4], [3, 0, 4], d=64, nl=2, attn=attn)     # single atoms on both sides


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["nie", None])
def test_relgcn_fused_layers_on_the_same_edge_cases(attn):
    """The fused RelGCN layer and the readout tile kernel (d = 64) on a tile-filling molecule, single atoms without
    bonds (degree 0: rescale_adj divides by 1) and a one-pair batch."""
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(3)
    lone = synth.Molecule(atoms=np.array([7], np.int32), bonds=np.zeros((0, 3), np.int32))
    store = [_mol(rs, 127), _mol(rs, 90), lone, _mol(rs, 31), _mol(rs, 64), lone]
    _run(dev, store, [0, 2, 3], [1, 4, 5], d=64, nl=2, attn=attn, encoder="relgcn")
    _run(dev, store, [2], [5], d=64, nl=3, attn=attn, encoder="relgcn")


@pytest.mark.gpu
@pytest.mark.parametrize("attn", ["nie", "pool", None])
@pytest.mark.parametrize("encoder", ["ggnn", "relgcn"])
def test_molecules_larger_than_a_tile_match_the_oracle(attn, encoder):
    """The reference builds its dataset without a size limit (train_ddi_modify.py:256, parsers.py:156-335) and concat_mols
    pads to whatever the batch holds: molecules of 130 and 300 atoms (2 and 3 tiles) next to small ones, Nie / Pooling /
    no co-attention, GGNN and RelGCN at a width the fused kernels cover (so the tile-local kernels must step aside for the
    row-wise operators and the pair kernels take their global-memory class) -- logits and every gradient vs the oracle."""
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(5)
    store = [_mol(rs, 130), _mol(rs, 9), _mol(rs, 300), _mol(rs, 40), _mol(rs, 127), _mol(rs, 3), _mol(rs, 128)]
    pb = _run(dev, store, [0, 1, 2, 3, 5, 6], [3, 2, 0, 4, 6, 1], d=64, nl=2, attn=attn, encoder=encoder)
    assert pb.oversized and pb.max_rows_per_mol == 301
    if attn is not None:
        from bmp.coattention import _size_classes
        _, counts, _, counts_f, np_big = _size_classes(pb.mol_nrows_host[:6], pb.mol_nrows_host[6:], "cpu")
        assert counts[4] == 5 and counts_f[4] == 5 and np_big == 301 and sum(counts) == 6


@pytest.mark.gpu
def test_oversized_batch_through_the_planned_path_and_the_device_collate():
    """FlatAdam.functional_forward (layout plan on) on a batch with a 200-atom molecule, collated on the device: the encoder
    leaves the plan's fused kernels for the row-wise operators (its gradients come back through autograd), the co-attention
    stays planned; flat gradient vs the oracle, twice (nothing stale between steps), then a normal batch on the same plan."""
    from bmp import packed
    from bmp.dp import FlatAdam
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from parity_util import close
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(9)
    store = [_mol(rs, 200), _mol(rs, 12), _mol(rs, 33), _mol(rs, 64), _mol(rs, 5), _mol(rs, 90)]
    ms = packed.MolStore(store)
    p = O.make_pair_params(777, hidden_dim=128, out_dim=128, n_layers=3, attn="nie", head=8, dtype=torch.float64, bias_scale=0.05)
    model = build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=3, attn="nie", head=8).to(dev)
    load_param_dict(model, p)
    opt = FlatAdam(model, alpha=1e-3)
    ds = packed.DeviceMolStore(ms, dev)
    for i1, i2 in (([0, 1, 2, 3], [4, 0, 5, 1]), ([1, 2, 3, 4], [5, 4, 1, 2]), ([3, 0], [0, 0])):
        i1, i2 = np.asarray(i1), np.asarray(i2)
        lab = (np.arange(len(i1)).reshape(-1, 1) % 2).astype(np.int32)
        po = {k: v.clone().requires_grad_() for k, v in p.items()}
        a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
        yo, _, _ = O.pair_forward(po, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=3, attn="nie")
        lo = O.sigmoid_cross_entropy(yo, T(lab))
        names = sorted(po)
        go = dict(zip(names, torch.autograd.grad(lo, [po[n] for n in names], allow_unused=True)))
        pb, t = packed.pack_from_store_device(ds, [i1, i2], labels=lab)
        ref = packed.pack_from_store(ms, [i1, i2])
        for k in ("atom_id", "row_w", "csr_ptr", "csr_col", "csrT_ptr", "csrT_col", "mol_row0", "mol_nrows", "row_mol"):
            assert torch.equal(getattr(pb, k).cpu(), getattr(ref, k)), k          # the device collate == the numpy packer
        assert pb.oversized == (0 in i1 or 0 in i2)
        y = opt.functional_forward(pb)
        loss = model.loss(y, t)
        loss.backward()
        opt.collect_grads()
        close(y, yo, "logits"); close(loss, lo, "loss")
        off = 0
        for name, shp in zip(opt.names, opt.shapes):
            n = int(np.prod(shp))
            want = go[name.replace(".", "/")]
            close(opt.grad[off:off + n].view(shp), want if want is not None else torch.zeros(shp, dtype=torch.float64), f"grad {name}")
            off += n


@pytest.mark.gpu
def test_dense_atom_arrays_wider_than_a_tile_into_the_coattention():
    """The reference hands the co-attention dense (mb, N, hid) arrays (nie_coattention.py:335-341): N = 150 > 128 positions."""
    from bmp.coattention import NieFineCoattention
    from bmp.snapshot import load_param_dict, grad_dict
    from parity_util import close
    dev = torch.device("cuda:0")
    d, o, mb, N1, N2 = 32, 16, 3, 150, 40
    dr = O._Draw(5, torch.float64, 0.2)
    O.init_nie(dr, "", d, o, 8)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(mb, N1, d, generator=g, dtype=torch.float64, requires_grad=True)
    x2 = torch.randn(mb, N2, d, generator=g, dtype=torch.float64, requires_grad=True)
    c1, c2 = O.nie_coattention(p, x1, x2, "tanh")
    w1 = torch.randn(mb, o, generator=g, dtype=torch.float64); w2 = torch.randn(mb, o, generator=g, dtype=torch.float64)
    ((c1 * w1).sum() + (c2 * w2).sum()).backward()
    att = NieFineCoattention(d, o, 8, activation="tanh").to(dev)
    load_param_dict(att, p)
    y1 = x1.detach().float().to(dev).requires_grad_(); y2 = x2.detach().float().to(dev).requires_grad_()
    o1, o2 = att(y1, None, y2, None)
    close(o1, c1, "compact_1"); close(o2, c2, "compact_2")
    ((o1 * w1.float().to(dev)).sum() + (o2 * w2.float().to(dev)).sum()).backward()
    close(y1.grad, x1.grad, "d atoms_1"); close(y2.grad, x2.grad, "d atoms_2")
    for name, gr in grad_dict(att).items():
        close(gr, p[name].grad, f"grad {name}")

"""Edge cases of the packed path against the dense oracle: the largest molecule a tile can hold (127 atoms + the
virtual pad row = 128 rows, which also puts its pair in the 128-row co-attention size class the benchmark data never
reaches), single-atom molecules without bonds, a one-pair batch, equal-size pairs (pad rows of weight zero) and a
molecule that does not fit a tile at all."""
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
    _, counts, _, counts_f = _size_classes(pb.mol_nrows_host[:4], pb.mol_nrows_host[4:], "cpu")
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


def test_molecule_larger_than_a_tile_is_rejected():
    from bmp import packed
    rs = np.random.RandomState(2)
    with pytest.raises(ValueError):
        packed.pack_from_store(packed.MolStore([_mol(rs, 128)]), [np.array([0])])

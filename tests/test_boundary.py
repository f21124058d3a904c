"""Boundary behaviour that needs no GPU: the reference's lazy constructor forms, argument checks that raise before any
kernel launch (the reference raises ValueError for bad options / inputs, e.g. models/ggnn.py:250; EmbedID type-checks its ids)."""
import numpy as np
import pytest
import torch

from bmp import packed, synth
from bmp.mlp import MLP


def test_mlp_lazy_input_width_as_the_reference_builds_it():
    """train_ddi_modify.py:136: MLP(out_dim=class_num, hidden_dims=net_hidden_dims) -- no input width."""
    torch.manual_seed(0)
    mlp = MLP(out_dim=1, hidden_dims=(32, 16))
    assert isinstance(mlp.layers[0].W, torch.nn.UninitializedParameter)
    x = torch.randn(5, 24)
    y = mlp(x)                                   # models/mlp.py:40-45 on the concatenated pair vector
    assert y.shape == (5, 1) and mlp.layers[0].W.shape == (32, 24) and mlp.in_dim == 24
    y2 = mlp(x[:, :10], x[:, 10:])               # the [g1 | g2] call form of the pair glue: same function
    assert torch.allclose(y, y2, atol=1e-6)
    with pytest.raises(ValueError):
        mlp(torch.randn(5, 30))                  # a second width is an error, as in Chainer


def test_link_predictors_lazy_fp_dim():
    from bmp.link import HolE, SymMLP
    for cls, width in ((SymMLP, 2 * 12), (HolE, 12)):
        m = cls(out_dim=3, hidden_dims=(8,))
        assert isinstance(m.layers[0].W, torch.nn.UninitializedParameter)
        m.materialize_input(12)
        assert m.layers[0].W.shape == (8, width)


def test_predictor_materialises_the_lazy_mlp_and_takes_both_positional_forms():
    from bmp.coattention import NieFineCoattention
    from bmp.dp import FlatAdam
    from bmp.ggnn import GGNN
    from bmp.predictor import GraphConvPredictorForPair
    ggnn = GGNN(out_dim=24, hidden_dim=16, n_layers=2)
    p = GraphConvPredictorForPair(ggnn, MLP(out_dim=1, hidden_dims=(32, 16)))           # train_ddi_modify.py:150
    assert p.attn is None and p.mlp.layers[0].W.shape == (32, 48)
    attn = NieFineCoattention(hidden_dim=16, out_dim=8, head=4, activation="tanh")
    p2 = GraphConvPredictorForPair(ggnn, attn, MLP(out_dim=1, hidden_dims=(32, 16)))    # train_binary.py:264
    assert p2.attn is attn and p2.mlp.layers[0].W.shape == (32, 16)
    FlatAdam(p2)                                                                        # every parameter exists
    ggnn_c = GGNN(out_dim=8, hidden_dim=16, n_layers=3, concat_hidden=True)
    assert GraphConvPredictorForPair(ggnn_c, MLP(out_dim=1)).mlp.layers[0].W.shape == (32, 2 * 3 * 8)
    with pytest.raises(ValueError):
        FlatAdam(MLP(out_dim=1))                 # still lazy: refuse to flatten


def test_atom_ids_outside_the_embedding_table_are_rejected_on_the_host():
    """an id >= n_atom_types (or negative) must raise before a kernel indexes the table."""
    store = synth.make_store(6, seed=1, n_lo=3, n_hi=8, n_mean=5)
    ms = packed.MolStore(store)
    pb = packed.pack_from_store(ms, [np.arange(3), np.arange(3, 6)])
    lo, hi = pb.atom_id_range
    assert lo == 0 and hi == int(ms.atom_flat.max())
    pb.check_atom_ids(117)
    with pytest.raises(ValueError):
        pb.check_atom_ids(hi)                    # a table one row too short
    a = np.array([[6, 200, 0]], np.int32); adj = np.zeros((1, 4, 3, 3), np.float32)
    with pytest.raises(ValueError):
        packed.pack_from_dense([a], [adj]).check_atom_ids(117)
    a[0, 1] = -3
    with pytest.raises(ValueError):
        packed.pack_from_dense([a], [adj]).check_atom_ids(117)


def test_real_atom_count_ignores_weight_one_pad_rows():
    """a molecule with n = A - 1 has a virtual pad row of multiplicity exactly 1; it is not a real atom."""
    store = synth.make_store(10, seed=2, n_lo=4, n_hi=12, n_mean=8)
    ms = packed.MolStore(store)
    idx = np.arange(10)
    pb = packed.pack_from_store(ms, [idx])
    assert pb.n_real_atoms == int(ms.n_atoms.sum())
    A = int(ms.n_atoms.max())
    pb2 = packed.pack_from_store(ms, [idx], pad_to=[A + 1])     # now the largest molecule's pad row has weight 1 too
    assert pb2.n_real_atoms == int(ms.n_atoms.sum())


def test_dropout_flag_is_accepted():
    from bmp.ggnn import GGNN
    enc = GGNN(out_dim=8, hidden_dim=16, n_layers=2, dropout_rate=0.2)      # train_ddi_modify.py:149
    assert enc.dropout_rate == 0.2
    with pytest.raises(ValueError):
        GGNN(out_dim=8, hidden_dim=16, dropout_rate=1.5)


def test_bimpm_builder_takes_its_head_from_the_output_width():
    """train_binary.py:253-256 builds BiMPM(head=fp_out_dim): the builder's generic ``head`` (8) is not BiMPM's."""
    from bmp.predictor import build_pair_predictor
    m = build_pair_predictor(hidden_dim=64, out_dim=16, n_layers=2, attn="bimpm")
    assert m.attn.head == 16 and m.attn.out_dim == 3 * 16
    assert tuple(m.mlp.layers[0].W.shape) == (32, 2 * 3 * 16)

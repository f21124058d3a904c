"""On-disk formats (bmp/snapshot.py): Chainer npz snapshot key layout and NumpyTupleDataset npz (parsers.py:91-120)."""
import numpy as np
import torch

from bmp.mlp import MLP
from bmp.snapshot import (TRAINER_PREFIX, load_chainer_snapshot, load_tuple_dataset, param_dict, save_chainer_snapshot,
                          save_tuple_dataset)
from torch import nn


class _Pred(nn.Module):                      # the link tree of GraphConvPredictorForPair without the HIP encoder
    def __init__(self):
        super().__init__()
        self.graph_conv = MLP(4, (6,), in_dim=5)
        self.mlp = MLP(1, (3, 2), in_dim=8)


def test_chainer_snapshot_round_trip_all_prefixes(tmp_path):
    torch.manual_seed(0)
    a, b = _Pred(), _Pred()
    for prefix in (TRAINER_PREFIX, "predictor/", ""):
        path = str(tmp_path / f"snap{len(prefix)}.npz")
        save_chainer_snapshot(path, a, prefix=prefix)
        keys = np.load(path).files
        assert prefix + "graph_conv/layers/0/W" in keys and prefix + "mlp/l_out/b" in keys
        assert load_chainer_snapshot(path, b) == prefix
        for (k, v), (_, w) in zip(param_dict(a).items(), param_dict(b).items()):
            assert torch.equal(v, w), k
        with torch.no_grad():
            for p in b.parameters():
                p.zero_()


def test_snapshot_with_adam_state(tmp_path):
    from bmp.dp import FlatAdam
    m = _Pred()
    opt = FlatAdam(m, alpha=1e-2)
    x = torch.randn(7, 5)
    for _ in range(2):
        opt.zero_grad()
        m.graph_conv(x).sum().backward()
        opt.step()
    path = str(tmp_path / "trainer.npz")
    save_chainer_snapshot(path, m, adam=opt)
    z = np.load(path)
    assert int(z["updater/optimizer:main/t"]) == 2
    mm = z["updater/optimizer:main/predictor/graph_conv/layers/0/W/m"]
    assert mm.shape == (6, 5) and np.abs(mm).max() > 0
    assert load_chainer_snapshot(path, _Pred()) == TRAINER_PREFIX          # optimizer entries are skipped


def test_tuple_dataset_round_trip(tmp_path):
    path = str(tmp_path / "ds.npz")
    arrs = (np.arange(6, dtype=np.int32).reshape(2, 3), np.ones((2, 4, 3, 3), np.float32), np.array([[1], [0]], np.int32))
    save_tuple_dataset(path, arrs)
    assert np.load(path).files == ["arr_0", "arr_1", "arr_2"]
    back = load_tuple_dataset(path)
    assert all(np.array_equal(x, y) for x, y in zip(arrs, back))
    assert load_tuple_dataset(str(tmp_path / "missing.npz")) is None

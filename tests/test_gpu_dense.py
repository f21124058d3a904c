"""bmp_dense_count / bmp_dense_to_csr: the reference's dense device batch -> packed CSR on the GPU, bit for bit the
host packer's result (integer / index work: exact)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same(pa, pb):
    ka = (pa.n_tiles, pa.n_mols, pa.side_tiles, pa.side_mols, pa.n_edges, pa.n_real_atoms)
    assert ka == (pb.n_tiles, pb.n_mols, pb.side_tiles, pb.side_mols, pb.n_edges, pb.n_real_atoms)
    for name in ("atom_id", "row_w", "csr_ptr", "csr_col", "csr_val", "csrT_ptr", "csrT_col", "csrT_val", "mol_row0", "mol_nrows"):
        x, y = getattr(pa, name).cpu(), getattr(pb, name).cpu()
        assert x.dtype == y.dtype and torch.equal(x, y), name
    for x, y in zip(pa.dense_maps, pb.dense_maps):
        assert torch.equal(x.cpu(), y.cpu())


def test_device_packer_equals_host_packer():
    from bmp import packed, synth
    dev = torch.device("cuda:0")
    store = synth.make_store(80, seed=12, n_lo=1, n_hi=90, n_mean=20)
    rs = np.random.RandomState(0)
    i1, i2 = rs.randint(0, 80, 40), rs.randint(0, 80, 40)
    a1, j1 = synth.concat_mols([store[k] for k in i1]); a2, j2 = synth.concat_mols([store[k] for k in i2])
    j1 = j1 * rs.uniform(0.5, 2.0, size=j1.shape).astype(np.float32)        # weighted (RelGCN-style) values survive
    host = packed.pack_from_dense([a1, a2], [j1, j2], device=dev)
    devp = packed.pack_from_dense_device([a1, a2], [torch.from_numpy(j1).to(dev), torch.from_numpy(j2).to(dev)])
    _same(host, devp)


def test_asymmetric_input_and_fallback():
    from bmp import packed
    dev = torch.device("cuda:0")
    atoms = np.array([[6, 7, 0, 0], [8, 0, 0, 0]], np.int32)
    adj = np.zeros((2, 4, 4, 4), np.float32)
    adj[0, 0, 0, 1] = 1.0; adj[0, 3, 1, 0] = 2.0          # asymmetric, but between real atoms: device path
    _same(packed.pack_from_dense([atoms], [adj], device=dev), packed.pack_from_dense_device([atoms], [torch.from_numpy(adj).to(dev)]))
    adj[1, 1, 0, 2] = 1.0                                   # a bond LEAVING a padded position: host fallback, same result
    _same(packed.pack_from_dense([atoms], [adj], device=dev), packed.pack_from_dense_device([atoms], [torch.from_numpy(adj).to(dev)]))


def test_encoder_takes_dense_device_arrays():
    """Drop-in form with device tensors: same embedding as with host arrays."""
    from bmp import synth
    from bmp.ggnn import GGNN
    dev = torch.device("cuda:0")
    store = synth.make_store(12, seed=3, n_lo=3, n_hi=30, n_mean=10)
    a, j = synth.concat_mols(store)
    torch.manual_seed(0)
    enc = GGNN(out_dim=16, hidden_dim=64, n_layers=2).to(dev)
    with torch.no_grad():
        g_host = enc(a, j)
        g_dev = enc(torch.from_numpy(a).to(dev), torch.from_numpy(j).to(dev))
    assert torch.equal(g_host, g_dev)

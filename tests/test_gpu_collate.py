"""bmp_collate_emit on the GPU: a batch built from index pairs against the HBM-resident drug store is bit for bit the
numpy packer's batch (integer / index work: exact), at a small size, at BASELINE.json's full batch size, with pad_to and
for a one-sided batch; the model gives identical results on either; batches built back to back from the pinned staging
ring do not overwrite each other."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ARRAYS = ("atom_id", "row_w", "csr_ptr", "csr_col", "csr_val", "csrT_ptr", "csrT_col", "csrT_val", "mol_row0", "mol_nrows", "row_mol")


def _same(pa, pb):
    ka = (pa.n_tiles, pa.n_mols, pa.side_tiles, pa.side_mols, pa.n_edges, pa.n_real_atoms, pa.max_rows_per_mol)
    assert ka == (pb.n_tiles, pb.n_mols, pb.side_tiles, pb.side_mols, pb.n_edges, pb.n_real_atoms, pb.max_rows_per_mol)
    assert np.array_equal(pa.mol_nrows_host, pb.mol_nrows_host)
    for name in ARRAYS:
        x, y = getattr(pa, name).cpu(), getattr(pb, name).cpu()
        assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), name


@pytest.mark.parametrize("n_mols,B,pad_to,two_sided", [(60, 37, None, True), (544, 1024, None, True), (40, 9, (100, 97), True),
                                                       (80, 50, None, False), (5, 1, None, True)])
def test_device_collate_equals_host_packer(n_mols, B, pad_to, two_sided):
    from bmp import packed, synth
    dev = torch.device("cuda:0")
    store = synth.make_store(n_mols, seed=21, n_lo=1, n_hi=96, n_mean=24)
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, dev)
    rs = np.random.RandomState(B)
    sides = [rs.randint(0, n_mols, B), rs.randint(0, n_mols, B)] if two_sided else [rs.randint(0, n_mols, B)]
    if pad_to is not None and not two_sided:
        pad_to = pad_to[:1]
    host = packed.pack_from_store(ms, sides, device=dev, pad_to=pad_to)
    got = packed.pack_from_store_device(ds, sides, pad_to=pad_to)
    _same(host, got)


def test_pair_meta_and_labels_ride_along():
    from bmp import packed, synth
    from bmp.coattention import pair_rows
    from bmp.ggnn import PackedAtoms
    dev = torch.device("cuda:0")
    store = synth.make_store(100, seed=4, n_lo=2, n_hi=96, n_mean=26)
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, dev)
    rs = np.random.RandomState(2)
    B = 200
    sides = [rs.randint(0, 100, B), rs.randint(0, 100, B)]
    lab = rs.randint(0, 2, (B, 1)).astype(np.int32)
    host = packed.pack_from_store(ms, sides, device=dev)
    got, lab_d = packed.pack_from_store_device(ds, sides, labels=lab)
    assert lab_d.shape == (B, 1) and np.array_equal(lab_d.cpu().numpy(), lab)
    rows = torch.zeros(host.n_rows, 8, device=dev)
    mh = pair_rows(PackedAtoms(rows, host), PackedAtoms(rows, host))[4]
    mg = pair_rows(PackedAtoms(rows, got), PackedAtoms(rows, got))[4]
    for k in ("B", "T1", "T2", "counts", "counts_f", "ctotal"):
        assert mh[k] == mg[k], k
    for k in ("r1", "n1", "r2", "n2", "coff", "order", "order_f"):
        assert mh[k].dtype == mg[k].dtype and torch.equal(mh[k].cpu(), mg[k].cpu()), k


def test_staging_ring_keeps_batches_apart():
    """More batches in flight than staging buffers, no synchronisation in between: every batch must still be its own."""
    from bmp import packed, synth
    dev = torch.device("cuda:0")
    store = synth.make_store(120, seed=8, n_lo=2, n_hi=90, n_mean=24)
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, dev)
    rs = np.random.RandomState(0)
    sides = [[rs.randint(0, 120, 300), rs.randint(0, 120, 300)] for _ in range(3 * ds.N_STAGE)]
    got = [packed.pack_from_store_device(ds, s) for s in sides]
    torch.cuda.synchronize()
    for s, g in zip(sides, got):
        _same(packed.pack_from_store(ms, s, device=dev), g)


def test_model_step_is_identical_on_either_batch():
    from bmp import packed, synth
    from bmp.predictor import build_pair_predictor
    dev = torch.device("cuda:0")
    store = synth.make_store(64, seed=6, n_lo=3, n_hi=60, n_mean=20)
    ms = packed.MolStore(store)
    ds = packed.DeviceMolStore(ms, dev)
    rs = np.random.RandomState(3)
    sides = [rs.randint(0, 64, 48), rs.randint(0, 64, 48)]
    t = torch.from_numpy(rs.randint(0, 2, (48, 1)).astype(np.int32)).to(dev)
    torch.manual_seed(1)
    model = build_pair_predictor(hidden_dim=64, out_dim=64, n_layers=3, attn="nie").to(dev)
    outs = []
    for pb in (packed.pack_from_store(ms, sides, device=dev), packed.pack_from_store_device(ds, sides)):
        model.zero_grad()
        y = model(pb)
        model.loss(y, t).backward()
        outs.append((y.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)

"""Rows by bond type (bmp_type_rows) and the weight-gradient launches that walk them (round 4).

The gathered gradient G_e of a row is an exact zero unless the row has a bond of type e, so the per-type blocks of a step's
weight gradients may sum over the listed rows only: the lists must be exactly those rows (bit-exact integer work, against
numpy), and the step's / layer's weight gradients with the lists must equal the all-rows launches up to float32 summation
order (the rows are dealt to other workgroups) -- and, like every other result, the float64 oracle (the oracle tests of
test_gpu_ops.py / test_gpu_fullsize_backward.py run with the lists on)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(n_pairs=96, seed=3):
    from bmp import packed, synth
    store = synth.make_store(60, seed=seed, n_lo=2, n_hi=70, n_mean=20)
    ms = packed.MolStore(store)
    rs = np.random.RandomState(seed)
    i1, i2 = rs.randint(0, 60, n_pairs), rs.randint(0, 60, n_pairs)
    return packed.pack_from_store(ms, [i1, i2], device=torch.device("cuda:0"))


def test_type_rows_are_the_rows_with_an_entry_of_the_type():
    pb = _batch()
    idx, cnt = pb.type_rows_T()
    torch.cuda.synchronize()
    N = pb.n_rows
    ptr, col = pb.csrT_ptr.cpu().numpy(), pb.csrT_col.cpu().numpy()
    rows = np.repeat(np.arange(N), np.diff(ptr))
    idx, cnt = idx.cpu().numpy().reshape(4, N), cnt.cpu().numpy()
    for e in range(4):
        want = np.unique(rows[(col & 3) == e])
        assert cnt[e] == len(want)
        assert np.array_equal(idx[e, :cnt[e]], want)          # ascending, exact
    assert cnt[0] > cnt[1] > cnt[2] >= 0 and pb.type_rows_T()[0] is pb._cache["type_rows_T"][0]        # built once, kept


@pytest.mark.parametrize("first", [True, False])
@pytest.mark.parametrize("d", [64, 128])
def test_step_wgrad_with_row_lists_equals_all_rows(d, first):
    from bmp import _lib
    from bmp._lib import check, ptr, stream
    from parity_util import close
    L = _lib.lib()
    pb = _batch(160, seed=5)
    dev = pb.device
    N = pb.n_rows
    g = torch.Generator().manual_seed(d + int(first))
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    h, m, rz = rnd(N, d), rnd(N, d), torch.rand(N, 2 * d, generator=g).to(dev)
    # a gda as the backward writes it: the G_e block of a row is zero unless the row has a bond of the type
    gda = rnd(N, 7 * d) * 1e-2
    idx, cnt = pb.type_rows_T()
    torch.cuda.synchronize()
    mask = torch.zeros(4, N, device=dev)
    for e in range(4):
        mask[e, idx[e * N: e * N + int(cnt[e])].long()] = 1.0
    for e in range(4):
        gda[:, e * d:(e + 1) * d] *= mask[e][:, None]
    out = []
    for lists in (False, True):
        o1, o2, dU, cs = (torch.full(s_, 7.0, device=dev) for s_ in ((d, 7 * d), (d, 3 * d), (d, d), (7 * d,)))
        nws = L.bmp_ggnn_step_wgrad_ws_floats(N, d)
        ws = torch.empty(nws, device=dev)
        for acc in (0, 1):                   # written, then accumulated into (tied layers): twice the sums
            check(L.bmp_ggnn_step_wgrad(ptr(h), ptr(m), ptr(rz), ptr(gda), N, d, int(first), ptr(o1), ptr(o2), ptr(dU), ptr(cs), acc,
                                        ptr(idx if lists else None), ptr(cnt if lists else None), None, None, ptr(ws), nws, stream()), "wgrad")
        torch.cuda.synchronize()
        out.append((o1.clone(), o2.clone(), dU.clone(), cs.clone()))
    ref = 2.0 * (h.double().t() @ gda.double())
    if first:
        ref[:, 4 * d:5 * d] = 0.0
    close(out[1][0], ref, f"o1 with lists vs float64 (d {d}, first {first})", tol=2e-5)
    for name, a_, b_ in zip(("o1", "o2", "dUcT", "cs"), out[0], out[1]):
        close(b_, a_.double(), f"{name} with lists vs all rows", tol=2e-5, floor=1e-3 * float(a_.abs().max()) + 1e-12)
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])       # o2 / dUcT: the same products in the same order


@pytest.mark.parametrize("first", [True, False])
def test_step_wgrad_over_the_live_rows_of_a_fixed_stride_batch(first):
    """A batch with one molecule per 128-row tile (bmp.packed.StaticPairBatch): bmp_type_rows_live's fifth list names the rows
    of a molecule, the step's weight-gradient launch walks it for the gate blocks -- the same o1 / o2 / dUcT / cs as the launch
    over all rows, whose other rows hold zeros in gda."""
    from bmp import _lib, packed, synth
    from bmp._lib import check, ptr, stream
    from parity_util import close
    L = _lib.lib()
    dev = torch.device("cuda:0")
    store = synth.make_store(60, seed=4, n_lo=2, n_hi=100, n_mean=22)
    ds = packed.DeviceMolStore(packed.MolStore(store), dev)
    rs = np.random.RandomState(2)
    sb = packed.StaticPairBatch(ds, 24)
    sb.load([rs.randint(0, 60, 24), rs.randint(0, 60, 24)], np.zeros((24, 1), np.int32)); sb.emit()
    pb = sb.pb
    N, d = pb.n_rows, 128
    idx, cnt = pb.type_rows_T()
    lvi, lvc = pb._cache["live_rows"]
    torch.cuda.synchronize()
    rm = pb.row_mol.cpu().numpy()
    assert int(lvc[0]) == int((rm >= 0).sum()) and np.array_equal(lvi[:int(lvc[0])].cpu().numpy(), np.nonzero(rm >= 0)[0])
    assert int(lvc[0]) < N // 2                                   # most rows belong to no molecule
    g = torch.Generator().manual_seed(3 + int(first))
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    live = torch.from_numpy((rm >= 0).astype(np.float32)).to(dev)[:, None]
    h, m, rz = rnd(N, d), rnd(N, d), torch.rand(N, 2 * d, generator=g).to(dev)
    gda = rnd(N, 7 * d) * 1e-2 * live                             # what the backward leaves: zeros in the rows of no molecule
    mask = torch.zeros(4, N, device=dev)
    for e in range(4):
        mask[e, idx[e * N: e * N + int(cnt[e])].long()] = 1.0
        gda[:, e * d:(e + 1) * d] *= mask[e][:, None]
    out = []
    for use_live in (False, True):
        o1, o2, dU, cs = (torch.full(s_, 7.0, device=dev) for s_ in ((d, 7 * d), (d, 3 * d), (d, d), (7 * d,)))
        nws = L.bmp_ggnn_step_wgrad_ws_floats(N, d)
        ws = torch.empty(nws, device=dev)
        check(L.bmp_ggnn_step_wgrad(ptr(h), ptr(m), ptr(rz), ptr(gda), N, d, int(first), ptr(o1), ptr(o2), ptr(dU), ptr(cs), 0,
                                    ptr(idx), ptr(cnt), ptr(lvi if use_live else None), ptr(lvc if use_live else None), ptr(ws), nws,
                                    stream()), "wgrad")
        torch.cuda.synchronize()
        out.append((o1.clone(), o2.clone(), dU.clone(), cs.clone()))
    ref = h.double().t() @ gda.double()
    if first:
        ref[:, 4 * d:5 * d] = 0.0
    close(out[1][0], ref, f"o1 over the live rows vs float64 (first {first})", tol=2e-5)
    for name, a_, b_ in zip(("o1", "o2", "dUcT", "cs"), out[0], out[1]):
        close(b_, a_.double(), f"{name} over the live rows vs all rows", tol=2e-5, floor=1e-3 * float(a_.abs().max()) + 1e-12)


def test_lists_off_switch_reaches_the_library():
    """BMP_WGRAD_LISTS=0 (bench.py's A/B): the planned step passes no lists and still equals the oracle (the all-rows launches are
    what rounds 1-3 tested); checked in a child process because the switch is read at import."""
    code = ("import sys; sys.path[:0] = [%r, %r, %r]\n"
            "from bmp import functional as Fn\n"
            "assert Fn._WGRAD_LISTS is False and Fn.type_rows(None) == (None, None)\n"
            "import pytest; sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', %r, '-k', 'fused_step_fwd_bwd and 128']))\n"
            % (ROOT, os.path.join(ROOT, "gcn-bmp_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "test_gpu_ops.py")))
    env = dict(os.environ, BMP_WGRAD_LISTS="0", BMP_PARITY_CHILD="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("encoder", ["ggnn", "relgcn"])
def test_backward_that_skips_zero_blocks_equals_the_one_that_writes_them(encoder):
    """With the row lists on, the backward tile kernels do not write the G_e block of a row without a bond of type e
    (skip_zero_g) and the weight-gradient launch reads those blocks through the lists only.  The same encoder step by step with
    the lists switched off (every block written, every row read) must give the same gradients -- also when the memory the
    unwritten blocks land in is full of NaNs."""
    from bmp import functional as Fn
    from bmp.ggnn import GGNN
    from bmp.relgcn import RelGCN
    from parity_util import close
    pb = _batch(160, seed=9)
    dev = pb.device
    d = 128
    assert Fn._lib.lib().bmp_step_wgrad_lists_used(pb.n_rows, d) == 1
    torch.manual_seed(5)
    enc = (GGNN(out_dim=d, hidden_dim=d, n_layers=3) if encoder == "ggnn" else RelGCN(out_channels=d, ch_list=[d] * 4, scale_adj=True)).to(dev)
    cw = torch.randn(pb.n_mols, d, device=dev)
    res = []
    saved = Fn._WGRAD_LISTS
    try:
        for lists in (True, False):
            Fn._WGRAD_LISTS = lists
            enc.zero_grad()
            poison = torch.full((pb.n_rows, 7 * d), float("nan"), device=dev)        # what torch.empty hands out next
            del poison
            g = enc(pb)
            (g * cw).sum().backward()
            torch.cuda.synchronize()
            res.append({n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None})
    finally:
        Fn._WGRAD_LISTS = saved
    assert res[0].keys() == res[1].keys() and len(res[0]) > 4
    for n in res[0]:
        assert torch.isfinite(res[0][n]).all(), n
        close(res[0][n], res[1][n].double(), f"{encoder} grad {n}: lists + skipped blocks vs all rows", tol=2e-5,
              floor=1e-3 * float(res[1][n].abs().max()) + 1e-12)

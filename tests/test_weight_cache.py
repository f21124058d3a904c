"""The per-call cache of packed weight copies (bmp/functional.py:_cached) is keyed by storage address, so an entry has to
keep its source tensor alive: under no_grad nothing else holds a layer's kernel-layout weights once the next layer's are
bound, and the allocator hands the freed block to the next layer -- whose lookup would then answer with the previous
layer's packed copy (seen as an intermittent 3e-2 error of predict() with untied weights)."""
import gc
import weakref

import numpy as np
import pytest
import torch


def test_cache_entry_keeps_its_source_alive():
    from bmp.functional import _cached
    cache = {}
    src = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    alive = weakref.ref(src)
    out = _cached(cache, "f", src, lambda: src * 2)
    assert _cached(cache, "f", src, lambda: None) is out          # second lookup: the stored copy
    assert _cached(cache, "b", src, lambda: src * 3) is not out   # the tag is part of the key
    del src
    gc.collect()
    assert alive() is not None                                    # the block cannot be re-issued while the cache lives
    cache.clear()
    gc.collect()
    assert alive() is None


def test_no_cache_builds_every_time():
    from bmp.functional import _cached
    src = torch.ones(2, 2)
    assert _cached(None, "f", src, lambda: src + 1) is not _cached(None, "f", src, lambda: src + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [32, 64])
def test_untied_predict_under_no_grad_equals_training_forward(d):
    """Untied layers under no_grad: every step's weights are temporaries.  The allocator is primed so that freed weight
    blocks are re-issued at once (the condition of the stale lookup); predict() must still equal sigmoid(forward)."""
    from oracle import ref_cpu as O
    from bmp import synth, packed
    from bmp.predictor import build_pair_predictor
    from bmp.snapshot import load_param_dict
    from test_gpu_ops import dev, to_dev
    store = synth.make_store(32, seed=9, n_lo=2, n_hi=30, n_mean=12)
    rs = np.random.RandomState(3)
    i1, i2 = rs.randint(0, 32, 16), rs.randint(0, 32, 16)
    pbd = to_dev(packed.pack_from_store(packed.MolStore(store), [i1, i2], device="cpu", with_dense_map=True))
    nl = 5
    p = O.make_pair_params(41, hidden_dim=d, out_dim=d, n_layers=nl, weight_tying=False, attn="nie", dtype=torch.float64)
    model = build_pair_predictor(hidden_dim=d, out_dim=d, n_layers=nl, weight_tying=False, attn="nie").to(dev())
    load_param_dict(model, p)
    want = torch.sigmoid(model(pbd)).detach().clone()
    for _ in range(4):
        torch.cuda.empty_cache()                       # a small pool: frees are re-issued immediately
        got = model.predict(pbd)
        assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()

"""Shared comparison helper of the GPU parity tests: asserts  max|got - want| <= tol * max|want|  and RECORDS the achieved
relative error, so the margin under the north star's 1e-4 is a number and not a guess (What's weak #1b).
Every call appends one line to gpurun_out/parity_errors.jsonl; tests/conftest.py folds the lines into
gpurun_out/parity_summary.json (worst achieved error per test) at the end of the session."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl")


def close(got, want, name, tol=1e-4, floor=1e-6):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = max(want.abs().max().item(), floor)
    err = (got - want).abs().max().item() if got.numel() else 0.0
    rel = err / scale
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    print(f"[parity] {name}: rel err {rel:.3e} (abs {err:.3e}, scale {scale:.3e}, tol {tol:.0e})")
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps(dict(test=test, name=name, rel=rel, abs=err, scale=scale, tol=tol)) + "\n")
    except OSError:
        pass
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    assert rel <= tol, f"{name}: rel err {rel:.3e} > {tol:.0e} (abs {err:.3e}, scale {scale:.3e})"
    return rel


def summarize():
    """worst achieved relative error per test -> gpurun_out/parity_summary.json"""
    if not os.path.exists(LOG):
        return None
    worst = {}
    for ln in open(LOG):
        try:
            r = json.loads(ln)
        except ValueError:
            continue
        w = worst.get(r["test"])
        if w is None or r["rel"] > w["rel"]:
            worst[r["test"]] = dict(rel=r["rel"], name=r["name"], tol=r["tol"])
    out = os.path.join(os.path.dirname(LOG), "parity_summary.json")
    with open(out, "w") as f:
        json.dump(dict(sorted(worst.items())), f, indent=1)
    return worst

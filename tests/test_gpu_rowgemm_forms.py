"""The row GEMM's two forms (weights straight from L2 / weights staged through LDS) must agree: the launcher picks the
LDS form only for big launches, which the operator tests (a handful of tiles) never reach."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, form):
    """form: "db" (default launcher choice for big launches: both operands double-buffered in LDS, 32-deep chunks), "lds"
    (single-buffered 64-deep chunks, BMP_ROWGEMM_FORM=1) or "direct" (weights straight from L2, BMP_ROWGEMM_DIRECT=1)."""
    out = str(tmp_path / f"{form}.pt")
    env = dict(os.environ)
    env.pop("BMP_ROWGEMM_DIRECT", None); env.pop("BMP_ROWGEMM_FORM", None)
    if form == "direct":
        env["BMP_ROWGEMM_DIRECT"] = "1"
    elif form == "lds":
        env["BMP_ROWGEMM_FORM"] = "1"
    subprocess.run([sys.executable, os.path.join(HERE, "rowgemm_forms_worker.py"), out], check=True, env=env, timeout=300)
    return torch.load(out, weights_only=True)


def test_lds_forms_equal_direct_form(tmp_path):
    b = _run(tmp_path, "direct")
    for form in ("db", "lds"):
        a = _run(tmp_path, form)
        assert set(a) == set(b)
        for key in a:
            for k, (x, y) in enumerate(zip(a[key], b[key])):
                scale = max(float(y.abs().max()), 1e-6)
                err = float((x - y).abs().max())
                assert err <= 2e-5 * scale, f"{form} {key}[{k}]: {err:.3e} vs scale {scale:.3e}"

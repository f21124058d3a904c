"""The row GEMM's two forms (weights straight from L2 / weights staged through LDS) must agree: the launcher picks the
LDS form only for big launches, which the operator tests (a handful of tiles) never reach."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, direct):
    out = str(tmp_path / ("direct.pt" if direct else "lds.pt"))
    env = dict(os.environ)
    env.pop("BMP_ROWGEMM_DIRECT", None)
    if direct:
        env["BMP_ROWGEMM_DIRECT"] = "1"
    subprocess.run([sys.executable, os.path.join(HERE, "rowgemm_forms_worker.py"), out], check=True, env=env, timeout=300)
    return torch.load(out, weights_only=True)


def test_lds_form_equals_direct_form(tmp_path):
    a, b = _run(tmp_path, False), _run(tmp_path, True)
    assert set(a) == set(b)
    for key in a:
        for k, (x, y) in enumerate(zip(a[key], b[key])):
            scale = max(float(y.abs().max()), 1e-6)
            err = float((x - y).abs().max())
            assert err <= 2e-5 * scale, f"{key}[{k}]: {err:.3e} vs scale {scale:.3e}"

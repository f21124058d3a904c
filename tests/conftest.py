import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-bmp_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_finish(session):
    # achieved-error log of tests/parity_util.py: a fresh file for every session that RUNS gpu-marked tests; a CPU-only
    # session (-m "not gpu") logs nothing and must leave the last GPU session's log and summary alone
    # (a test may start a child pytest session -- tests/test_gpu_type_rows.py --: the child appends to the parent's log and
    #  leaves rotation and summary to the parent: BMP_PARITY_CHILD=1)
    session.config._bmp_gpu_session = (any(it.get_closest_marker("gpu") is not None for it in session.items)
                                       and os.environ.get("BMP_PARITY_CHILD") != "1")
    if session.config._bmp_gpu_session:
        for name in ("parity_errors.jsonl", "parity_summary.json"):
            try:
                os.remove(os.path.join(ROOT, "gpurun_out", name))
            except OSError:
                pass


def pytest_sessionfinish(session, exitstatus):
    if not getattr(session.config, "_bmp_gpu_session", False):
        return
    try:
        import parity_util
        parity_util.summarize()
    except Exception:
        pass

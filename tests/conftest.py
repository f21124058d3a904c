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


def pytest_sessionstart(session):
    # achieved-error log of tests/parity_util.py: one file per session
    log = os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl")
    try:
        os.remove(log)
    except OSError:
        pass


def pytest_sessionfinish(session, exitstatus):
    try:
        import parity_util
        parity_util.summarize()
    except Exception:
        pass

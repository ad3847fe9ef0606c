import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """MANTIS_CHECK_REPORT_DIR=<dir>: the per-case gradient-parity table of the whole-step checks that ran (profiles/r05_grad_parity.md)."""
    import os
    d = os.environ.get("MANTIS_CHECK_REPORT_DIR")
    if not d:
        return
    try:
        from tests import helpers
        if helpers.GRAD_REPORTS:
            helpers.write_grad_parity(os.path.join(d, "grad_parity.md"))
    except Exception:
        pass

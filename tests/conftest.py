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
    """Dump every parity error the GPU tests measured (tests/parity_bars.py) next to the other run outputs."""
    try:
        from tests import parity_bars
    except ImportError:
        return
    if not parity_bars.RECORDS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_measured.json"), "w") as f:
        json.dump(dict(sorted(parity_bars.RECORDS.items())), f, indent=1)

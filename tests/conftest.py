import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Achieved parity errors of the GPU tests (helpers.record) -> gpurun_out/parity_errors.json."""
    try:
        import helpers
        if not helpers.RECORD:
            return
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump({"exitstatus": int(exitstatus), "records": helpers.RECORD}, f, indent=1)
    except Exception as e:      # never turn a reporting problem into a test failure
        print("parity record not written: %r" % (e,))

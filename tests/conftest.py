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


# ---------------------------------------------------------------------------------------------------------------------
# Every GPU parity test runs twice: in the library's default arithmetic (fp32 MFMA, v_mfma_f32_16x16x4_f32) and in mode 2, "f32 via
# bf16x3" (dpmn_set_compute_dtype(2): fp32 products as six bf16 MFMAs of an exact three-term operand split in the kernels that have
# the variant) -- same tests, same tolerances.  Modules that switch the mode themselves, spawn their own processes or hold no MFMA
# kernel with a variant run once.  DPMN_TEST_MODES=f32 | x3 | f32,x3 restricts the passes.
# (test_gpu_xred.py: the in-L2 split-K reduction is an opt-in fp32 experiment whose tests assert WHICH kernels ran)
X3_EXEMPT = ("test_gpu_x3.py", "test_gpu_bf16.py", "test_gpu_multirank.py", "test_gpu_dataset.py", "test_gpu_stn.py", "test_gpu_dp.py", "test_gpu_xred.py")


def _modes():
    m = [t for t in os.environ.get("DPMN_TEST_MODES", "f32,x3").split(",") if t in ("f32", "x3")]
    return m or ["f32"]


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("gpu") is None:
        return
    if os.path.basename(str(metafunc.definition.fspath)) in X3_EXEMPT:
        return
    metafunc.parametrize("_compute_mode", _modes(), indirect=True, scope="function")


@pytest.fixture(autouse=True)      # autouse: in every test's fixture closure (a name appended in pytest_generate_tests is pruned again), a no-op
def _compute_mode(request):        # unless the test was parametrised above; tests/test_gpu_abi.py checks that the [x3] pass runs in mode 2
    mode = getattr(request, "param", "f32")
    if mode == "f32":
        yield mode
        return
    from dpmn_amd import _abi
    _abi.check(_abi.lib.dpmn_set_compute_dtype(2))
    try:
        yield mode          # (helpers.record labels the rows of this pass "x3:")
    finally:
        _abi.lib.dpmn_set_compute_dtype(0)


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

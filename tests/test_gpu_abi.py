"""GPU-side checks of C-ABI test hooks (device self-tests)."""
import pytest


@pytest.mark.gpu
def test_lane_exchange_selftest():
    """common.h xshfl<O> (DPP quad_perm / row shifts under bank masks / row_ror, v_permlane16_swap, v_permlane32_swap) returns exactly
    __shfl_xor's value for every lane and offset, on 1-D and 2-D blocks, and wave_sum built on it is bitwise the ds_bpermute butterfly."""
    import ctypes
    from dpmn_amd._abi import lib, check
    n = ctypes.c_uint(12345)
    check(lib.dpmn_selftest_xshfl(ctypes.byref(n)))
    assert n.value == 0


@pytest.mark.gpu
def test_compute_mode_pass_sets_the_library_mode(request):
    """tests/conftest.py runs every GPU test twice: the [x3] instance must really execute in mode 2 (dpmn_set_compute_dtype(2)), the
    [f32] instance in mode 0 -- a parametrisation that silently ran both in fp32 would double the suite for nothing."""
    from dpmn_amd import _abi
    mode = request.node.callspec.params.get("_compute_mode") if hasattr(request.node, "callspec") else "f32"
    assert _abi.lib.dpmn_get_compute_dtype() == (2 if mode == "x3" else 0), mode

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

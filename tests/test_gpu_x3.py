""""f32 via bf16x3" (dpmn_set_compute_dtype(2)): fp32 products computed as six bf16 MFMAs of an exact three-term operand split.
The bar is fp32 itself: against a float64 reference on the same fp32 inputs, the x3 kernel may not be further away than the
fp32-MFMA kernel by more than a small factor, and the two kernels agree to fp32 round-off.  Never the default: mode 0 is."""
import pytest
import torch

from dpmn_amd.utils import synth
from helpers import record

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


@pytest.fixture
def x3_mode():
    from dpmn_amd import _abi

    class Mode:
        def __enter__(self):
            _abi.check(_abi.lib.dpmn_set_compute_dtype(2))

        def __exit__(self, *exc):
            _abi.check(_abi.lib.dpmn_set_compute_dtype(0))
    yield Mode()
    _abi.lib.dpmn_set_compute_dtype(0)


def u(name, shape, lo=-1.0, hi=1.0):
    return synth.uniform(name, shape, lo, hi, 91)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def test_mode_switch_roundtrip():
    from dpmn_amd import _abi
    for m in (2, 1, 0):
        _abi.check(_abi.lib.dpmn_set_compute_dtype(m))
        assert _abi.lib.dpmn_get_compute_dtype() == m
    assert _abi.lib.dpmn_set_compute_dtype(3) != 0
    assert _abi.lib.dpmn_get_compute_dtype() == 0


@pytest.mark.parametrize("B,Ch,L", [(3, 384, 1024), (2, 768, 4096), (1, 128, 128)])
def test_pointwise_gemm_x3_is_fp32_class(x3_mode, B, Ch, L):
    """pgrm.py:37 pointwise conv on the raw (B, Ch, L) view: x3 vs fp32-MFMA vs float64."""
    from dpmn_amd import ops
    g, w, b = u("g", (B, L, Ch), -2, 2).to(dev), u("w", (Ch, Ch), -0.3, 0.3).to(dev), u("b", (Ch,)).to(dev)
    ref32 = ops.pointwise(g, w, b)
    with x3_mode:
        got = ops.pointwise(g, w, b)
    ref64 = (w.double() @ g.reshape(B, Ch, L).double() + b.double()[None, :, None]).reshape(B, L, Ch)
    e32, e3, d = rel(ref32, ref64), rel(got, ref64), rel(got, ref32)
    tag = "x3_pointwise_Ch%d" % Ch
    record(tag, "fp32-MFMA kernel rel L2 vs float64", e32)
    record(tag, "bf16x3 kernel rel L2 vs float64", e3, 2.0 * e32)
    record(tag, "bf16x3 vs fp32-MFMA kernel rel L2", d, 1e-6)
    assert e3 <= 2.0 * e32, "bf16x3 pointwise is further from float64 (%.2e) than twice the fp32 kernel (%.2e)" % (e3, e32)
    assert d < 1e-6
    # operands that hit the split's corners: exact bf16 values, values needing all three terms, tiny and huge magnitudes, zeros
    g2 = g.clone()
    g2[..., 0::4] = g2[..., 0::4].bfloat16().float()
    g2[..., 1::4] *= 1e-30
    g2[..., 2::4] *= 1e+20
    g2[0, :7] = 0.0
    ref32 = ops.pointwise(g2, w, b)
    with x3_mode:
        got = ops.pointwise(g2, w, b)
    assert torch.isfinite(got).all()
    ref64 = (w.double() @ g2.reshape(B, Ch, L).double() + b.double()[None, :, None]).reshape(B, L, Ch)
    assert rel(got, ref64) <= 2.0 * rel(ref32, ref64) + 1e-9

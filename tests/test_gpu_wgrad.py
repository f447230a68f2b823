"""GPU parity of the convolution weight gradient (csrc/conv_bwd.hip: k_conv_wgrad + pack_tile<UNPACK>) against torch autograd,
in both reduction modes: exclusive slots (default: plain stores, one copy per pixel split, deterministic) and slotted atomics.
Shapes: the EncodeBlock / DecodeBlock convs of the reference's cmm.py:40-75 at small spatial sizes."""
import pytest
import torch
import torch.nn.functional as F

from dpmn_amd import ops
from dpmn_amd.train import pgrm_train
from dpmn_amd.utils import synth
from helpers import record

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _wgrad(x_nhwc, dy_nhwc, wshape, k, stride, pad, dil, layout="conv"):
    cout = wshape[1] if layout == "convT_s1" else wshape[0]
    d = ops.conv_desc([x_nhwc], k, stride, pad, dil, cout=cout)
    dw = torch.zeros(*wshape, device=x_nhwc.device)
    pgrm_train.conv_wgrad_into(d, dy_nhwc, dw, layout)
    return dw


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,B,H,W", [
    (64, 64, 4, 2, 3, 2, 4, 32, 128),     # EncodeBlock first conv (cmm.py:44): 1 tile column, many pixel splits
    (64, 128, 3, 1, 1, 1, 4, 16, 64),     # EncodeBlock second conv (cmm.py:49)
    (256, 512, 3, 1, 1, 1, 6, 4, 16),     # deep level: many tiles, few splits
    (8, 4, 3, 1, 1, 1, 2, 32, 128),       # DistillModule conv (distill_module.py:9): advised to the slotted-atomic path
    (96, 12, 3, 1, 1, 1, 2, 16, 64),      # narrow output, 16-row tile
])
def test_conv_wgrad_matches_autograd_and_is_deterministic(dev, cin, cout, k, stride, pad, dil, B, H, W, monkeypatch):
    x = synth.uniform("wg.x", (B, cin, H, W), -1, 1, 3).to(dev)
    w = synth.uniform("wg.w", (cout, cin, k, k), -0.1, 0.1, 4).to(dev).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad, dil)
    dy = synth.uniform("wg.dy", tuple(y.shape), -1, 1, 5).to(dev)
    y.backward(dy)
    ref = w.grad
    xn, dyn = x.permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
    got = {}
    for mode in ("excl", "atomic"):
        monkeypatch.setattr(pgrm_train, "WGRAD_MODE", mode)
        got[mode] = _wgrad(xn, dyn, w.shape, k, stride, pad, dil)
        err = float((got[mode] - ref).abs().max() / ref.abs().max())
        record("wgrad_%s_%dx%dx%d_k%d" % (mode, cin, cout, B * H * W, k), "max_rel", err, 3e-6)
        assert err < 3e-6, (mode, err)            # fp32 sums of up to 2^15 products in a different order
    monkeypatch.setattr(pgrm_train, "WGRAD_MODE", "excl")
    again = _wgrad(xn, dyn, w.shape, k, stride, pad, dil)
    if cout > 4:      # (the tiny-gradient case falls back to atomics and is exempt)
        assert torch.equal(again, got["excl"]), "exclusive-slot weight gradient must be bitwise reproducible"


def test_conv_transpose_s1_wgrad(dev):
    """DecodeBlock's ConvTranspose2d(3,1,1) (cmm.py:66): flipped taps, (Cin, Cout, KH, KW) parameter layout."""
    B, cin, cout, H, W = 3, 96, 64, 8, 32
    x = synth.uniform("wgt.x", (B, cin, H, W), -1, 1, 6).to(dev)
    w = synth.uniform("wgt.w", (cin, cout, 3, 3), -0.1, 0.1, 7).to(dev).requires_grad_(True)
    y = F.conv_transpose2d(x, w, None, 1, 1)
    dy = synth.uniform("wgt.dy", tuple(y.shape), -1, 1, 8).to(dev)
    y.backward(dy)
    got = _wgrad(x.permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous(), w.shape, 3, 1, 1, 1, "convT_s1")
    err = float((got - w.grad).abs().max() / w.grad.abs().max())
    record("wgrad_convT_s1", "max_rel", err, 3e-6)
    assert err < 3e-6

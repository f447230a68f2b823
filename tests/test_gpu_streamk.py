"""GPU parity of the stream-K implicit-GEMM conv launch (csrc/conv.hip k_conv_igemm_sk): the split-K layers of the CMM
(cmm.py:86-118: deep encoder convs, 3-segment decoder convs, the phase-fused ConvTranspose2d(4,2,1)) as one persistent launch
whose partial tiles are summed by the last workgroup to arrive.  Checked against torch's fp32 conv on the CPU, against the
fixed-split two-launch path, and for run-to-run bitwise equality (the sum order does not depend on the arrival order).
The library uses the launch by default only for the 64-pixel-row layers (1x4 bottleneck maps); DPMN_CONV_SK=2 (read once per
process, hence the child processes) routes every qualifying layer through it."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from dpmn_amd.utils import synth
from helpers import assert_close

pytestmark = pytest.mark.gpu
ATOL, RTOL = 1e-4, 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def u(name, shape, lo=-1.0, hi=1.0, seed=77):
    return synth.uniform(name, shape, lo, hi, seed)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CASES = [
    # segs, cout, k, stride, pad, dil, B, H, W
    ((256,), 512, 4, 2, 1, 1, 8, 8, 32),          # en_4/en_5 first conv: M = 512, K = 4096, 16 tiles
    ((128,), 128, 4, 2, 3, 2, 6, 16, 64),         # en_3 first conv (dilated, cmm.py:44): M = 1536, K = 2048
    ((256, 256, 256), 128, 3, 1, 1, 1, 6, 4, 16),  # de_3-like 3-segment decoder conv: M = 384, K = 6912
    ((512,), 512, 4, 2, 1, 1, 40, 2, 8),          # en_6: M = 160 ... a ragged second row tile
    ((512,), 512, 3, 1, 1, 1, 48, 2, 8),          # en_5 second conv at the bench batch: M = 768
    ((160,), 136, 3, 1, 2, 2, 5, 6, 10),          # ragged everything: Cout not a multiple of 128, odd plane, K padding chunk
]


def _conv_case(dev, segs, cout, k, stride, pad, dil, B, H, W, affine):
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    xs = [u("s%d" % i, (B, c, H, W)) for i, c in enumerate(segs)]
    cin = sum(segs)
    w = u("w", (cout, cin, k, k)) * (1.0 / (cin * k * k) ** 0.5)
    b = u("b", (cout,))
    if affine:
        sc = [u("sc%d" % i, (c,), 0.5, 1.5) for i, c in enumerate(segs)]
        sh = [u("sh%d" % i, (c,), -0.3, 0.3) for i, c in enumerate(segs)]
        xa = torch.cat([x * s_[None, :, None, None] + h[None, :, None, None] for x, s_, h in zip(xs, sc, sh)], 1)
        aff = [(s_.to(dev), h.to(dev)) for s_, h in zip(sc, sh)]
    else:
        xa, aff = torch.cat(xs, 1), None
    ref = F.conv2d(F.leaky_relu(xa, 0.2), w, b, stride=stride, padding=pad, dilation=dil)
    wp, bp = packing.pack_conv(w.to(dev), b.to(dev))
    xd = [nhwc(x).to(dev) for x in xs]
    run = lambda: ops.conv2d(xd, wp, bp, cout, k, stride=stride, pad=pad, dil=dil, pro_act="leaky02", affine=aff)
    return ref, run


def check_conv_vs_torch_fixed_split_and_itself(dev, segs, cout, k, stride, pad, dil, B, H, W, affine):
    from dpmn_amd import ops, _abi
    ref, run = _conv_case(dev, segs, cout, k, stride, pad, dil, B, H, W, affine)
    _abi.profile_begin(None)
    got = run()
    torch.cuda.synchronize()
    rows = _abi.profile_end()
    assert any(r["kernel"] == "k_conv_igemm_sk" for r in rows), "the stream-K launch did not run: %s" % [r["kernel"] for r in rows]
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "stream-K conv %s" % ((segs, cout, k, stride, pad, dil),))
    for _ in range(3):
        assert torch.equal(run(), got), "stream-K result depends on the arrival order"
    ops.STREAM_K = False
    try:
        old = run()
    finally:
        ops.STREAM_K = True
    # same products, another summation tree over K
    assert_close(got, old, 2e-5, 2e-5, "stream-K vs fixed-split")


def check_bn_statistics_and_counters_left_zero(dev):
    """train-mode conv: the per-channel sum / sum of squares of the epilogue (dpmn_conv_desc.stats) from the fix-up path, and
    the arrival counters are zero again after the launches."""
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B, cin, cout, H, W = 8, 256, 256, 8, 32
    x = u("x", (B, cin, H, W))
    w = u("w", (cout, cin, 4, 4)) * (1.0 / (cin * 16) ** 0.5)
    ref = F.conv2d(F.leaky_relu(x, 0.2), w, None, stride=2, padding=1)
    wp, _ = packing.pack_conv(w.to(dev), None)
    stats = torch.zeros(32, 2, cout, dtype=torch.float64, device=dev)
    got = ops.conv2d([nhwc(x).to(dev)], wp, None, cout, 4, stride=2, pad=1, pro_act="leaky02", stats=stats)
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "conv")
    s = stats.sum(0).cpu()
    assert_close(s[0].float(), ref.double().sum((0, 2, 3)).float(), 1e-3, 1e-4, "sum")
    assert_close(s[1].float(), (ref.double() ** 2).sum((0, 2, 3)).float(), 1e-3, 1e-4, "sum of squares")
    assert all(int(c.abs().max()) == 0 for c in ops._ARRIVE_CNT.values())


def check_phase_fused_conv_transpose(dev, cin, cout, B, H, W):
    from dpmn_amd import ops, _abi
    from dpmn_amd.model import packing
    x = u("x", (B, cin, H, W))
    w4 = u("w4", (cin, cout, 4, 4)) * (1.0 / (cin * 4) ** 0.5)
    b = u("b", (cout,))
    ref = F.conv_transpose2d(F.relu(x), w4, b, stride=2, padding=1)
    packs = ops.stack_phase_packs(packing.pack_convT_s2k4(w4.to(dev), b.to(dev)))
    _abi.profile_begin(None)
    got = ops.convT_s2k4([nhwc(x).to(dev)], packs, cout, pro_act="relu")
    torch.cuda.synchronize()
    rows = _abi.profile_end()
    assert any(r["kernel"] == "k_conv_igemm_sk" for r in rows)
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "convT 4x4 s2")
    assert torch.equal(ops.convT_s2k4([nhwc(x).to(dev)], packs, cout, pro_act="relu"), got)


def test_streamk_two_groups_with_64_pixel_row_tiles(dev):
    """en_6 of the twin encoder branches at the bench batch (cmm.py:93): 192 pixels per branch -- 64-pixel row tiles, one launch."""
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B, cin, cout = 96, 512, 512
    x = u("gx", (B, cin, 2, 8))
    ws = [u("gw%d" % g, (cout, cin, 4, 4)) * (1.0 / (cin * 16) ** 0.5) for g in range(2)]
    bs = [u("gb%d" % g, (cout,)) for g in range(2)]
    h = B // 2
    ref = torch.cat([F.conv2d(F.leaky_relu(x[g * h:(g + 1) * h], 0.2), ws[g], bs[g], stride=2, padding=1) for g in range(2)], 0)
    packs = [packing.pack_conv(ws[g].to(dev), bs[g].to(dev)) for g in range(2)]
    wp = torch.stack([p[0] for p in packs]).contiguous()
    bp = torch.stack([p[1] for p in packs]).contiguous()
    got = ops.conv2d([nhwc(x).to(dev)], wp, bp, cout, 4, stride=2, pad=1, pro_act="leaky02", groups=2)
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "grouped en_6")
    ops.STREAM_K = False
    try:
        old = ops.conv2d([nhwc(x).to(dev)], wp, bp, cout, 4, stride=2, pad=1, pro_act="leaky02", groups=2)   # two launches in the library
    finally:
        ops.STREAM_K = True
    assert_close(got, old, 2e-5, 2e-5, "grouped: stream-K vs one launch per half")


def test_streamk_default_dispatch_phase_fused_bottleneck(dev):
    """de_6 at the bench batch (cmm.py:100-102): 192 pixels per phase -> 64-pixel row tiles, stream-K by default."""
    check_phase_fused_conv_transpose(dev, 1024, 512, 48, 1, 4)


_CHILD = r"""
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_streamk as m
dev = torch.device("cuda:0")
for affine in (False, True):
    for case in m.CASES:
        m.check_conv_vs_torch_fixed_split_and_itself(dev, *case, affine)
m.check_bn_statistics_and_counters_left_zero(dev)
for shape in ((128, 128, 4, 8, 32), (1024, 512, 48, 1, 4), (256, 256, 6, 4, 16)):
    m.check_phase_fused_conv_transpose(dev, *shape)
print("ok")
"""


@pytest.mark.parametrize("blocks", ["", "37", "32", "1000"])
def test_streamk_every_layer_and_other_workgroup_counts(blocks):
    """DPMN_CONV_SK=2: every qualifying layer (deep encoder convs, 3-segment decoder convs, phase-fused transposed convs, affine
    on load, BatchNorm statistics) through the stream-K launch.  DPMN_SK_BLOCKS: ranges that cut tiles in other places -- dozens
    of contributors per tile (1000), odd counts without the XCD remap (37), whole tiles only where the count divides (32)."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, DPMN_CONV_SK="2")
    if blocks:
        env["DPMN_SK_BLOCKS"] = blocks
    r = subprocess.run([sys.executable, "-c", _CHILD % (os.path.dirname(here), here)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

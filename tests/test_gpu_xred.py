"""GPU parity of the in-launch split-K reduction of the implicit-GEMM conv (csrc/conv.hip XRED, k_conv_igemm_xr): every k split
of a 128 x 128 output tile runs on ONE XCD, the partial tiles stay in that XCD's L2 and the last workgroup to arrive adds them in
split order and runs the epilogue -- no k_conv_splitk_reduce launch (cmm.py:86-118: the deep CMM levels).  Checked against torch's
fp32 conv on the CPU, against the two-launch path, for run-to-run bitwise equality, and through the recompute path a misplaced
tile would take (forced by the test hook: must be bitwise equal to the fast path)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close
from test_gpu_streamk import _conv_case, u, nhwc

pytestmark = pytest.mark.gpu
ATOL, RTOL = 1e-4, 1e-4


@pytest.fixture(scope="module")
def dev():
    """The in-launch reduction is off by default (the reduce launch measured faster on MI355X); these tests switch it on."""
    from dpmn_amd import _abi
    assert torch.cuda.is_available()
    _abi.check(_abi.lib.dpmn_xred_enable(1))
    yield torch.device("cuda:0")
    _abi.check(_abi.lib.dpmn_xred_enable(-1))


def fallbacks(reset=True):
    from dpmn_amd import _abi
    n = C.c_uint(0)
    _abi.check(_abi.lib.dpmn_xred_fallbacks(C.byref(n), 1 if reset else 0))
    return n.value


def profiled(run):
    from dpmn_amd import _abi
    _abi.profile_begin(None)
    out = run()
    torch.cuda.synchronize()
    return out, [r["kernel"] for r in _abi.profile_end()]


def check(dev, run, ref_nchw, what):
    """ref, no reduce launch, bitwise repeatable, equal to the two-launch path up to the summation tree, and the recompute
    path bitwise equal to the fast one."""
    from dpmn_amd import ops, _abi
    fallbacks()
    got, kernels = profiled(run)
    assert "k_conv_igemm<128,128>" in kernels and "k_conv_splitk_reduce" not in kernels, kernels
    assert_close(got.permute(0, 3, 1, 2), ref_nchw, ATOL, RTOL, what)
    for _ in range(3):
        assert torch.equal(run(), got), "result depends on the arrival order: " + what
    assert fallbacks() == 0, "workgroups of one tile ran on different XCDs (the placement the fast path relies on does not hold)"
    ops.STREAM_K = False      # no arrival counters -> the fixed split with its reduce launch
    try:
        old, k_old = profiled(run)
    finally:
        ops.STREAM_K = True
    assert "k_conv_splitk_reduce" in k_old, k_old
    assert_close(got, old, 2e-5, 2e-5, "in-launch reduction vs reduce launch: " + what)
    _abi.check(_abi.lib.dpmn_xred_test_force_recompute(1))
    try:
        redo = run()
        torch.cuda.synchronize()
    finally:
        _abi.check(_abi.lib.dpmn_xred_test_force_recompute(0))
    assert fallbacks() > 0
    assert torch.equal(redo, got), "recompute path differs from the collect path: " + what
    assert all(int(c.abs().max()) == 0 for c in ops._ARRIVE_CNT.values()), "arrival words not left at zero"


CASES = [
    # segs, cout, k, stride, pad, dil, B, H, W
    ((256,), 512, 4, 2, 1, 1, 8, 8, 32),          # M = 512, K = 4096: 16 tiles, 16 splits
    ((128,), 128, 4, 2, 3, 2, 6, 16, 64),         # en_3 first conv (dilated): 12 tiles -> two XCDs stay empty
    ((256, 256, 256), 128, 3, 1, 1, 1, 6, 4, 16),  # 3-segment decoder conv: 3 tiles, 27 splits
    ((512,), 512, 4, 2, 1, 1, 40, 2, 8),          # M = 160: a ragged second row tile
    ((512,), 512, 3, 1, 1, 1, 48, 2, 8),          # en_5 second conv at the bench batch: 24 tiles, 18 splits
    ((160,), 136, 3, 1, 2, 2, 5, 6, 10),          # Cout not a multiple of 128, odd plane, a K padding chunk
    ((128,), 256, 3, 1, 1, 1, 16, 4, 16),         # 8 row tiles x 2 column tiles: the column-tile-fastest order
]


@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_xred_conv_vs_torch_two_launch_path_and_recompute(dev, case, affine):
    ref, run = _conv_case(dev, *case, affine)
    check(dev, run, ref, "conv %s affine=%s" % (case, affine))


def test_xred_tile_orders_agree(dev, monkeypatch):
    """DPMN_XRED_ORDER is read once per process, so both orders cannot be forced here; the order only permutes which XCD owns a
    tile -- the result of a tile does not depend on it.  Covered: the default choice on a layer where each order wins (cases 0
    and 6 above).  This test pins the wide-N shape against torch at a second batch size."""
    ref, run = _conv_case(dev, (128,), 384, 3, 1, 1, 1, 24, 4, 16, False)
    check(dev, run, ref, "3 column tiles")


def test_xred_two_groups(dev):
    """the CMM's twin encoder branches in one launch (cmm.py:86-99): images [B/2, B) use the second weight set."""
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B, cin, cout = 8, 256, 256
    x = u("gx", (B, cin, 8, 32))
    ws = [u("gw%d" % g, (cout, cin, 4, 4)) * (1.0 / (cin * 16) ** 0.5) for g in range(2)]
    bs = [u("gb%d" % g, (cout,)) for g in range(2)]
    h = B // 2
    ref = torch.cat([F.conv2d(F.leaky_relu(x[g * h:(g + 1) * h], 0.2), ws[g], bs[g], stride=2, padding=3, dilation=2) for g in range(2)], 0)
    packs = [packing.pack_conv(ws[g].to(dev), bs[g].to(dev)) for g in range(2)]
    wp = torch.stack([p[0] for p in packs]).contiguous()
    bp = torch.stack([p[1] for p in packs]).contiguous()
    xd = nhwc(x).to(dev)
    check(dev, lambda: ops.conv2d([xd], wp, bp, cout, 4, stride=2, pad=3, dil=2, pro_act="leaky02", groups=2), ref, "grouped en_4a")


@pytest.mark.parametrize("shape", [(256, 256, 6, 4, 16), (512, 512, 48, 2, 8)])
def test_xred_phase_fused_conv_transpose(dev, shape):
    """nn.ConvTranspose2d(4,2,1) (cmm.py:100-118) as four phases in one launch: tiles = row tiles x column tiles x phases."""
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    cin, cout, B, H, W = shape
    x = u("x", (B, cin, H, W))
    w4 = u("w4", (cin, cout, 4, 4)) * (1.0 / (cin * 4) ** 0.5)
    b = u("b", (cout,))
    ref = F.conv_transpose2d(F.relu(x), w4, b, stride=2, padding=1)
    packs = ops.stack_phase_packs(packing.pack_convT_s2k4(w4.to(dev), b.to(dev)))
    xd = nhwc(x).to(dev)
    check(dev, lambda: ops.convT_s2k4([xd], packs, cout, pro_act="relu"), ref, "convT 4x4 s2 %s" % (shape,))


def test_xred_batchnorm_statistics(dev):
    """train-mode conv: per-channel sum / sum of squares from the last-arriving block's epilogue."""
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B, cin, cout, H, W = 16, 256, 256, 8, 32
    x = u("x", (B, cin, H, W))
    w = u("w", (cout, cin, 4, 4)) * (1.0 / (cin * 16) ** 0.5)
    ref = F.conv2d(F.leaky_relu(x, 0.2), w, None, stride=2, padding=1)
    wp, _ = packing.pack_conv(w.to(dev), None)
    stats = torch.zeros(32, 2, cout, dtype=torch.float64, device=dev)
    xd = nhwc(x).to(dev)
    got, kernels = profiled(lambda: ops.conv2d([xd], wp, None, cout, 4, stride=2, pad=1, pro_act="leaky02", stats=stats))
    assert "k_conv_splitk_reduce" not in kernels and "k_conv_igemm<128,128>" in kernels, kernels
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "conv")
    s = stats.sum(0).cpu()
    assert_close(s[0].float(), ref.double().sum((0, 2, 3)).float(), 1e-3, 1e-4, "sum")
    assert_close(s[1].float(), (ref.double() ** 2).sum((0, 2, 3)).float(), 1e-3, 1e-4, "sum of squares")

"""GPU parity: NHWC implicit-GEMM conv (all the Conv2d / ConvTranspose2d shapes of cmm.py), the CMM
channel gate and the whole CMM module vs the oracle and the reference golden vectors."""
import pytest
import torch
import torch.nn.functional as F

from dpmn_amd.utils import synth
from helpers import load_golden, sd_from_manifest, t, assert_close

pytestmark = pytest.mark.gpu
ATOL, RTOL = 1e-4, 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def u(name, shape, lo=-1.0, hi=1.0, seed=50):
    return synth.uniform(name, shape, lo, hi, seed)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,H,W", [
    (64, 64, 4, 2, 3, 2, 32, 128),    # EncodeBlock first conv (cmm.py:44)
    (64, 128, 3, 1, 1, 1, 16, 64),    # EncodeBlock second conv (cmm.py:49)
    (8, 16, 3, 1, 1, 1, 16, 64),      # cnum=8 variant: Cin not a multiple of 32
    (512, 512, 4, 2, 1, 1, 2, 8),     # en_6 (cmm.py:93)
    (32, 12, 3, 1, 1, 1, 16, 64),     # small Cout (N masking)
    (64, 256, 3, 1, 1, 1, 16, 64),    # UpsampleBLock conv (tsrn.py:110)
    (4, 64, 9, 1, 4, 1, 16, 64),      # TSRN block1 9x9 (tsrn.py:26), NHWC input padded to 4 channels
    (64, 4, 9, 1, 4, 1, 32, 128),     # TSRN last conv 9x9 (tsrn.py:40)
])
def test_conv2d_matches_torch(dev, cin, cout, k, stride, pad, dil, H, W):
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B = 3
    x = u("x", (B, cin, H, W))
    w = u("w", (cout, cin, k, k), -1, 1) * (1.0 / (cin * k * k) ** 0.5)
    b = u("b", (cout,))
    ref = F.conv2d(F.leaky_relu(x, 0.2), w, b, stride=stride, padding=pad, dilation=dil)
    wp, bp = packing.pack_conv(w.to(dev), b.to(dev))
    got = ops.conv2d([nhwc(x).to(dev)], wp, bp, cout, k, stride=stride, pad=pad, dil=dil, pro_act="leaky02")
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "conv %s" % ((cin, cout, k, stride, pad, dil),))
    got2 = ops.conv2d([nhwc(x).to(dev)], wp, bp, cout, k, stride=stride, pad=pad, dil=dil, pro_act="leaky02", out_nchw=True,
                      epi_act="tanh")
    assert_close(got2, torch.tanh(ref), ATOL, RTOL, "conv nchw+tanh")


@pytest.mark.parametrize("segs,cout,k,stride,pad,dil,H,W", [
    ((32,), 40, 3, 1, 1, 1, 5, 7),            # odd plane, Cout not a multiple of 4*16: borders + the num_records row check
    ((64, 32), 96, 4, 2, 1, 1, 6, 10),        # two segments, stride 2, even kernel
    ((32, 32, 64), 132, 3, 1, 2, 2, 9, 12),   # three segments, dilation 2, padding wider than the kernel radius
    ((96,), 64, 1, 1, 0, 1, 7, 9),            # 1x1
    ((256, 256, 256), 128, 3, 1, 1, 1, 4, 16),  # deep decoder shape: split-K path (M = 192 pixels at B = 3)
])
def test_conv2d_buffer_load_instantiation_borders_segments_affine(dev, segs, cout, k, stride, pad, dil, H, W):
    """Channel counts that are multiples of 32 take the raw-buffer-load implicit GEMM (hardware range check instead of
    predicated loads): odd planes, strides, dilation, 1-3 input segments, affine + activation on load (padding must stay 0)."""
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B = 3
    xs = [u("s%d" % i, (B, c, H, W)) for i, c in enumerate(segs)]
    cin = sum(segs)
    w = u("w", (cout, cin, k, k), -1, 1) * (1.0 / (cin * k * k) ** 0.5)
    b = u("b", (cout,))
    sc = [u("sc%d" % i, (c,), 0.5, 1.5) for i, c in enumerate(segs)]
    sh = [u("sh%d" % i, (c,), -0.3, 0.3) for i, c in enumerate(segs)]
    xa = torch.cat([x * s_[None, :, None, None] + h[None, :, None, None] for x, s_, h in zip(xs, sc, sh)], 1)
    ref = F.conv2d(F.leaky_relu(xa, 0.2), w, b, stride=stride, padding=pad, dilation=dil)
    wp, bp = packing.pack_conv(w.to(dev), b.to(dev))
    got = ops.conv2d([nhwc(x).to(dev) for x in xs], wp, bp, cout, k, stride=stride, pad=pad, dil=dil, pro_act="leaky02",
                     affine=[(s_.to(dev), h.to(dev)) for s_, h in zip(sc, sh)])
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "affine + leaky on load, segs %s" % (segs,))
    ref0 = F.conv2d(torch.cat(xs, 1), w, b, stride=stride, padding=pad, dilation=dil)
    got0 = ops.conv2d([nhwc(x).to(dev) for x in xs], wp, bp, cout, k, stride=stride, pad=pad, dil=dil)
    assert_close(got0.permute(0, 3, 1, 2), ref0, ATOL, RTOL, "plain, segs %s" % (segs,))


def test_conv2d_bn_fold_segments_residual_and_pixelshuffle(dev):
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B, H, W = 4, 8, 32   # M = 1024 pixels: exercises the LDS halo-tile kernel with 3 input segments
    xs = [u("s%d" % i, (B, c, H, W)) for i, c in enumerate((64, 32, 32))]
    cin, cout = 128, 64
    w = u("w", (cout, cin, 3, 3)) * 0.05
    b = u("b", (cout,))
    bn = (u("g", (cout,), 0.5, 1.5), u("be", (cout,)), u("mu", (cout,), -0.2, 0.2), u("var", (cout,), 0.5, 1.5), 1e-5)
    res = u("res", (B, cout, H, W))
    xcat = torch.cat(xs, 1)
    ref = F.batch_norm(F.conv2d(F.relu(xcat), w, b, padding=1), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    wp, bp = packing.pack_conv(w.to(dev), b.to(dev), tuple(v.to(dev) if torch.is_tensor(v) else v for v in bn))
    got = ops.conv2d([nhwc(x).to(dev) for x in xs], wp, bp, cout, 3, pad=1, pro_act="relu", epi_act="mish",
                     res=nhwc(res).to(dev))
    assert_close(got.permute(0, 3, 1, 2), ref * torch.tanh(F.softplus(ref)) + res, ATOL, RTOL, "segments+bn+mish+res")
    # per-segment affine on load (train-mode BN of the producers) + stats accumulation
    sc = [u("sc%d" % i, (c,), 0.5, 1.5) for i, c in enumerate((64, 32, 32))]
    sh = [u("sh%d" % i, (c,), -0.3, 0.3) for i, c in enumerate((64, 32, 32))]
    xa = torch.cat([x * s[None, :, None, None] + h[None, :, None, None] for x, s, h in zip(xs, sc, sh)], 1)
    ref2 = F.conv2d(F.relu(xa), w, b, padding=1)
    wp2, bp2 = packing.pack_conv(w.to(dev), b.to(dev))
    stats = torch.zeros(32, 2, cout, device=dev, dtype=torch.float64)
    got2 = ops.conv2d([nhwc(x).to(dev) for x in xs], wp2, bp2, cout, 3, pad=1, pro_act="relu",
                      affine=[(s.to(dev), h.to(dev)) for s, h in zip(sc, sh)], stats=stats)
    assert_close(got2.permute(0, 3, 1, 2), ref2, ATOL, RTOL, "affine on load")
    assert_close(stats.sum(0)[0].float(), ref2.sum(dim=(0, 2, 3)), 2e-3, 1e-4, "stats sum")
    assert_close(stats.sum(0)[1].float(), (ref2 * ref2).sum(dim=(0, 2, 3)), 2e-3, 1e-4, "stats sumsq")
    # PixelShuffle(2) epilogue (UpsampleBLock, tsrn.py:110-112)
    w4 = u("w4", (256, 64, 3, 3)) * 0.05
    b4 = u("b4", (256,))
    x4 = u("x4", (B, 64, H, W))
    r = F.pixel_shuffle(F.conv2d(x4, w4, b4, padding=1), 2)
    ref3 = r * torch.tanh(F.softplus(r))
    wp4, bp4 = packing.pack_conv(w4.to(dev), b4.to(dev))
    got3 = ops.conv2d([nhwc(x4).to(dev)], wp4, bp4, 256, 3, pad=1, epi_act="mish", pixel_shuffle=True)
    assert_close(got3.permute(0, 3, 1, 2), ref3, ATOL, RTOL, "pixel shuffle + mish")


@pytest.mark.parametrize("cin,cout,H,W", [(64, 32, 4, 16), (1024, 512, 1, 4), (24, 8, 8, 32)])
def test_conv_transpose_matches_torch(dev, cin, cout, H, W):
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B = 2
    x = u("x", (B, cin, H, W))
    w3 = u("w3", (cin, cout, 3, 3)) * (1.0 / (cin * 9) ** 0.5)
    w4 = u("w4", (cin, cout, 4, 4)) * (1.0 / (cin * 4) ** 0.5)
    b = u("b", (cout,))
    xd = nhwc(x).to(dev)
    ref = F.conv_transpose2d(F.relu(x), w3, b, stride=1, padding=1)
    wp, bp = packing.pack_convT_s1(w3.to(dev), b.to(dev))
    got = ops.conv2d([xd], wp, bp, cout, 3, pad=1, pro_act="relu")
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "convT 3x3 s1")
    ref = F.conv_transpose2d(F.relu(x), w4, b, stride=2, padding=1)
    got = ops.convT_s2k4([xd], packing.pack_convT_s2k4(w4.to(dev), b.to(dev)), cout, pro_act="relu")
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "convT 4x4 s2")


def test_layout_helpers(dev):
    from dpmn_amd import ops
    x = u("x", (3, 3, 32, 128))
    y = ops.nchw_to_nhwc(x.to(dev), 4)
    assert_close(y[..., :3], nhwc(x), 0, 0, "nchw->nhwc")
    assert float(y[..., 3].abs().max()) == 0.0
    assert_close(ops.nhwc_to_nchw(y), torch.cat([x, torch.zeros(3, 1, 32, 128)], 1), 0, 0, "nhwc->nchw")


@pytest.mark.parametrize("cnum", [8, 64])
def test_cmm_module_vs_reference_golden(dev, cnum):
    from dpmn_amd.model.cmm import ComplementationModulationModule
    g = load_golden("cmm_cnum%d" % cnum)
    m = ComplementationModulationModule(cnum=cnum).eval()
    sd = m.state_dict()
    man = [str(r).split("|")[0] for r in g["manifest"]]
    assert list(sd.keys()) == man
    synth.synth_fill_(sd, 31)
    m.load_state_dict(sd)
    m = m.to(dev)
    x1 = synth.uniform("cmm_x1", (2, 3, 32, 128), 0, 1, 7).to(dev)
    x2 = synth.uniform("cmm_x2", (2, 3, 32, 128), 0, 1, 7).to(dev)
    with torch.no_grad():
        out = m(x1, x2)
    assert_close(out, t(g["out_eval"]), 2e-4, 2e-4, "CMM eval vs reference golden cnum=%d" % cnum)


@pytest.mark.parametrize("cnum,B,H,W", [(8, 2, 32, 128), (64, 3, 32, 128), (64, 48, 32, 128), (16, 2, 64, 256)])
def test_cmm_native_driver_equals_composed_ops(dev, cnum, B, H, W):
    """dpmn_cmm_forward_f32 (one native call per CMM forward, csrc/cmm_forward.hip) issues the launches of the per-op host path
    (model/cmm.py) with the same arguments: bitwise-equal outputs, also on the second call (cached weights struct / workspace) and
    after the eval pack was rebuilt; 64x256 = the stress configuration's map."""
    from dpmn_amd.model import cmm as cmm_mod
    m = cmm_mod.ComplementationModulationModule(cnum=cnum).eval()
    sd = m.state_dict()
    synth.synth_fill_(sd, 35)
    m.load_state_dict(sd)
    m = m.to(dev)
    x1, x2 = u("nx1", (B, 3, H, W), 0, 1).to(dev), u("nx2", (B, 3, H, W), 0, 1).to(dev)
    assert cmm_mod.NATIVE_FORWARD
    with torch.no_grad():
        nat = m(x1, x2)
        nat2 = m(x1, x2)
        cmm_mod.NATIVE_FORWARD = False
        try:
            ref = m(x1, x2)
        finally:
            cmm_mod.NATIVE_FORWARD = True
        m._pack = None
        nat3 = m(x1, x2)
    assert torch.equal(nat, ref) and torch.equal(nat2, ref) and torch.equal(nat3, ref)


def test_cmm_module_batch48_vs_oracle(dev):
    from dpmn_amd.model.cmm import ComplementationModulationModule
    from oracle import cmm as ocmm
    B = 48
    m = ComplementationModulationModule(cnum=16).eval()
    sd = m.state_dict()
    synth.synth_fill_(sd, 33)
    m.load_state_dict(sd)
    x1, x2 = u("x1", (B, 3, 32, 128), 0, 1), u("x2", (B, 3, 32, 128), 0, 1)
    ref = ocmm.cmm_forward({k: v.clone() for k, v in sd.items()}, x1, x2, False)
    m = m.to(dev)
    with torch.no_grad():
        out = m(x1.to(dev), x2.to(dev))
    assert_close(out, ref, 2e-4, 2e-4, "CMM B=48 vs oracle")


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,B,H,W", [
    (64, 64, 4, 2, 3, 2, 4, 32, 128),     # grouped implicit GEMM, 64x64 tiles (EncodeBlock first conv, cmm.py:44)
    (128, 128, 4, 2, 3, 2, 8, 16, 64),    # grouped 128x128 tiles
    (64, 128, 3, 1, 1, 1, 4, 16, 64),     # grouped halo kernel (the tile's image picks the weights)
    (256, 512, 4, 2, 1, 1, 16, 8, 32),    # grouped split-K (M = 1024 pixels, K = 4096)
    (4, 64, 3, 1, 1, 1, 2, 32, 128),      # en_1: 4 padded input channels (generic loader)
    (512, 512, 4, 2, 1, 1, 6, 2, 8),      # half a batch = 12 pixels: not whole row tiles -> two launches over the halves
])
def test_conv2d_two_groups_equals_two_convs(dev, cin, cout, k, stride, pad, dil, B, H, W):
    """groups = 2 (the twin encoder branches of cmm.py:86-99 in one launch): images [B/2, B) use the second weight set."""
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    x = u("gx", (B, cin, H, W))
    ws = [u("gw%d" % g, (cout, cin, k, k), -1, 1) * (1.0 / (cin * k * k) ** 0.5) for g in range(2)]
    bs = [u("gb%d" % g, (cout,)) for g in range(2)]
    h = B // 2
    ref = torch.cat([F.conv2d(F.leaky_relu(x[g * h:(g + 1) * h], 0.2), ws[g], bs[g], stride=stride, padding=pad, dilation=dil)
                     for g in range(2)], 0)
    packs = [packing.pack_conv(ws[g].to(dev), bs[g].to(dev)) for g in range(2)]
    wp = torch.stack([p[0] for p in packs]).contiguous()
    bp = torch.stack([p[1] for p in packs]).contiguous()
    got = ops.conv2d([nhwc(x).to(dev)], wp, bp, cout, k, stride=stride, pad=pad, dil=dil, pro_act="leaky02", groups=2)
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "grouped conv %s" % ((cin, cout, k, stride, pad, dil),))


@pytest.mark.parametrize("segs,cout,act,affine,stats,B,H,W", [
    ((4, 4), 4, "leaky02", True, True, 2, 32, 128),     # DistillModule conv on cat(deep, shallow) (distill_module.py:9) + BN statistics
    ((4,), 4, "none", False, True, 2, 32, 128),
    ((4,), 12, "relu", False, False, 3, 16, 64),        # three output quads
    ((8, 4, 4), 4, "none", True, False, 2, 16, 64),     # three segments, 16 input channels
    ((4,), 16, "none", False, True, 2, 16, 64),         # four output quads + statistics
    ((8,), 8, "leaky02", False, False, 2, 16, 64),      # two output quads
    ((4,), 3, "none", False, False, 2, 32, 128),        # Cout not a multiple of 4
    ((12,), 12, "relu", False, False, 3, 16, 64),       # PGRM 12 -> 12 tail conv: cin * Cout > 64 stays on the MFMA tiles
])
def test_conv2d_direct_small_channels(dev, segs, cout, act, affine, stats, B, H, W):
    """k_conv_direct (cin * Cout <= 64, one thread per pixel): prologue affine + activation with exact-zero padding,
    conv_store epilogue, block-reduced BatchNorm statistics."""
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    cin = sum(segs)
    xs = [u("dx%d" % i, (B, c, H, W)) for i, c in enumerate(segs)]
    aff = [(u("ds%d" % i, (c,), 0.5, 1.5), u("dh%d" % i, (c,), -0.5, 0.5)) for i, c in enumerate(segs)] if affine else None
    w = u("dw", (cout, cin, 3, 3), -1, 1) * (1.0 / (cin * 9) ** 0.5)
    b = u("db", (cout,))
    pre = torch.cat([x * aff[i][0][None, :, None, None] + aff[i][1][None, :, None, None] if affine else x for i, x in enumerate(xs)], 1)
    pre = {"none": lambda t_: t_, "relu": F.relu, "leaky02": lambda t_: F.leaky_relu(t_, 0.2)}[act](pre)
    ref = F.conv2d(pre, w, b, padding=1)
    wp, bp = packing.pack_conv(w.to(dev), b.to(dev))
    st = torch.zeros(32, 2, cout, device=dev, dtype=torch.float64) if stats else None
    got = ops.conv2d([nhwc(x).to(dev) for x in xs], wp, bp, cout, 3, pad=1, pro_act=act,
                     affine=None if not affine else [(s.to(dev), h.to(dev)) for s, h in aff], stats=st)
    assert_close(got.permute(0, 3, 1, 2), ref, ATOL, RTOL, "direct conv %s" % (segs,))
    if stats:
        s12 = st.sum(0).float().cpu()
        assert_close(s12[0], ref.sum((0, 2, 3)), 2e-2, 1e-4, "stats sum")
        assert_close(s12[1], (ref * ref).sum((0, 2, 3)), 2e-2, 1e-4, "stats sumsq")

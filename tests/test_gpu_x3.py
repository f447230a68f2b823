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


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("segs,cout,k,stride,pad,dil,H,W,B,aff", [
    ((64,), 128, 4, 2, 3, 2, 32, 128, 4, False),       # EncodeBlock first conv (cmm.py:44): implicit GEMM, 128 x 128 tiles
    ((128,), 256, 4, 2, 3, 2, 16, 64, 4, True),        # the same with the BatchNorm affine on load (training forward)
    ((256, 256, 256), 128, 3, 1, 1, 1, 4, 16, 3, False),   # deep decoder conv: split-K partial slabs + reduce launch
    ((256, 256), 256, 3, 1, 1, 1, 2, 8, 6, True),      # split-K with affine, M = 96 (ragged row tile)
    ((64,), 64, 4, 2, 1, 1, 32, 128, 2, False),        # 64 x 64 tiles
    ((32, 32), 96, 4, 2, 1, 1, 6, 10, 3, True),        # 64 x 64 tiles, two segments, odd plane, Cout not a multiple of 64
    ((64,), 64, 3, 1, 1, 1, 16, 64, 4, False),         # halo kernel, 8-row tiles
    ((64, 64), 128, 3, 1, 1, 1, 32, 128, 2, True),     # halo kernel, two segments, affine
    ((32,), 64, 3, 1, 1, 1, 8, 16, 2, False),          # halo kernel, 4-row tiles (few blocks)
    ((96,), 64, 1, 1, 0, 1, 7, 9, 3, False),           # 1 x 1
])
def test_conv_x3_is_fp32_class(x3_mode, segs, cout, k, stride, pad, dil, H, W, B, aff):
    """Implicit-GEMM and halo convs (cmm.py:38-77, tsrn.py:83-110) in mode 2 vs fp32-MFMA vs float64 on the same fp32 operands."""
    import torch.nn.functional as F
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    xs = [u("s%d" % i, (B, c, H, W), -2, 2) for i, c in enumerate(segs)]
    cin = sum(segs)
    w = u("w", (cout, cin, k, k)) * (1.0 / (cin * k * k) ** 0.5)
    b = u("b", (cout,))
    sc = [u("sc%d" % i, (c,), 0.5, 1.5) for i, c in enumerate(segs)]
    sh = [u("sh%d" % i, (c,), -0.3, 0.3) for i, c in enumerate(segs)]
    if aff:      # the kernel's fp32 affine (mul, then add) is part of the operand: reproduce it in fp32, then go to float64
        xa = torch.cat([x * s_[None, :, None, None] + h[None, :, None, None] for x, s_, h in zip(xs, sc, sh)], 1)
    else:
        xa = torch.cat(xs, 1)
    ref64 = F.conv2d(F.leaky_relu(xa, 0.2).double(), w.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    wp, bp = packing.pack_conv(w.to(dev), b.to(dev))
    run = lambda: ops.conv2d([nhwc(x).to(dev) for x in xs], wp, bp, cout, k, stride=stride, pad=pad, dil=dil, pro_act="leaky02",
                             affine=[(s_.to(dev), h.to(dev)) for s_, h in zip(sc, sh)] if aff else None).permute(0, 3, 1, 2)
    ref32 = run()
    with x3_mode:
        got = run()
        got_again = run()
    assert torch.equal(got, got_again)
    e32, e3, d = rel(ref32, ref64), rel(got, ref64), rel(got, ref32)
    tag = "x3_conv_%s_%d_k%d" % ("+".join(map(str, segs)), cout, k)
    record(tag, "fp32-MFMA kernel rel L2 vs float64", e32)
    record(tag, "bf16x3 kernel rel L2 vs float64", e3, 2.0 * e32)
    record(tag, "bf16x3 vs fp32-MFMA kernel rel L2", d, 2e-6)
    assert e3 <= 2.0 * e32 + 1e-8, "bf16x3 conv is further from float64 (%.2e) than twice the fp32 kernel (%.2e)" % (e3, e32)
    assert d < 2e-6


def test_conv_x3_transposed_phases_groups_and_extreme_operands(x3_mode):
    """ConvTranspose2d(4, 2, 1) as four fused phases, the grouped twin-encoder launch, and operands at the corners of the split
    (exact bf16 values, 1e-30 / 1e+20 scales, zeros, near-FLT_MAX finite values): finite results, fp32-class errors."""
    import torch.nn.functional as F
    from dpmn_amd import ops
    from dpmn_amd.model import packing
    B, cin, cout, H, W = 4, 128, 128, 8, 32
    x = u("xt", (B, cin, H, W), -2, 2)
    w4 = u("w4", (cin, cout, 4, 4)) * (1.0 / (cin * 4) ** 0.5)
    b = u("bt", (cout,))
    ref64 = F.conv_transpose2d(F.relu(x).double(), w4.double(), b.double(), stride=2, padding=1)
    xd = nhwc(x).to(dev)
    packs = packing.pack_convT_s2k4(w4.to(dev), b.to(dev))
    ref32 = ops.convT_s2k4([xd], packs, cout, pro_act="relu").permute(0, 3, 1, 2)
    with x3_mode:
        got = ops.convT_s2k4([xd], packs, cout, pro_act="relu").permute(0, 3, 1, 2)
    assert rel(got, ref64) <= 2.0 * rel(ref32, ref64) + 1e-8 and rel(got, ref32) < 2e-6
    # grouped launch: images [B/2, B) use the second weight set
    ws = [u("wg%d" % g, (cout, cin, 4, 4)) * (1.0 / (cin * 16) ** 0.5) for g in range(2)]
    bs = [u("bg%d" % g, (cout,)) for g in range(2)]
    x2 = u("xg", (B, cin, 16, 64), -2, 2)
    pk = [packing.pack_conv(ws[g].to(dev), bs[g].to(dev)) for g in range(2)]
    wp, bp = torch.stack([p[0] for p in pk]).contiguous(), torch.stack([p[1] for p in pk]).contiguous()
    ref64 = torch.cat([F.conv2d(F.leaky_relu(x2[g * 2:(g + 1) * 2], 0.2).double(), ws[g].double(), bs[g].double(), stride=2, padding=1)
                       for g in range(2)], 0)
    run = lambda: ops.conv2d([nhwc(x2).to(dev)], wp, bp, cout, 4, stride=2, pad=1, pro_act="leaky02", groups=2).permute(0, 3, 1, 2)
    ref32 = run()
    with x3_mode:
        got = run()
    assert rel(got, ref64) <= 2.0 * rel(ref32, ref64) + 1e-8 and rel(got, ref32) < 2e-6
    # corners of the split
    x3 = u("xc", (2, 64, 16, 64), -2, 2)
    x3[:, 0::4] = x3[:, 0::4].bfloat16().float()
    x3[:, 1::4] *= 1e-30
    x3[:, 2::4] *= 1e+20
    x3[0, :, :3] = 0.0
    w = u("wc", (64, 64, 3, 3)) * (1.0 / 24.0)
    ref64 = F.conv2d(x3.double(), w.double(), None, padding=1)
    wpc, _ = packing.pack_conv(w.to(dev), None)
    run = lambda: ops.conv2d([nhwc(x3).to(dev)], wpc, None, 64, 3, pad=1).permute(0, 3, 1, 2)
    ref32 = run()
    with x3_mode:
        got = run()
    assert torch.isfinite(got).all()
    assert rel(got, ref64) <= 2.0 * rel(ref32, ref64) + 1e-9
    # finite operands next to FLT_MAX stay finite in every plane (truncation never rounds up to inf): 1 x 1 conv with one-hot weights
    big = torch.full((1, 32, 8, 16), 3.4e38)
    eye = torch.zeros(64, 32, 1, 1)
    eye[torch.arange(32), torch.arange(32)] = 1.0
    wpe, _ = packing.pack_conv(eye.to(dev), None)
    with x3_mode:
        got = ops.conv2d([nhwc(big).to(dev)], wpe, None, 64, 1)
    assert torch.isfinite(got).all() and torch.equal(got[..., :32].cpu(), nhwc(big))


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,B,H,W", [
    (64, 128, 3, 1, 1, 1, 4, 16, 64),     # EncodeBlock second conv (cmm.py:49)
    (128, 256, 4, 2, 3, 2, 4, 16, 64),    # EncodeBlock first conv: stride 2, dilation 2
    (256, 512, 3, 1, 1, 1, 6, 4, 16),     # deep level: many tiles, few pixel splits
])
def test_conv_wgrad_x3_is_fp32_class(x3_mode, cin, cout, k, stride, pad, dil, B, H, W):
    """Weight gradient of the implicit-GEMM conv (128 x 128 tile of the power-of-two fast path) in mode 2 vs fp32-MFMA vs float64."""
    import torch.nn.functional as F
    from dpmn_amd import ops
    from dpmn_amd.train import pgrm_train
    x = u("wx", (B, cin, H, W))
    w = (u("ww", (cout, cin, k, k)) * 0.1).double().requires_grad_(True)
    y = F.conv2d(F.leaky_relu(x, 0.2).double(), w, None, stride, pad, dil)
    dy = u("wdy", tuple(y.shape))
    y.backward(dy.double())
    ref64 = w.grad
    xn, dyn = nhwc(x).to(dev), nhwc(dy).to(dev)

    def run():
        d = ops.conv_desc([xn], k, stride, pad, dil, cout=cout, pro_act="leaky02")
        dw = torch.zeros(cout, cin, k, k, device=dev)
        pgrm_train.conv_wgrad_into(d, dyn, dw, "conv")
        return dw
    ref32 = run()
    with x3_mode:
        got = run()
        again = run()
    assert torch.equal(got, again)
    e32, e3, dd = rel(ref32, ref64), rel(got, ref64), rel(got, ref32)
    tag = "x3_wgrad_%dx%d_k%d" % (cin, cout, k)
    record(tag, "fp32-MFMA kernel rel L2 vs float64", e32)
    record(tag, "bf16x3 kernel rel L2 vs float64", e3, 2.0 * e32)
    assert e3 <= 2.0 * e32 + 1e-8 and dd < 2e-6, (e32, e3, dd)


@pytest.mark.parametrize("M,N,K", [(49152, 96, 96), (49152, 384, 96), (24576, 96, 384), (8190, 192, 96), (1001, 144, 48)])
def test_linear_weight_gradient_x3_is_fp32_class(x3_mode, M, N, K):
    """dpmn_gemm_tn_f32 in mode 2 (gemm_tn_x3.hip: 32-row steps, the operands split in registers): dW = dY^T X and db vs the fp32-MFMA
    kernel and float64 -- ragged row counts (the last 32-row step and the last block run past M), 144 = one and a half block tiles,
    accumulation into a non-zero dW, two runs bitwise equal, operands spanning 30 binades."""
    from dpmn_amd.train import pgrm_train
    dy, x = u("tndy", (M, N), -2, 2), u("tnx", (M, K), -2, 2)
    dy = dy * torch.exp2(torch.randint(-15, 15, (M, 1), generator=torch.Generator().manual_seed(5)).float())
    dw0, db0 = u("tndw0", (N, K)), u("tndb0", (N,))
    ref_w = dw0.double() + dy.double().t() @ x.double()
    ref_b = db0.double() + dy.double().sum(0)

    def run():
        dw, db = dw0.clone().to(dev), db0.clone().to(dev)
        pgrm_train.gemm_tn(dy.to(dev), x.to(dev), dw, db)
        return dw, db
    w32, b32 = run()
    with x3_mode:
        w3, b3 = run()
        w3b, b3b = run()
    assert torch.equal(w3, w3b) and torch.equal(b3, b3b)
    e32, e3, d = rel(w32, ref_w), rel(w3, ref_w), rel(w3, w32)
    tag = "x3_gemm_tn_M%d_N%d_K%d" % (M, N, K)
    record(tag, "fp32-MFMA kernel dW rel L2 vs float64", e32)
    record(tag, "bf16x3 kernel dW rel L2 vs float64", e3, 2.0 * e32)
    record(tag, "bf16x3 vs fp32-MFMA kernel rel L2", d, 1e-6)
    record(tag, "bf16x3 db rel L2 vs float64", rel(b3, ref_b), 2e-6)
    assert e3 <= 2.0 * e32 + 1e-8, "bf16x3 dW is further from float64 (%.2e) than twice the fp32 kernel (%.2e)" % (e3, e32)
    assert d <= 1e-6 and rel(b3, ref_b) <= 2e-6


@pytest.mark.parametrize("M,N,K,epi", [(49152, 96, 96, "bias"), (49152, 384, 96, "gelu"), (49152, 96, 192, "res2"), (6144, 192, 192, "res1"),
                                       (49152, 96, 96, "none"), (1024, 96, 96, "bias")])
def test_rows_in_registers_linear_x3_is_fp32_class(x3_mode, M, N, K, epi):
    """dpmn_linear_f32 on the shapes of k_gemm_rowreg (K = 96 / 192, N multiple of 96) in mode 2 (gemm_rowreg_x3.hip: weight planes split
    once per block, rows split in registers): vs the fp32-MFMA kernel and float64, every epilogue; rows spanning 30 binades."""
    from dpmn_amd import ops
    x = u("rrx", (M, K), -2, 2) * torch.exp2(torch.randint(-15, 15, (M, 1), generator=torch.Generator().manual_seed(3)).float())
    w, b = u("rrw", (N, K), -0.3, 0.3), u("rrb", (N,))
    r1, r2 = u("rr1", (M, N)), u("rr2", (M, N))
    kw = dict(bias=None if epi == "none" else b.to(dev), act="gelu" if epi == "gelu" else "none",
              res1=r1.to(dev) if epi in ("res1", "res2") else None, res2=r2.to(dev) if epi == "res2" else None)
    xd, wd = x.to(dev), w.to(dev)
    ref32 = ops.linear(xd, wd, **kw)
    with x3_mode:
        got = ops.linear(xd, wd, **kw)
        got2 = ops.linear(xd, wd, **kw)
    assert torch.equal(got, got2)
    ref64 = x.double() @ w.double().t() + (0 if epi == "none" else b.double())
    if epi == "gelu":
        ref64 = torch.nn.functional.gelu(ref64)
    if epi in ("res1", "res2"):
        ref64 = ref64 + r1.double()
    if epi == "res2":
        ref64 = ref64 + r2.double()
    e32, e3, d = rel(ref32, ref64), rel(got, ref64), rel(got, ref32)
    tag = "x3_rowreg_M%d_N%d_K%d_%s" % (M, N, K, epi)
    record(tag, "fp32-MFMA kernel rel L2 vs float64", e32)
    record(tag, "bf16x3 kernel rel L2 vs float64", e3, 2.0 * e32)
    record(tag, "bf16x3 vs fp32-MFMA kernel rel L2", d, 1e-6)
    assert e3 <= 2.0 * e32 + 1e-8 and d <= 1e-6, (e32, e3, d)


@pytest.mark.parametrize("M,N,K,act", [(49152, 384, 96, "none"), (49152, 96, 96, "gelu"), (6144, 192, 192, "none")])
def test_layernorm_linear_x3_is_fp32_class(x3_mode, M, N, K, act):
    """dpmn_ln_linear_f32 (k_gemm_rowreg<K, PRO_LN>: gamma folded into the weights BEFORE the split, the normalisation behind the MFMAs)
    in mode 2 vs the fp32 kernel and float64; token rows with a large common offset (the text-prior branch's shape of input)."""
    from dpmn_amd import ops
    x = u("lnx", (M, K), -1, 1) + 3.0 * u("lno", (M, 1), -1, 1)
    g, bt = u("lng", (K,), 0.5, 1.5), u("lnb", (K,), -0.5, 0.5)
    w, b = u("lnw", (N, K), -0.3, 0.3), u("lnwb", (N,))
    args = [t.to(dev) for t in (x, g, bt, w, b)]
    ref32 = ops.ln_linear(*args, act=act)
    with x3_mode:
        got = ops.ln_linear(*args, act=act)
    ref64 = torch.nn.functional.layer_norm(x.double(), (K,), g.double(), bt.double(), 1e-5) @ w.double().t() + b.double()
    if act == "gelu":
        ref64 = torch.nn.functional.gelu(ref64)
    e32, e3, d = rel(ref32, ref64), rel(got, ref64), rel(got, ref32)
    tag = "x3_ln_linear_M%d_N%d_K%d" % (M, N, K)
    record(tag, "fp32-MFMA kernel rel L2 vs float64", e32)
    record(tag, "bf16x3 kernel rel L2 vs float64", e3, 2.0 * e32)
    record(tag, "bf16x3 vs fp32-MFMA kernel rel L2", d, 2e-6)
    assert e3 <= 2.0 * e32 + 1e-8 and d <= 2e-6, (e32, e3, d)


@pytest.mark.parametrize("save", [False, True])
def test_sk_select_layernorm_fc1_x3_is_fp32_class(x3_mode, save):
    """dpmn_sk_mlp_in_f32 in mode 2 (k_sk_mlp_in_x3): x1 (fp32 MFMAs in both modes) is bitwise the fp32 kernel's; y = fc1(LayerNorm2(x1)) vs
    the fp32 kernel and a float64 restatement of pgrm.py:91-96, 327-331, 31 on the same inputs; eval (folded LayerNorm) and training outputs."""
    from dpmn_amd import ops
    B, L, C, G, N = 48, 1024, 96, 3, 384
    M = B * L
    cat, avec = u("skc", (M, C), -2, 2), u("ska", (B, G, C // G), 0, 1)
    hw, hb = u("skhw", (C, C // G), -0.3, 0.3), u("skhb", (C,))
    feats, short = u("skf", (M, C)), u("sks", (M, C), -1, 1) + 2.0
    g, bt = u("skg", (C,), 0.5, 1.5), u("skbt", (C,), -0.5, 0.5)
    fw, fb = u("skfw", (N, C), -0.3, 0.3), u("skfb", (N,))
    args = [t.to(dev) for t in (cat, avec, hw, hb, feats, short, g, bt, fw, fb)]
    out32 = ops.sk_mlp_in(*args, L, save=save)
    with x3_mode:
        out3 = ops.sk_mlp_in(*args, L, save=save)
    assert torch.equal(out32[0], out3[0]), "x1 differs between the modes"
    if save:
        assert torch.equal(out32[2], out3[2]) and torch.equal(out32[3], out3[3])
    sel = (cat.double().reshape(B, L, G, C // G) * avec.double()[:, None]).sum(2).reshape(M, C // G)
    x1 = sel @ hw.double().t() + hb.double() + feats.double() + short.double()
    y64 = torch.nn.functional.layer_norm(x1, (C,), g.double(), bt.double(), 1e-5) @ fw.double().t() + fb.double()
    e32, e3, d = rel(out32[1], y64), rel(out3[1], y64), rel(out3[1], out32[1])
    tag = "x3_sk_mlp_in_%s" % ("train" if save else "eval")
    record(tag, "fp32-MFMA kernel y rel L2 vs float64", e32)
    record(tag, "bf16x3 kernel y rel L2 vs float64", e3, 2.0 * e32)
    record(tag, "bf16x3 vs fp32-MFMA kernel rel L2", d, 2e-6)
    assert e3 <= 2.0 * e32 + 1e-8 and d <= 2e-6, (e32, e3, d)

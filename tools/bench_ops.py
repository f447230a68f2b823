"""Micro-benchmarks of individual kernels at the config-1 shapes (B=48).  Usage: python tools/bench_ops.py [filter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops
from dpmn_amd.model import packing
from dpmn_amd.utils import synth

dev = torch.device("cuda:0")
flt = sys.argv[1] if len(sys.argv) > 1 else ""


def u(name, shape, lo=-1.0, hi=1.0):
    return synth.uniform(name, shape, lo, hi, 70).to(dev)


def timeit(name, fn, flops, bytes_=0, reps=20):
    if flt and flt not in name:
        return
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    ms = ts[len(ts) // 2]
    print("%-46s %8.1f us  %7.1f TFLOP/s  %6.2f TB/s" % (name, ms * 1e3, flops / ms / 1e9, bytes_ / ms / 1e9))


B, L, C = 48, 1024, 96
M = B * L
x = u("x", (M, C)); g = u("g", (C,), 0.5, 1.5); be = u("be", (C,))
for N in (96, 192, 384):
    w = u("w%d" % N, (N, C), -0.2, 0.2); b = u("b%d" % N, (N,))
    timeit("ln_linear K=96 N=%d" % N, lambda: ops.ln_linear(x, g, be, w, b), 2.0 * M * N * C, 4.0 * M * (C + N))
    timeit("ln_linear+gelu K=96 N=%d" % N, lambda: ops.ln_linear(x, g, be, w, b, act="gelu"), 2.0 * M * N * C, 4.0 * M * (C + N))
    timeit("linear K=96 N=%d" % N, lambda: ops.linear(x, w, b), 2.0 * M * N * C, 4.0 * M * (C + N))
x4 = u("x4", (M, 384)); w2 = u("w2", (96, 384), -0.1, 0.1); b2 = u("b2", (96,)); r = u("r", (M, 96))
timeit("linear(fc2) K=384 N=96 +res", lambda: ops.linear(x4, w2, b2, res1=r), 2.0 * M * 96 * 384, 4.0 * M * (384 + 192))
gg = u("gg", (B, L, 384)); wp = u("wp", (384, 384), -0.1, 0.1); bp = u("bp", (384,))
timeit("pointwise 384x384x1024", lambda: ops.pointwise(gg, wp, bp), 2.0 * B * 384 * 384 * 1024, 8.0 * B * 384 * 1024)
dw = u("dw", (384, 1, 3, 3)); db = u("db", (384,))
timeit("dwconv3x3+gelu", lambda: ops.dwconv3x3_gelu(gg, dw, db, 32), 18.0 * B * 384 * 1024, 8.0 * B * 384 * 1024)


def conv_case(name, cin, cout, k, H, W, stride=1, pad=None, dil=1, segs=None, **kw):
    pad = (k - 1) // 2 if pad is None else pad
    segs = segs or [cin]
    xs = [u("cx%d_%s" % (i, name), (B, H, W, c)) for i, c in enumerate(segs)]
    w = u("cw" + name, (cout, cin, k, k), -1, 1) * (1.0 / (cin * k * k) ** 0.5)
    bb = u("cb" + name, (cout,))
    wpk, bpk = packing.pack_conv(w, bb)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    fl = 2.0 * B * Ho * Wo * cout * cin * k * k
    by = 4.0 * B * (H * W * cin + Ho * Wo * cout)
    timeit("conv " + name, lambda: ops.conv2d(xs, wpk, bpk, cout, k, stride=stride, pad=pad, dil=dil, **kw), fl, by)


conv_case("tatt 3x3 64->64 @16x64", 64, 64, 3, 16, 64)
conv_case("tatt 3x3 64->64 @16x64 +mish", 64, 64, 3, 16, 64, epi_act="mish")
conv_case("tatt up 3x3 64->256 @16x64 ps", 64, 256, 3, 16, 64, epi_act="mish", pixel_shuffle=True)
conv_case("tatt gi 1x1 128->192 @16x64", 128, 192, 1, 16, 64, segs=[64, 64])
conv_case("tatt gi 1x1 64->192 @16x64", 64, 192, 1, 16, 64)
conv_case("tatt last 9x9 64->4 @32x128", 64, 4, 9, 32, 128, epi_act="tanh", out_nchw=True)
conv_case("tatt first 9x9 4->64 @16x64", 4, 64, 9, 16, 64, epi_act="prelu", slope=0.25)
conv_case("cmm en2a 4x4s2d2 64->64 @32x128", 64, 64, 4, 32, 128, stride=2, pad=3, dil=2, pro_act="leaky02")
conv_case("cmm en2b 3x3 64->128 @16x64", 64, 128, 3, 16, 64, pro_act="leaky02")
conv_case("cmm en3a 4x4s2d2 128->128 @16x64", 128, 128, 4, 16, 64, stride=2, pad=3, dil=2, pro_act="leaky02")
conv_case("cmm en3b 3x3 128->256 @8x32", 128, 256, 3, 8, 32, pro_act="leaky02")
conv_case("cmm en4a 4x4s2d2 256->256 @8x32", 256, 256, 4, 8, 32, stride=2, pad=3, dil=2, pro_act="leaky02")
conv_case("cmm en4b 3x3 256->512 @4x16", 256, 512, 3, 4, 16, pro_act="leaky02")
conv_case("cmm en5a 4x4s2d2 512->512 @4x16", 512, 512, 4, 4, 16, stride=2, pad=3, dil=2, pro_act="leaky02")
conv_case("cmm en5b 3x3 512->512 @2x8", 512, 512, 3, 2, 8, pro_act="leaky02")
conv_case("cmm en6 4x4s2 512->512 @2x8", 512, 512, 4, 2, 8, stride=2, pad=1, pro_act="leaky02")
conv_case("cmm de5a 3x3 1536->512 @2x8", 1536, 512, 3, 2, 8, segs=[512, 512, 512], pro_act="relu")
conv_case("cmm de4a 3x3 1536->256 @4x16", 1536, 256, 3, 4, 16, segs=[512, 512, 512], pro_act="relu")
conv_case("cmm de3a 3x3 768->128 @8x32", 768, 128, 3, 8, 32, segs=[256, 256, 256], pro_act="relu")
conv_case("cmm de2a 3x3 384->64 @16x64", 384, 64, 3, 16, 64, segs=[128, 128, 128], pro_act="relu")
conv_case("cmm de1 3x3 192->3 @32x128", 192, 3, 3, 32, 128, segs=[64, 64, 64], pro_act="relu", out_nchw=True)
conv_case("pgrm tail 3x3 96->12 @16x64", 96, 12, 3, 16, 64)


# ---- conv weight gradients (training): packed layout vs straight into the nn.Conv2d layout
def wgrad_case(name, cin, cout, k, H, W):
    if flt and flt not in "wgrad " + name:
        return
    import ctypes as C_
    from dpmn_amd._abi import lib, check, dptr, stream
    from dpmn_amd.train.pgrm_train import conv_wgrad_into
    x = u("wgx" + name, (B, H, W, cin)); dy = u("wgy" + name, (B, H, W, cout))
    d = ops.conv_desc([x], k, pad=(k - 1) // 2, cout=cout)
    kp = (k * k * cin + 31) // 32 * 32
    dwp = torch.zeros(cout, kp, device=dev); dw = torch.zeros(cout, cin, k, k, device=dev)
    fl = 2.0 * B * H * W * cout * cin * k * k
    by = 4.0 * B * H * W * (cin + cout)
    timeit("wgrad packed " + name, lambda: check(lib.dpmn_conv2d_wgrad_f32(C_.byref(d), dptr(dy), dptr(dwp), 1, stream())), fl, by)
    timeit("wgrad param  " + name, lambda: conv_wgrad_into(d, dy, dw), fl, by)


wgrad_case("96->64 k3 16x64", 96, 64, 3, 16, 64)
wgrad_case("64->64 k3 32x128", 64, 64, 3, 32, 128)
wgrad_case("128->128 k3 16x64", 128, 128, 3, 16, 64)
wgrad_case("256->256 k3 8x32", 256, 256, 3, 8, 32)
wgrad_case("256->512 k3 4x16", 256, 512, 3, 4, 16)
wgrad_case("512->512 k3 2x8", 512, 512, 3, 2, 8)
wgrad_case("192->4 k3 32x128", 192, 4, 3, 32, 128)
wgrad_case("8->4 k3 32x128", 8, 4, 3, 32, 128)


# ---- linear weight gradients dW = dY^T X (+ fused bias gradient), tokens M = 49152
def gemm_tn_case(N, K):
    from dpmn_amd.train.pgrm_train import gemm_tn
    dy = u("tn_dy%d" % N, (M, N)); xx = u("tn_x%d" % K, (M, K))
    big = torch.zeros(64 << 20, device=dev)          # cold destination, like the optimizer's flat gradient bucket
    dw = big[(32 << 20):(32 << 20) + N * K].view(N, K); db = big[(48 << 20):(48 << 20) + N]
    timeit("gemm_tn N=%d K=%d" % (N, K), lambda: gemm_tn(dy, xx, dw), 2.0 * M * N * K, 4.0 * M * (N + K))
    timeit("gemm_tn+db N=%d K=%d" % (N, K), lambda: gemm_tn(dy, xx, dw, db), 2.0 * M * N * K, 4.0 * M * (N + K))


for N_, K_ in ((96, 96), (192, 96), (384, 96), (96, 384)):
    gemm_tn_case(N_, K_)


# ---- k-loop GEMM users: fc2 (K = 384 -> 96, + residual) and the pointwise-conv weight gradient (384 x 384 outputs, K = B * L)
def kloop_cases():
    from dpmn_amd._abi import lib, check, dptr, stream
    Ch = 384
    y = u("kl_y", (M, Ch)); w2 = u("kl_w2", (C, Ch), -0.1, 0.1); b2 = u("kl_b2", (C,)); res = u("kl_r", (M, C))
    timeit("kloop fc2 K=384 N=96 +res", lambda: ops.linear(y, w2, b2, res1=res), 2.0 * M * C * Ch, 4.0 * M * (Ch + 2 * C))
    dz = u("kl_dz", (B, Ch, L)); g = u("kl_g", (B, Ch, L)); dw = torch.zeros(Ch, Ch, device=dev)
    timeit("kloop pointwise wgrad 384x384 K=49152", lambda: check(lib.dpmn_pointwise_wgrad_f32(dptr(dz), dptr(g), dptr(dw), B, Ch, L, stream())),
           2.0 * Ch * Ch * B * L, 8.0 * B * Ch * L)


if not flt or "kloop" in flt:
    kloop_cases()

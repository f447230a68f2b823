"""ORACLE (test infrastructure only) -- CPU restatement of the reference CMM, DistillModule,
ImageLoss, PSNR/SSIM and the branch-2 mask prior.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
Pinned by tests/golden/{cmm_*,distill,loss,metrics}.npz (tools/gen_golden.py, imported reference).

Reference lines restated:
  model/cmm.py: EncodeBlock 38-55, DecodeBlock 58-77, ComplementationModulationModule.forward 120-161
  model/distill_module.py: DistillModule.forward 18-31
  loss/image_loss.py: ImageLoss.forward 15-21, GradientPriorLoss.gradient_map 35-43
  utils/ssim_psnr.py: calculate_psnr 9-13, _ssim 28-48, SSIM.forward 62-79
  utils/util.py: toMask 27-35 (PIL 'L' conversion restated in integer arithmetic: parity UNPINNED
  for ToPILImage's float->uint8 cast of out-of-range values)
"""
import math

import torch
import torch.nn.functional as F


class _KinkAct(torch.autograd.Function):
    """LeakyReLU / ReLU whose DERIVATIVE branch is dictated: forward = the ordinary activation, backward multiplies by 1 where
    `pos` is set and by `slope` elsewhere, whatever the sign of x.  The float64 adjudication of the gradient fixtures
    (tests/helpers.py kink_adjudicated_f64) uses it to differentiate, in float64, the SAME piecewise-linear branch an fp32
    implementation took at the few pre-activations that lie within fp32 round-off of the kink (|x| ~ 1e-7: act(x) ~ 0 either way, the
    derivative is a coin flip between 1 and slope, and the flip moves every upstream gradient by O(1e-3 ... 1e-2))."""

    @staticmethod
    def forward(ctx, x, pos, slope):
        ctx.save_for_backward(pos)
        ctx.slope = slope
        return torch.where(x > 0, x, x * slope)

    @staticmethod
    def backward(ctx, g):
        pos, = ctx.saved_tensors
        return torch.where(pos, g, g * ctx.slope), None, None


def _act(x, slope, site, hook):
    """The activation that opens an Encode / DecodeBlock (cmm.py:40,47,60,68) on the tensor produced by module `site`.
    hook(site, x) -> None (plain activation) or a bool tensor: where the derivative takes the x > 0 branch."""
    pos = hook(site, x) if hook is not None else None
    if pos is None:
        return F.leaky_relu(x, slope) if slope else F.relu(x)
    return _KinkAct.apply(x, pos, slope)


def _bn(x, sd, pre, training, eps=1e-5):
    """BatchNorm2d: batch statistics when training (biased var for normalisation), else running."""
    if training:
        return F.batch_norm(x, None, None, sd[pre + "weight"], sd[pre + "bias"], True, 0.0, eps)
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"],
                        sd[pre + "bias"], False, 0.0, eps)


def _encode_block(x, sd, pre, training, src, hook=None):
    """src: name of the module that produced x (the activation site, see _act)"""
    x = _act(x, 0.2, src + ">" + pre + "encode.0", hook)
    x = F.conv2d(x, sd[pre + "encode.1.weight"], sd[pre + "encode.1.bias"], stride=2, padding=3, dilation=2)
    x = _bn(x, sd, pre + "encode.2.", training)
    x = _act(x, 0.2, pre + "encode.2>" + pre + "encode.3", hook)
    x = F.conv2d(x, sd[pre + "encode.4.weight"], sd[pre + "encode.4.bias"], padding=1)
    return _bn(x, sd, pre + "encode.5.", training)


def _decode_block(parts, names, sd, pre, training, hook=None):
    """parts / names: the channel-concatenated inputs and the modules that produced them (relu(cat(...)) = cat(relu(...)))"""
    x = torch.cat([_act(t, 0.0, n + ">" + pre + "decode.0", hook) for t, n in zip(parts, names)], 1)
    x = F.conv_transpose2d(x, sd[pre + "decode.1.weight"], sd[pre + "decode.1.bias"], stride=1, padding=1)
    x = _bn(x, sd, pre + "decode.2.", training)
    x = _act(x, 0.0, pre + "decode.2>" + pre + "decode.3", hook)
    x = F.conv_transpose2d(x, sd[pre + "decode.4.weight"], sd[pre + "decode.4.bias"], stride=2, padding=1)
    return _bn(x, sd, pre + "decode.5.", training)


def cmm_forward(sd, x1, x2, training=False, act_hook=None):
    """act_hook: see _act -- sites are named "<producer module>><consumer activation module>", e.g. "en_4_1.encode.5>en_5_1.encode.0"
    (the same tensor feeds an encoder LeakyReLU and a decoder ReLU: two sites)."""
    enc = []
    for br, x in (("1", x1), ("2", x2)):
        o1 = F.conv2d(x, sd["en_1_%s.weight" % br], sd["en_1_%s.bias" % br], padding=1)
        o2 = _encode_block(o1, sd, "en_2_%s." % br, training, "en_1_%s" % br, act_hook)
        o3 = _encode_block(o2, sd, "en_3_%s." % br, training, "en_2_%s.encode.5" % br, act_hook)
        o4 = _encode_block(o3, sd, "en_4_%s." % br, training, "en_3_%s.encode.5" % br, act_hook)
        o5 = _encode_block(o4, sd, "en_5_%s." % br, training, "en_4_%s.encode.5" % br, act_hook)
        o6 = F.conv2d(_act(o5, 0.2, "en_5_%s.encode.5>en_6_%s.0" % (br, br), act_hook), sd["en_6_%s.1.weight" % br], sd["en_6_%s.1.bias" % br],
                      stride=2, padding=1)
        enc.append((o1, o2, o3, o4, o5, o6))
    a, b = enc
    res = torch.cat([a[5], b[5]], 1)
    s = res.mean(dim=(2, 3))  # (N, C) squeeze
    w = torch.sigmoid(F.linear(F.relu(F.linear(s, sd["fc_1.weight"], sd["fc_1.bias"])),
                               sd["fc_2.weight"], sd["fc_2.bias"]))
    o6 = res * w[:, :, None, None] + res
    d = F.conv_transpose2d(_act(o6, 0.0, "gate>de_6.0", act_hook), sd["de_6.1.weight"], sd["de_6.1.bias"], stride=2, padding=1)
    d = _bn(d, sd, "de_6.2.", training)
    dn = "de_6.2"
    enc_name = lambda br, i: "en_1_%s" % br if i == 0 else "en_%d_%s.encode.5" % (i + 1, br)
    for lvl, skip in ((5, 4), (4, 3), (3, 2), (2, 1)):
        d = _decode_block([d, a[skip], b[skip]], [dn, enc_name("1", skip), enc_name("2", skip)], sd, "de_%d." % lvl, training, act_hook)
        dn = "de_%d.decode.5" % lvl
    d = torch.cat([_act(t, 0.0, n + ">de_1.0", act_hook) for t, n in zip((d, a[0], b[0]), (dn, "en_1_1", "en_1_2"))], 1)
    return F.conv_transpose2d(d, sd["de_1.1.weight"], sd["de_1.1.bias"], stride=1, padding=1)


def distill_forward(sd, x_deep, x_shallow, training=True):
    fc = F.conv2d(torch.cat([x_deep, x_shallow], 1), sd["conv_cat_feature.weight"],
                  sd["conv_cat_feature.bias"], padding=1)
    fc = F.relu(_bn(fc, sd, "bn_1.", training))
    fs = F.conv2d(x_shallow, sd["conv_feature.weight"], sd["conv_feature.bias"], padding=1)
    fs = F.relu(_bn(fs, sd, "bn_2.", training))
    return (fc - fs).abs().mean(), fc


def gradient_map(x):
    xp = F.pad(x, (1, 1, 1, 1))
    gx = (xp[:, :, 1:-1, 2:] - xp[:, :, 1:-1, :-2]) * 0.5
    gy = (xp[:, :, :-2, 1:-1] - xp[:, :, 2:, 1:-1]) * 0.5
    return torch.sqrt(gx * gx + gy * gy + 1e-6)


def image_loss(out, tgt, gradient=True, w=(1.0, 1.0)):
    loss = w[0] * ((out - tgt) ** 2).mean()
    if gradient:
        loss = loss + w[1] * (gradient_map(out[:, :3]) - gradient_map(tgt[:, :3])).abs().mean()
    return loss


def psnr(a, b):
    mse = ((a[:, :3] * 255 - b[:, :3] * 255) ** 2).mean()
    return 20 * torch.log10(255.0 / torch.sqrt(mse))


def ssim(a, b, ws=11, sigma=1.5):
    a, b = a[:, :3], b[:, :3]
    C = a.shape[1]
    g = torch.tensor([math.exp(-(x - ws // 2) ** 2 / float(2 * sigma ** 2)) for x in range(ws)])
    g = g / g.sum()
    win = (g[:, None] @ g[None, :]).float().expand(C, 1, ws, ws).contiguous()

    def blur(x):
        return F.conv2d(x, win, padding=ws // 2, groups=C)

    mu1, mu2 = blur(a), blur(b)
    s11 = blur(a * a) - mu1 * mu1
    s22 = blur(b * b) - mu2 * mu2
    s12 = blur(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def to_mask(img):
    """toMask for a batch (B,3,H,W) -> (B,3,H,W) in {0,1}.  ToPILImage: mul(255).byte() (wraps for
    out-of-range floats, implementation-defined -> we use floor-mod 256 of the truncated value);
    PIL RGB->L: (R*19595 + G*38470 + B*7471 + 0x8000) >> 16; pixel -> 255 where L <= mean(L)."""
    u = (img * 255.0).to(torch.int64) % 256  # trunc toward zero then wrap, as .byte() does
    L = (u[:, 0] * 19595 + u[:, 1] * 38470 + u[:, 2] * 7471 + 0x8000) >> 16
    thr = L.double().mean(dim=(1, 2), keepdim=True)
    m = (L.double() <= thr).float()
    return m[:, None].repeat(1, 3, 1, 1)

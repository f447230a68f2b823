"""Weight packing for the NHWC implicit-GEMM conv kernel (csrc/conv.hip).

Host-side layout plumbing only (torch permute / reshape on whatever device the parameter
lives on); results are cached per parameter version by the owning module.  Packed layout:
(Cout, Kp) with K index = (ky*KW + kx)*Cin_p + ci, Kp = K rounded up to 32, zero filled.

Eval-mode BatchNorm is folded into the producing conv (quirk Q12):
    W' = W * g/sqrt(var+eps),  b' = (b - mean) * g/sqrt(var+eps) + beta.
"""
import torch


def _finish(w_okkc, bias, bn):
    """w_okkc: (Cout, KH, KW, Cin_p) -> (Cout, Kp) fp32 contiguous, optional BN fold."""
    cout = w_okkc.shape[0]
    w2 = w_okkc.reshape(cout, -1)
    b = bias if bias is not None else torch.zeros(cout, device=w2.device, dtype=w2.dtype)
    if bn is not None:
        gamma, beta, mean, var, eps = bn
        s = gamma / torch.sqrt(var + eps)
        w2 = w2 * s[:, None]
        b = (b - mean) * s + beta
    k = w2.shape[1]
    kp = (k + 31) // 32 * 32
    if kp != k:
        w2 = torch.cat([w2, w2.new_zeros(cout, kp - k)], 1)
    return w2.contiguous().float(), b.contiguous().float()


def pack_conv(weight, bias=None, bn=None, cin_pad=None):
    """nn.Conv2d weight (Cout, Cin, KH, KW)."""
    w = weight.permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad != w.shape[-1]:
        w = torch.cat([w, w.new_zeros(*w.shape[:-1], cin_pad - w.shape[-1])], -1)
    return _finish(w, bias, bn)


def pack_convT_s1(weight, bias=None, bn=None):
    """nn.ConvTranspose2d(k, stride=1, padding=p) weight (Cin, Cout, KH, KW) == Conv2d with the
    flipped, in/out-swapped kernel and padding KH-1-p."""
    w = weight.flip(2, 3).permute(1, 2, 3, 0)
    return _finish(w, bias, bn)


def pack_convT_s2k4(weight, bias=None, bn=None):
    """nn.ConvTranspose2d(4, stride=2, padding=1) weight (Cin, Cout, 4, 4) -> 4 phase packs.
    Output pixel (2m+py, 2n+px) = sum_{ky',kx'} in[m + py - ky', n + px - kx'] * W[:, :, (1-py)+2ky', (1-px)+2kx'],
    i.e. a 2x2 conv with dilation -1 and padding -phase."""
    packs = []
    for py in range(2):
        for px in range(2):
            sub = weight[:, :, (1 - py)::2, (1 - px)::2]          # (Cin, Cout, 2, 2), index ky' -> ky=(1-py)+2ky'
            packs.append(_finish(sub.permute(1, 2, 3, 0), bias, bn))
    return packs


def bn_tuple(mod_sd, prefix, eps=1e-5):
    return (mod_sd[prefix + "weight"], mod_sd[prefix + "bias"], mod_sd[prefix + "running_mean"],
            mod_sd[prefix + "running_var"], eps)


# ---------------------------------------------------------------------------------------- training: un-packing
def unpack_conv(dwp, weight_shape, cin_pad=None):
    """inverse of pack_conv for the weight gradient: (Cout, Kp) -> (Cout, Cin, KH, KW)."""
    cout, cin, kh, kw = weight_shape
    cp = cin_pad or cin
    return dwp[:, :kh * kw * cp].reshape(cout, kh, kw, cp)[..., :cin].permute(0, 3, 1, 2)


def unpack_convT_s1(dwp, weight_shape):
    """inverse of pack_convT_s1: packed (Cout, KH*KW*Cin) -> ConvTranspose2d weight gradient (Cin, Cout, KH, KW)."""
    cin, cout, kh, kw = weight_shape
    return dwp[:, :kh * kw * cin].reshape(cout, kh, kw, cin).flip(1, 2).permute(3, 0, 1, 2)


def unpack_convT_s2k4_into(dweight, dwps):
    """accumulate the 4 phase gradients (each (Cout, 4*Cin) packed) into the (Cin, Cout, 4, 4) weight gradient."""
    cin, cout = dweight.shape[0], dweight.shape[1]
    i = 0
    for py in range(2):
        for px in range(2):
            g = dwps[i][:, :4 * cin].reshape(cout, 2, 2, cin).permute(3, 0, 1, 2)
            dweight[:, :, (1 - py)::2, (1 - px)::2] += g
            i += 1


def pack_bwd_data_generic(weight):
    """Conv2d weight (Cout, Cin, KH, KW) -> (Cin, KH*KW*Cout) pack for the data gradient run as a dil=-1 conv of dY."""
    cout, cin, kh, kw = weight.shape
    return _finish(weight.permute(1, 2, 3, 0), None, None)

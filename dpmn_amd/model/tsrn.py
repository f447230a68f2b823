"""Host-side mirrors of the reference PSN backbones ``model/tsrn.py::TSRN`` and
``model/tatt.py::TSRN_TL_TRANS`` (TATT) -- eval mode, which is the only mode DPMN runs them in
(frozen PSN, super_resolution.py:56-59).

Same constructors, forward signatures/returns and ``state_dict`` key layouts as the reference; the
``torch.nn`` layers below only HOLD parameters (they are never called).  All arithmetic runs in
libdpmn_hip.so: NHWC implicit-GEMM convs with folded eval BatchNorm and fused mish / PReLU / tanh /
PixelShuffle epilogues (csrc/conv.hip), the BiGRU recurrence with the 1x1 conv folded into the
input projection, and the TPInterpreter transformer pieces (csrc/tatt.hip, csrc/gemm.hip).
"""
import math
import os

import torch
import torch.nn as nn

from .. import ops
from . import packing


NATIVE_TRUNK = os.environ.get("DPMN_PSN_NATIVE", "1") != "0"      # SRBs + tail through dpmn_psn_trunk_f32 (one native call)


class _GruBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 1)
        self.gru = nn.GRU(cout, cout // 2, bidirectional=True, batch_first=True)


class _SRB(nn.Module):
    def __init__(self, ch, text_ch=0):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(ch)
        self.gru1 = _GruBlock(ch + text_ch, ch)
        self.prelu = nn.Identity()   # mish, parameter-free
        self.conv2 = nn.Conv2d(ch, ch, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(ch)
        self.gru2 = _GruBlock(ch, ch)


class _Upsample(nn.Module):
    def __init__(self, ch, scale):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch * scale * scale, 3, padding=1)


def _bn(m):
    return (m.weight, m.bias, m.running_mean, m.running_var, m.eps)


def _pack_gru_block(gb):
    """Fold conv1x1 into the GRU input projection: gi = W_ih (Wc x + bc) + b_ih, both directions stacked."""
    g = gb.gru
    wc = gb.conv1.weight.reshape(gb.conv1.weight.shape[0], -1)   # (C, Cin)
    bc = gb.conv1.bias
    wi = torch.cat([g.weight_ih_l0, g.weight_ih_l0_reverse], 0)  # (6H, C)
    bi = torch.cat([g.bias_ih_l0, g.bias_ih_l0_reverse], 0)
    w = (wi.double() @ wc.double()).float()
    b = (wi.double() @ bc.double()).float() + bi
    kp = (w.shape[1] + 31) // 32 * 32
    if kp != w.shape[1]:
        w = torch.cat([w, w.new_zeros(w.shape[0], kp - w.shape[1])], 1)
    whh = torch.stack([g.weight_hh_l0, g.weight_hh_l0_reverse], 0).contiguous()
    bhh = torch.stack([g.bias_hh_l0, g.bias_hh_l0_reverse], 0).contiguous()
    return w.contiguous(), b.contiguous(), whh, bhh


class _PSNBase(nn.Module):
    """Shared conv/GRU trunk of TSRN and TATT."""

    def _build_trunk(self, scale_factor, width, height, STN, srb_nums, mask, hidden_units, text_ch):
        self.stn = bool(STN)      # parameters are held for checkpoint compatibility (model/stn.py); eval never runs them
        self._stn_hw = (height // scale_factor, width // scale_factor)
        assert math.log(scale_factor, 2) % 1 == 0 and scale_factor == 2, "built for scale_factor=2"
        self.in_planes = 4 if mask else 3
        self.srb_nums = srb_nums
        ch = 2 * hidden_units
        self.ch = ch
        self.block1 = nn.Sequential(nn.Conv2d(self.in_planes, ch, 9, padding=4), nn.PReLU())
        for i in range(srb_nums):
            setattr(self, "block%d" % (i + 2), _SRB(ch, text_ch))

    def _build_tail(self, srb_nums):
        ch = self.ch
        setattr(self, "block%d" % (srb_nums + 2), nn.Sequential(nn.Conv2d(ch, ch, 3, padding=1), nn.BatchNorm2d(ch)))
        setattr(self, "block%d" % (srb_nums + 3), nn.Sequential(_Upsample(ch, 2), nn.Conv2d(ch, self.in_planes, 9, padding=4)))
        if getattr(self, "stn", False):   # tsrn.py:44-56 / tatt.py:57-72 / tbsrn.py:199-212: registered after the blocks
            from .stn import STNHead, TPSSpatialTransformer
            self.tps = TPSSpatialTransformer(output_image_size=tuple(self._stn_hw), num_control_points=20, margins=(0.05, 0.05))
            self.stn_head = STNHead(in_planes=self.in_planes, num_ctrlpoints=20, activation='none')

    # ------------------------------------------------------------------ packing (cached)
    def _trunk_pack(self):
        key = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if getattr(self, "_pk", None) is not None and self._pk[0] == key:
            return self._pk[1]
        P = {}
        n = self.srb_nums
        with torch.no_grad():
            cin_p = 4
            P["block1"] = packing.pack_conv(self.block1[0].weight, self.block1[0].bias, cin_pad=cin_p)
            P["prelu"] = float(self.block1[1].weight.reshape(-1)[0].item())
            for i in range(n):
                blk = getattr(self, "block%d" % (i + 2))
                P["srb%d.c1" % i] = packing.pack_conv(blk.conv1.weight, blk.conv1.bias, _bn(blk.bn1))
                P["srb%d.c2" % i] = packing.pack_conv(blk.conv2.weight, blk.conv2.bias, _bn(blk.bn2))
                self._pack_block_extra(P, i, blk)
            b7 = getattr(self, "block%d" % (n + 2))
            P["b7"] = packing.pack_conv(b7[0].weight, b7[0].bias, _bn(b7[1]))
            b8 = getattr(self, "block%d" % (n + 3))
            P["up"] = packing.pack_conv(b8[0].conv.weight, b8[0].conv.bias)
            P["last"] = packing.pack_conv(b8[1].weight, b8[1].bias)
            self._extra_pack(P)
        self._pk = (key, P)
        return P

    def _extra_pack(self, P):
        pass

    def _pack_block_extra(self, P, i, blk):
        P["srb%d.g1" % i] = _pack_gru_block(blk.gru1)
        P["srb%d.g2" % i] = _pack_gru_block(blk.gru2)

    def _check_mode(self):
        if self.training:
            raise NotImplementedError("dpmn_amd PSN: only eval mode is built (DPMN keeps the PSN frozen in .eval(), "
                                      "super_resolution.py:56-59)")

    # ------------------------------------------------------------------ kernels
    def _srb(self, x, P, i, tp=None):
        B, H, W, Cc = x.shape
        r = ops.conv2d([x], *P["srb%d.c1" % i], Cc, 3, pad=1, epi_act="mish")
        r = ops.conv2d([r], *P["srb%d.c2" % i], Cc, 3, pad=1)
        # GruBlock.conv1 (1x1) is folded into the GRU input projection: one (pixels, Cin) x (Cin, 192) GEMM
        w1, b1, whh1, bhh1 = P["srb%d.g1" % i]
        M = B * H * W
        if tp is None:
            gi = ops.linear(r.reshape(M, Cc), w1, b1)
        else:
            gi = ops.cat2_linear(r.reshape(M, Cc), tp.reshape(M, tp.shape[3]), w1, b1)
        s = ops.bigru(gi, whh1, bhh1, B, H, W, "h", res=x)              # x + gru1(...) (vertical pass)
        w2, b2, whh2, bhh2 = P["srb%d.g2" % i]
        gi = ops.linear(s.reshape(M, Cc), w2, b2)
        return ops.bigru(gi, whh2, bhh2, B, H, W, "w")

    def _native(self, P):
        """dpmn_psn_weights over the packed weights (include/dpmn_hip.h), rebuilt when the pack is."""
        from .. import _abi
        nat = getattr(self, "_nat", None)
        if nat is not None and nat[0] is P:
            return nat[1], nat[2]
        w = _abi.PsnWeights()
        w.in_planes, w.ch, w.hidden, w.srb_nums = self.in_planes, self.ch, self.ch // 2, self.srb_nums
        keep = []
        for i in range(self.srb_nums):
            q = w.srb[i]
            (q.c1_w, q.c1_b), (q.c2_w, q.c2_b) = [(t.data_ptr() for t in P["srb%d.c%d" % (i, j)]) for j in (1, 2)]
            for g in (1, 2):
                gw, gb, whh, bhh = P["srb%d.g%d" % (i, g)]
                setattr(q, "g%d_w" % g, gw.data_ptr()); setattr(q, "g%d_b" % g, gb.data_ptr())
                setattr(q, "g%d_whh" % g, whh.data_ptr()); setattr(q, "g%d_bhh" % g, bhh.data_ptr())
        (w.b7_w, w.b7_b), (w.up_w, w.up_b), (w.last_w, w.last_b) = [(t.data_ptr() for t in P[k]) for k in ("b7", "up", "last")]
        self._nat = (P, w, keep)
        if not hasattr(self, "_native_ws"):
            self._native_ws = {}
        return w, keep

    def _trunk(self, b1, P, tp=None):
        """SRBs + tail: one native call (csrc/psn_forward.hip), or (DPMN_PSN_NATIVE=0) the per-op composition below."""
        if NATIVE_TRUNK and "srb0.g1" in P:
            return ops.psn_trunk(*self._native(P), b1, tp, self.in_planes, self._native_ws)
        f = b1
        for i in range(self.srb_nums):
            f = self._srb(f, P, i, tp)
        return self._tail(b1, f, P)

    def _head(self, x, P):
        xin = ops.nchw_to_nhwc(x.contiguous().float(), 4)
        return ops.conv2d([xin], *P["block1"], self.ch, 9, pad=4, epi_act="prelu", slope=P["prelu"])

    def _tail(self, b1, f, P):
        t = ops.conv2d([f], *P["b7"], self.ch, 3, pad=1, res=b1)
        u = ops.conv2d([t], *P["up"], 4 * self.ch, 3, pad=1, epi_act="mish", pixel_shuffle=True)
        return ops.conv2d([u], *P["last"], self.in_planes, 9, pad=4, epi_act="tanh", out_nchw=True)


class TSRN(_PSNBase):
    """Drop-in for ``model.tsrn.TSRN`` (tsrn.py:14-74), eval forward."""

    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32):
        super().__init__()
        self._build_trunk(scale_factor, width, height, STN, srb_nums, mask, hidden_units, 0)
        self._build_tail(srb_nums)

    def forward(self, x):
        self._check_mode()
        P = self._trunk_pack()
        return self._trunk(self._head(x, P), P)


class _InfoGen(nn.Module):
    """parameter holder of tsrn.py:280-304"""

    def __init__(self, t_emb, output_size):
        super().__init__()
        self.tconv1 = nn.ConvTranspose2d(t_emb, 512, 3, 2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(512)
        self.tconv2 = nn.ConvTranspose2d(512, 128, 3, 2, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(128)
        self.tconv3 = nn.ConvTranspose2d(128, 64, 3, 2, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(64)
        self.tconv4 = nn.ConvTranspose2d(64, output_size, 3, (2, 1), padding=(1, 0), bias=False)
        self.bn4 = nn.BatchNorm2d(output_size)


class TSRN_TL(_PSNBase):
    """Drop-in for ``model.tsrn.TSRN_TL`` (tsrn.py:153-247; ``--arch tpgsr``), eval forward: TSRN whose residual blocks take a
    text-prior map -- the recogniser's (N, 37, 1, 26) probabilities through InfoGen (four ConvTranspose2d + BatchNorm + ReLU,
    tsrn.py:280-304) and a bilinear stretch to the LR feature map.  forward(x, text_emb) -> sr (N, 4, 2H, 2W).
    On the GPU the four transposed convolutions see a 1-pixel-high map, so only the middle kernel row acts: each is a 1 x 3
    convolution (flipped taps, eval BatchNorm folded, ReLU epilogue) of the zero-stuffed row through the implicit-GEMM kernel."""

    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32, word_vec_d=300,
                 text_emb=37, out_text_channels=32):
        super().__init__()
        self._build_trunk(scale_factor, width, height, STN, srb_nums, mask, hidden_units, out_text_channels)
        self.feature_enhancer = None
        self.infoGen = _InfoGen(text_emb, out_text_channels)
        self.emb_cls = text_emb
        self._build_tail(srb_nums)

    def _extra_pack(self, P):
        ig = self.infoGen
        for i in range(1, 5):
            w = getattr(ig, "tconv%d" % i).weight                      # (Cin, Cout, 3, 3)
            # H = 1 in, H = 1 out: only ky = 1 contributes; transposed conv along W = conv of the zero-stuffed row with flipped taps
            wc = w[:, :, 1, :].flip(2).permute(1, 0, 2).unsqueeze(2).contiguous()       # (Cout, Cin, 1, 3)
            cin = wc.shape[1]
            P["ig%d" % i] = packing.pack_conv(wc, None, _bn(getattr(ig, "bn%d" % i)), cin_pad=(cin + 3) // 4 * 4)

    def _info_gen(self, text_emb, H, W, P):
        """(N, 37, 1, 26) -> NHWC (N, H, W, 32)"""
        from .._abi import lib, check, dptr, stream
        N = text_emb.shape[0]
        x = text_emb.float().squeeze(2).transpose(1, 2)                # (N, 26, 37): layout plumbing
        cp = (x.shape[2] + 3) // 4 * 4
        cur = x.new_zeros(N, 1, x.shape[1], cp)
        cur[:, 0, :, :x.shape[2]] = x
        for i in range(1, 5):
            cout = getattr(self.infoGen, "tconv%d" % i).weight.shape[1]
            if i < 4:      # stride 2 along W, padding 1: zero-stuff to 2 Win - 1 columns, 1 x 3 conv with padding 1
                up = cur.new_zeros(N, 1, 2 * cur.shape[2] - 1, cur.shape[3])
                up[:, :, ::2] = cur
                cur = ops.conv2d([up], *P["ig%d" % i], cout, (1, 3), pad=(0, 1), epi_act="relu")
            else:          # stride 1 along W, padding 0: 1 x 3 conv with padding 2
                cur = ops.conv2d([cur], *P["ig%d" % i], cout, (1, 3), pad=(0, 2), epi_act="relu")
        out = torch.empty(N, H, W, cur.shape[3], device=cur.device)
        check(lib.dpmn_tl_interp_f32(dptr(cur), dptr(out), N, cur.shape[2], cur.shape[3], H, W, stream()))
        return out

    def forward(self, x, text_emb=None):
        self._check_mode()
        P = self._trunk_pack()
        if text_emb is None:
            text_emb = torch.zeros(x.shape[0], self.emb_cls, 1, 26, device=x.device)
        if text_emb.shape[0] != x.shape[0]:
            raise ValueError("TSRN_TL: text_emb batch must match the image batch")
        b1 = self._head(x, P)
        tp = self._info_gen(text_emb.to(x.device), x.shape[2], x.shape[3], P)
        return self._trunk(b1, P, tp)

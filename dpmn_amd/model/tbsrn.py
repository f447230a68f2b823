"""Host-side mirror of ``model/tbsrn.py::TBSRN`` (tbsrn.py:167-226), the PSN of config 3 -- eval mode, the only mode
DPMN runs a PSN in (super_resolution.py:56-59).  Same constructor, forward signature and ``state_dict`` keys (including
the reference's never-called members: ``conv``/``bn`` of the model and ``gru1``/``gru2`` of every block).

Per residual block (tbsrn.py:245-257): conv3x3+BN+mish, conv3x3+BN (eval BatchNorm folded, NHWC implicit GEMM), then the
FeatureEnhancer (tbsrn.py:76-92) over the 1024 positions of the 16x64 map.  NHWC makes the conv output the token matrix
already: conv2 writes channels 0..63 of a persistent (B, 1024, 128) buffer whose channels 64..127 hold the fixed 2-D
positional encoding, so neither the concat nor the two permutes of the reference exist here.  q/k/v projections are one
K=128 GEMM with stacked weights; attention is the flash-style ``dpmn_mha32_f32``; residual adds ride in GEMM epilogues.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .tsrn import _PSNBase, _GruBlock


class _LayerNormStd(nn.Module):
    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps


class _MultiHead(nn.Module):
    def __init__(self, h, d_model):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(d_model, d_model) for _ in range(4)])
        self.compress_attention_linear = nn.Linear(h, 1)     # present in the reference's state_dict, never used


class _PFF(nn.Module):
    def __init__(self, d_model, d_ff):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)


class _FeatureEnhancer(nn.Module):
    def __init__(self):
        super().__init__()
        self.multihead = _MultiHead(4, 128)
        self.mul_layernorm1 = _LayerNormStd(128)
        self.pff = _PFF(128, 128)
        self.mul_layernorm3 = _LayerNormStd(128)
        self.linear = nn.Linear(128, 64)


class _Block(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(ch)
        self.gru1 = _GruBlock(ch, ch)          # constructed by the reference, never called (tbsrn.py:236, 245-257)
        self.conv2 = nn.Conv2d(ch, ch, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(ch)
        self.gru2 = _GruBlock(ch, ch)
        self.feature_enhancer = _FeatureEnhancer()


def positionalencoding2d(d_model, height, width):
    """tbsrn.py:39-60: channels [0, d/2) encode the column (sin / cos interleaved), [d/2, d) the row."""
    pe = torch.zeros(d_model, height, width)
    half = d_model // 2
    div = torch.exp(torch.arange(0., half, 2) * -(math.log(10000.0) / half))
    pw = torch.arange(0., width)[:, None] * div
    ph = torch.arange(0., height)[:, None] * div
    pe[0:half:2] = torch.sin(pw).t()[:, None, :].expand(-1, height, -1)
    pe[1:half:2] = torch.cos(pw).t()[:, None, :].expand(-1, height, -1)
    pe[half::2] = torch.sin(ph).t()[:, :, None].expand(-1, -1, width)
    pe[half + 1::2] = torch.cos(ph).t()[:, :, None].expand(-1, -1, width)
    return pe


class TBSRN(_PSNBase):
    """Drop-in for ``model.tbsrn.TBSRN``, eval forward."""

    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32, input_channel=3):
        super().__init__()
        self.stn = bool(STN)       # held for checkpoint compatibility only (model/stn.py)
        self._stn_hw = (height // scale_factor, width // scale_factor)
        if hidden_units != 32 or (height, width) != (32, 128) or scale_factor != 2:
            raise NotImplementedError("dpmn_amd TBSRN: FeatureEnhancer is hard-wired to 64 channels on a 16x64 map "
                                      "(tbsrn.py:66-83): hidden_units=32, 32x128 output, scale 2")
        self.conv = nn.Conv2d(input_channel, 3, 3, 1, 1)      # reference members that forward() never touches
        self.bn = nn.BatchNorm2d(3)
        self.in_planes = 4 if mask else 3
        self.srb_nums = srb_nums
        self.ch = ch = 2 * hidden_units
        self.block1 = nn.Sequential(nn.Conv2d(self.in_planes, ch, 9, padding=4), nn.PReLU())
        for i in range(srb_nums):
            setattr(self, "block%d" % (i + 2), _Block(ch))
        self._build_tail(srb_nums)
        self._tok = {}

    def _pack_block_extra(self, P, i, blk):
        fe = blk.feature_enhancer
        ls = fe.multihead.linears
        P["fe%d.qkv" % i] = (torch.cat([ls[0].weight, ls[1].weight, ls[2].weight], 0).contiguous(),
                             torch.cat([ls[0].bias, ls[1].bias, ls[2].bias], 0).contiguous())

    def _tokens(self, B, dev):
        """(B, 16, 64, 128) token buffer: channels 64.. hold the positional encoding, 0..63 are rewritten by every block."""
        key = (B, dev)
        if key not in self._tok:
            pe = positionalencoding2d(64, 16, 64).permute(1, 2, 0)              # (16, 64, 64) NHWC
            buf = torch.zeros(B, 16, 64, 128, device=dev)
            buf[..., 64:] = pe.to(dev)
            self._tok[key] = buf
        return self._tok[key]

    def _block(self, x, P, i):
        B, H, W, Cc = x.shape
        blk = getattr(self, "block%d" % (i + 2))
        fe = blk.feature_enhancer
        M = B * H * W
        r = ops.conv2d([x], *P["srb%d.c1" % i], Cc, 3, pad=1, epi_act="mish")
        X = self._tokens(B, x.device)
        ops.conv2d([r], *P["srb%d.c2" % i], Cc, 3, pad=1, out=X, out_coff=0)
        Xt = X.view(M, 128)
        qkv = ops.linear(Xt, *P["fe%d.qkv" % i])
        att = ops.mha32(qkv, B, H * W, 4, 1.0 / math.sqrt(32.0))
        lo = fe.multihead.linears[3]
        y1 = ops.layernorm_std(ops.linear(att, lo.weight, lo.bias, res1=Xt), fe.mul_layernorm1.a_2, fe.mul_layernorm1.b_2,
                               fe.mul_layernorm1.eps)
        f = ops.linear(y1, fe.pff.w_1.weight, fe.pff.w_1.bias, act="relu")
        y2 = ops.layernorm_std(ops.linear(f, fe.pff.w_2.weight, fe.pff.w_2.bias, res1=y1), fe.mul_layernorm3.a_2,
                               fe.mul_layernorm3.b_2, fe.mul_layernorm3.eps)
        return ops.linear(y2, fe.linear.weight, fe.linear.bias, res1=x.view(M, Cc)).view(B, H, W, Cc)   # x + residual

    def forward(self, x):
        self._check_mode()
        P = self._trunk_pack()
        b1 = self._head(x, P)
        f = b1
        for i in range(self.srb_nums):
            f = self._block(f, P, i)
        return self._tail(b1, f, P)

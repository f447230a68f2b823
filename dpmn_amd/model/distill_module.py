"""Host-side mirror of ``model/distill_module.py::DistillModule`` (distill_module.py:4-31): two 3x3 convs + BatchNorm +
ReLU and an L1 between them; returns (loss, feature_cat).  Same state_dict keys.  All arithmetic runs in
libdpmn_hip.so through the NHWC implicit-GEMM conv (channels zero-padded 3 -> 4 / 6 -> 8 so rows stay 16-byte aligned)
with train-mode BatchNorm statistics from the conv epilogue; backward is explicit (dpmn_amd/train/cmm_train.Unit)."""
import torch
import torch.nn as nn

from .. import ops
from .._abi import lib, check, dptr, stream
from ..train import cmm_train as ct


class _PadConv(nn.Module):
    """view of a Conv2d with channels zero-padded for the kernels; maps gradients back."""

    def __init__(self, conv, in_map, cin_p):
        super().__init__()
        self.conv, self.in_map, self.cin_p = conv, in_map, cin_p

    def padded(self):
        w, b = self.conv.weight, self.conv.bias
        wp = w.new_zeros(4, self.cin_p, 3, 3)
        # in_map is made of runs of consecutive channels: slice copies only (a list index would be an H2D index tensor,
        # which cannot be captured into a hipGraph)
        src = 0
        while src < len(self.in_map):
            run = 1
            while src + run < len(self.in_map) and self.in_map[src + run] == self.in_map[src] + run:
                run += 1
            wp[:w.shape[0], self.in_map[src]:self.in_map[src] + run] = w[:, src:src + run]
            src += run
        bp = b.new_zeros(4)
        bp[:b.shape[0]] = b
        return wp, bp


class _PadBN:
    def __init__(self, bn):
        self.bn = bn
        self.eps, self.momentum = bn.eps, bn.momentum
        z = bn.weight.new_zeros(1)
        self.weight = torch.cat([bn.weight.detach(), z])
        self.bias = torch.cat([bn.bias.detach(), z])
        self.running_mean = torch.cat([bn.running_mean, z])
        self.running_var = torch.cat([bn.running_var, z + 1])
        self.num_batches_tracked = bn.num_batches_tracked

    def writeback(self):
        self.bn.running_mean.copy_(self.running_mean[:3])
        self.bn.running_var.copy_(self.running_var[:3])


class _Shim:
    """Conv2d-like holder of padded tensors for cmm_train.Unit."""

    def __init__(self, w, b):
        self.weight, self.bias = w, b


class _DistillFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, x_deep, x_shallow, *params):
        training = m.training
        d4 = ct.T(ops.nchw_to_nhwc(x_deep.contiguous().float(), 4))
        s4 = ct.T(ops.nchw_to_nhwc(x_shallow.contiguous().float(), 4))
        wc, bc = _PadConv(m.conv_cat_feature, [0, 1, 2, 4, 5, 6], 8).padded()
        wf, bf = _PadConv(m.conv_feature, [0, 1, 2], 4).padded()
        bn1, bn2 = _PadBN(m.bn_1), _PadBN(m.bn_2)
        u1 = ct.Unit("conv3", _Shim(wc, bc), bn1 if training else None, [d4, s4], "none")
        u2 = ct.Unit("conv3", _Shim(wf, bf), bn2 if training else None, [s4], "none")
        t1, t2 = u1.forward(), u2.forward()
        if training:
            bn1.writeback(); bn2.writeback()
        else:   # eval: running statistics as a fixed affine
            for t, bn in ((t1, bn1), (t2, bn2)):
                rstd = 1.0 / torch.sqrt(bn.running_var + bn.eps)
                t.scale = (bn.weight * rstd).contiguous()
                t.shift = (bn.bias - bn.running_mean * bn.weight * rstd).contiguous()
        B, H, W, _ = t1.r.shape
        pixels = B * H * W
        f1, f2 = torch.empty_like(t1.r), torch.empty_like(t2.r)
        RELU = ops.ACT["relu"]
        check(lib.dpmn_affine_act_fwd_f32(dptr(t1.r), dptr(t1.scale), dptr(t1.shift), RELU, dptr(f1), pixels, 4, stream()))
        check(lib.dpmn_affine_act_fwd_f32(dptr(t2.r), dptr(t2.scale), dptr(t2.shift), RELU, dptr(f2), pixels, 4, stream()))
        loss = torch.empty(1, device=f1.device)
        part = torch.empty((pixels * 4 + 255) // 256, device=f1.device)
        check(lib.dpmn_l1_loss_fwd_f32(dptr(f1), dptr(f2), 1.0 / (pixels * 3), dptr(loss), dptr(part), pixels * 4, stream()))
        feat = ops.nhwc_to_nchw(f1)[:, :3].contiguous()
        ctx.m, ctx.st = m, (u1, u2, t1, t2, f1, f2, d4, s4, bn1, bn2, training)
        ctx.need = (x_deep.requires_grad, x_shallow.requires_grad)
        return loss[0], feat

    @staticmethod
    def backward(ctx, dloss, dfeat):
        m = ctx.m
        u1, u2, t1, t2, f1, f2, d4, s4, bn1, bn2, training = ctx.st
        if not training:
            raise NotImplementedError("DistillModule: gradients through eval-mode BatchNorm are not built; use .train()")
        B, H, W, _ = t1.r.shape
        pixels = B * H * W
        extra = None
        if dfeat is not None:
            extra = ops.nchw_to_nhwc(dfeat.contiguous().float(), 4)
        df1, df2 = torch.empty_like(f1), torch.empty_like(f2)
        gs = dloss.reshape(1).float().contiguous()
        check(lib.dpmn_l1_loss_bwd_f32(dptr(f1), dptr(f2), dptr(gs), 1.0 / (pixels * 3), dptr(extra, True), dptr(df1), dptr(df2), pixels * 4, stream()))
        RELU = ops.ACT["relu"]
        for t, df in ((t1, df1), (t2, df2)):
            t.G = torch.empty_like(df)
            check(lib.dpmn_affine_act_bwd_f32(dptr(df), dptr(t.r), dptr(t.scale), dptr(t.shift), RELU, dptr(t.G), 0, pixels, 4, stream()))
        if not ctx.need[0]:
            d4.G = False
        gr = {}
        for u, bn in ((u1, bn1), (u2, bn2)):
            for tns in (u.conv.weight, u.conv.bias, bn.weight, bn.bias):
                gr[tns] = torch.zeros_like(tns)
        u1.backward(gr)
        u2.backward(gr)
        gw = gr[u1.conv.weight]
        g_wc = torch.cat([gw[:3, 0:3], gw[:3, 4:7]], dim=1).contiguous()   # slices, not a list index (hipGraph-capturable)
        g_wf = gr[u2.conv.weight][:3, :3].contiguous()
        dx_deep = ops.nhwc_to_nchw(d4.G)[:, :3].contiguous() if ctx.need[0] else None
        dx_sh = ops.nhwc_to_nchw(s4.G)[:, :3].contiguous() if ctx.need[1] else None
        by_param = {m.conv_cat_feature.weight: g_wc, m.conv_cat_feature.bias: gr[u1.conv.bias][:3].contiguous(),
                    m.bn_1.weight: gr[bn1.weight][:3].contiguous(), m.bn_1.bias: gr[bn1.bias][:3].contiguous(),
                    m.conv_feature.weight: g_wf, m.conv_feature.bias: gr[u2.conv.bias][:3].contiguous(),
                    m.bn_2.weight: gr[bn2.weight][:3].contiguous(), m.bn_2.bias: gr[bn2.bias][:3].contiguous()}
        ctx.st = None
        return (None, dx_deep, dx_sh) + tuple(by_param[p] for p in m.parameters())


class DistillModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv_cat_feature = nn.Conv2d(6, 3, 3, 1, 1)
        self.bn_1 = nn.BatchNorm2d(3)
        self.conv_feature = nn.Conv2d(3, 3, 3, 1, 1)
        self.bn_2 = nn.BatchNorm2d(3)

    def forward(self, x_deep, x_shallow):
        return _DistillFn.apply(self, x_deep, x_shallow, *list(self.parameters()))

"""Host-side mirror of ``model/distill_module.py::DistillModule`` (distill_module.py:4-31): conv_cat_feature (6 -> 3, 3x3) and
conv_feature (3 -> 3, 3x3) + BatchNorm2d(3) + ReLU and the L1 between the two feature maps; returns (loss, feature_cat).
Same constructor, forward signature and state_dict keys.  The torch layers are parameter holders that are never called: forward
and backward are the native module csrc/distill.hip (dpmn_distill_forward_f32 / dpmn_distill_backward_f32: 4 + 5 launches on the
NCHW images, every reduction a per-block partial row added in block order -- bitwise reproducible), bridged into autograd by one
torch.autograd.Function.  Gradients go straight into the trainer's flat gradient arena (direct mode, train/optim.py)."""
import ctypes as C

import torch
from .._abi import stream_of as _abi_stream_of
import torch.nn as nn

from .. import _abi
from .._abi import lib, check, dptr, stream

_WS = {}


def _workspace(dev, B, H, W):
    """one scratch per (device, stream, shape): statistics / loss rows, the raw-output gradient, the weight-gradient rows"""
    key = (dev.index, _abi_stream_of(dev), B, H, W)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(lib.dpmn_distill_workspace_bytes(B, H, W) // 4 + 64, device=dev)
    return ws


def _params(m):
    p = _abi.DistillParams()
    p.conv_cat_w, p.conv_cat_b = dptr(m.conv_cat_feature.weight), dptr(m.conv_cat_feature.bias)
    p.bn1_w, p.bn1_b, p.bn1_rm, p.bn1_rv = dptr(m.bn_1.weight), dptr(m.bn_1.bias), dptr(m.bn_1.running_mean), dptr(m.bn_1.running_var)
    p.bn1_nbt = m.bn_1.num_batches_tracked.data_ptr()
    p.conv_feat_w, p.conv_feat_b = dptr(m.conv_feature.weight), dptr(m.conv_feature.bias)
    p.bn2_w, p.bn2_b, p.bn2_rm, p.bn2_rv = dptr(m.bn_2.weight), dptr(m.bn_2.bias), dptr(m.bn_2.running_mean), dptr(m.bn_2.running_var)
    p.bn2_nbt = m.bn_2.num_batches_tracked.data_ptr()
    return p


def _forward(m, x_deep, x_shallow, training):
    xd, xs = x_deep.contiguous().float(), x_shallow.contiguous().float()
    if not xd.is_cuda:
        raise _abi.DpmnError("dpmn_amd DistillModule: tensors must live on the GPU (there is no CPU path)")
    B, c, H, W = xd.shape
    assert c == 3 and xs.shape == xd.shape, "DistillModule: (B, 3, H, W) images"
    for t in (m.conv_cat_feature.weight, m.conv_feature.weight):
        assert t.is_contiguous()
    r = torch.empty(B, 6, H, W, device=xd.device)
    state = torch.empty(24, device=xd.device)
    feat = torch.empty_like(xd)
    loss = torch.empty(1, device=xd.device)
    ws = _workspace(xd.device, B, H, W)
    check(lib.dpmn_distill_forward_f32(C.byref(_params(m)), dptr(xd), dptr(xs), int(training), dptr(r), dptr(state), dptr(feat), dptr(loss),
                                       dptr(ws), ws.numel() * 4, B, H, W, stream()))
    return loss, feat, (xd, xs, r, state)



def _plist(m):
    """list(m.parameters()), walked once per module (Parameter objects are never replaced on this path; same cache as
    train/pgrm_train.py params_of)."""
    ps = m.__dict__.get("_dpmn_plist")
    if ps is None:
        ps = m.__dict__["_dpmn_plist"] = list(m.parameters())
    return ps

class _DistillFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, x_deep, x_shallow, *params):
        loss, feat, saved = _forward(m, x_deep, x_shallow, m.training)
        ctx.m, ctx.saved, ctx.training = m, saved, m.training
        ctx.need = (x_deep.requires_grad, x_shallow.requires_grad)
        return loss[0], feat

    @staticmethod
    def backward(ctx, dloss, dfeat):
        from ..train.pgrm_train import grad_targets, finish_grads
        m = ctx.m
        if not ctx.training:
            raise NotImplementedError("DistillModule: gradients through eval-mode BatchNorm are not built; use .train()")
        xd, xs, r, state = ctx.saved
        B, _, H, W = xd.shape
        gr, direct = grad_targets(m)
        g = _abi.DistillGrads()
        g.dconv_cat_w, g.dconv_cat_b = dptr(gr[m.conv_cat_feature.weight]), dptr(gr[m.conv_cat_feature.bias])
        g.dbn1_w, g.dbn1_b = dptr(gr[m.bn_1.weight]), dptr(gr[m.bn_1.bias])
        g.dconv_feat_w, g.dconv_feat_b = dptr(gr[m.conv_feature.weight]), dptr(gr[m.conv_feature.bias])
        g.dbn2_w, g.dbn2_b = dptr(gr[m.bn_2.weight]), dptr(gr[m.bn_2.bias])
        gl = dloss.reshape(1).float().contiguous()
        df = None if dfeat is None else dfeat.contiguous().float()
        dxd = torch.empty_like(xd) if ctx.need[0] else None
        dxs = torch.empty_like(xs) if ctx.need[1] else None
        ws = _workspace(xd.device, B, H, W)
        check(lib.dpmn_distill_backward_f32(C.byref(_params(m)), C.byref(g), dptr(xd), dptr(xs), dptr(r), dptr(state), dptr(df, True), dptr(gl),
                                            dptr(dxd, True), dptr(dxs, True), dptr(ws), ws.numel() * 4, B, H, W, stream()))
        ctx.saved = None
        return (None, dxd, dxs) + finish_grads(m, gr, direct)


class DistillModule(nn.Module):
    direct_grad = True   # see train/optim.py (direct mode)

    def __init__(self):
        super().__init__()
        self.conv_cat_feature = nn.Conv2d(6, 3, 3, 1, 1)
        self.bn_1 = nn.BatchNorm2d(3)
        self.conv_feature = nn.Conv2d(3, 3, 3, 1, 1)
        self.bn_2 = nn.BatchNorm2d(3)

    def forward(self, x_deep, x_shallow):
        if torch.is_grad_enabled() and (any(p.requires_grad for p in _plist(self)) or x_deep.requires_grad or x_shallow.requires_grad):
            if getattr(self, "_dpmn_bucket", None) is not None:
                self._dpmn_bucket.note_use()
            from ..train.pgrm_train import fn_inputs
            return _DistillFn.apply(self, x_deep, x_shallow, *fn_inputs(self))
        loss, feat, _ = _forward(self, x_deep, x_shallow, self.training)
        return loss[0], feat

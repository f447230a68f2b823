"""Mirror of ``loss/image_loss.py::ImageLoss`` (MSE + L1 of gradient-magnitude maps) on HIP kernels with an explicit
backward (csrc/backward.hip), bridged into torch.autograd."""
import torch

from .. import ops
from .._abi import lib, check, dptr, stream


class _ImageLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, tgt, gradient, w_mse, w_grad):
        out_c, po, so = ops._nchw_view(out.float())
        tgt_c, pt, st = ops._nchw_view(tgt.float())
        B, Cc, H, W = out_c.shape
        loss = torch.empty(1, device=out.device)
        U = torch.empty(B, 3, H, W, device=out.device) if gradient else None
        V = torch.empty(B, 3, H, W, device=out.device) if gradient else None
        ws = torch.empty(lib.dpmn_image_loss_workspace_bytes(B, Cc, H, W), dtype=torch.uint8, device=out.device)
        check(lib.dpmn_image_loss_fwd_f32(po, so, pt, st, w_mse, w_grad, int(gradient), dptr(loss), dptr(U, True), dptr(V, True),
                                          ws.data_ptr(), B, Cc, H, W, stream()))
        ctx.save_for_backward(out_c, tgt_c, U, V)
        ctx.cfg = (gradient, w_mse, w_grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        out_c, tgt_c, U, V = ctx.saved_tensors
        gradient, w_mse, w_grad = ctx.cfg
        B, Cc, H, W = out_c.shape
        _, po, so = ops._nchw_view(out_c)
        _, pt, st = ops._nchw_view(tgt_c)
        grad = torch.empty(B, Cc, H, W, device=out_c.device)
        gs = g.reshape(1).float().contiguous()
        check(lib.dpmn_image_loss_bwd_f32(po, so, pt, st, dptr(U, True), dptr(V, True), dptr(gs), w_mse, w_grad, int(gradient),
                                          dptr(grad), 0, B, Cc, H, W, stream()))
        return grad, None, None, None, None


class ImageLoss(torch.nn.Module):
    def __init__(self, gradient=True, loss_weight=[20, 1e-4]):
        super().__init__()
        self.gradient = gradient
        self.loss_weight = loss_weight

    def forward(self, out_images, target_images):
        return _ImageLossFn.apply(out_images, target_images, bool(self.gradient), float(self.loss_weight[0]), float(self.loss_weight[1]))

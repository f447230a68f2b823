"""Row a15: the PSN's spatial-transformer front end (``model/stn_head.py::STNHead`` 26-106 and
``model/tps_spatial_transformer.py::TPSSpatialTransformer`` 54-112) as drop-in modules over libdpmn_hip.so.

The reference trains and tests with ``--STN`` (README.md:34,42), so released PSN checkpoints carry ``tps.*`` and
``stn_head.*`` entries -- but the branch only executes under ``self.training`` (tsrn.py:62, tatt.py:75-78, tbsrn.py:215)
and DPMN keeps the PSN in ``.eval()`` (super_resolution.py:56-59).  Inside DPMN these classes therefore only hold the
reference's state_dict layout and constructor values; called on their own they run the reference's forward (train mode:
BatchNorm batch statistics + running-stat updates; eval mode: BatchNorm folded into the convs): NHWC implicit-GEMM convs,
``dpmn_maxpool_f32``, ``dpmn_stn_fc_f32`` and ``dpmn_tps_sample_f32``.  Forward only (the PSN is frozen: no backward).
"""
import itertools
import math

import numpy as np
import torch
import torch.nn as nn


def _conv3x3_block(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class STNHead(nn.Module):
    def __init__(self, in_planes, num_ctrlpoints, activation='none', input_size=(16, 64)):
        super().__init__()
        self.in_planes, self.num_ctrlpoints, self.activation = in_planes, num_ctrlpoints, activation
        self.stn_convnet = nn.Sequential(
            _conv3x3_block(in_planes, 32), nn.MaxPool2d(2, 2), _conv3x3_block(32, 64), nn.MaxPool2d(2, 2),
            _conv3x3_block(64, 128), nn.MaxPool2d(2, 2), _conv3x3_block(128, 256), nn.MaxPool2d(2, 2),
            _conv3x3_block(256, 256), nn.MaxPool2d((1, 2), (1, 2)), _conv3x3_block(256, 256))
        self.stn_fc1 = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True))
        self.stn_fc2 = nn.Linear(512, num_ctrlpoints * 2)
        with torch.no_grad():
            for seq in (self.stn_convnet, self.stn_fc1):          # stn_head.py:57-69
                for m in seq.modules():
                    if isinstance(m, nn.Conv2d):
                        m.weight.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
                        m.bias.zero_()
                    elif isinstance(m, nn.BatchNorm2d):
                        m.weight.fill_(1)
                        m.bias.zero_()
                    elif isinstance(m, nn.Linear):
                        m.weight.normal_(0, 0.001)
                        m.bias.zero_()
            # stn_head.py:71-90: the last layer starts as the identity warp (control points on the top / bottom edges)
            margin, half = 0.01, num_ctrlpoints // 2
            xs = np.linspace(margin, 1. - margin, half)
            pts = np.concatenate([np.stack([xs, np.full(half, margin)], 1), np.stack([xs, np.full(half, 1 - margin)], 1)], 0)
            pts = pts.astype(np.float32)
            if activation == 'sigmoid':
                pts = -np.log(1. / pts - 1.)
            self.stn_fc2.weight.zero_()
            self.stn_fc2.bias.copy_(torch.from_numpy(pts).reshape(-1))

    def forward(self, x):
        """(B, in_planes, 16, 64) -> (img_feat (B,512), ctrl_points (B, num_ctrlpoints, 2))   (stn_head.py:92-106)"""
        from .. import ops
        from . import packing
        from ..train.cmm_train import _bn_finalize, stats_buffer
        if self.activation != 'none':
            raise NotImplementedError("dpmn_amd STNHead: activation 'none' is what every PSN constructs (tatt.py:66-70)")
        B, cin, H, W = x.shape
        if (H // 16) * (W // 32) * 256 != 512 or H % 16 or W % 32:
            raise RuntimeError("STNHead: stn_fc1 expects 512 flattened features, got %d (input %dx%d; quirk Q10: tsrn.py feeds "
                               "32x64 and fails the same way)" % ((H // 16) * (W // 32) * 256, H, W))
        cur = ops.nchw_to_nhwc(x.contiguous().float(), (cin + 3) // 4 * 4)
        pools = [(2, 2)] * 4 + [(1, 2)]
        last = None
        for bi in range(6):
            conv, bn = self.stn_convnet[2 * bi][0], self.stn_convnet[2 * bi][1]
            cout = conv.out_channels
            if self.training:       # batch statistics from the conv epilogue; affine + ReLU applied by the consumer on load
                stats = stats_buffer(cout, cur.device)
                r = ops.conv2d([cur], packing.tpack_conv(conv.weight, cin_pad=cur.shape[3]), conv.bias, cout, 3, pad=1, stats=stats)
                aff = _bn_finalize(stats, bn, r.shape[0] * r.shape[1] * r.shape[2])[:2]
            else:                   # running statistics folded into the conv, ReLU in its epilogue
                wp, b = packing.pack_conv(conv.weight, conv.bias, bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps),
                                          cin_pad=cur.shape[3])
                r = ops.conv2d([cur], wp, b, cout, 3, pad=1, epi_act="relu")
                aff = None
            if bi < 5:
                cur = ops.maxpool(r, *pools[bi], *(aff if aff is not None else (None, None)))
            else:
                last = (r, aff)
        fc1, bn1 = self.stn_fc1[0], self.stn_fc1[1]
        feat, ctrl = ops.stn_fc(last[0], last[1], fc1.weight.t().contiguous(), fc1.bias, bn1, self.training, self.stn_fc2.weight,
                                self.stn_fc2.bias)
        if self.training:
            bn1.num_batches_tracked += 1
        return feat, ctrl.view(-1, self.num_ctrlpoints, 2)


def _partial_repr(points, ctrl):
    """phi(r) = r^2 log r written as 0.5 * d2 * log(d2) with 0 log 0 := 0 (tps_spatial_transformer.py:22-34)."""
    d = points[:, None, :] - ctrl[None, :, :]
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
    r = 0.5 * d2 * torch.log(d2)
    return torch.where(torch.isnan(r), torch.zeros_like(r), r)


class TPSSpatialTransformer(nn.Module):
    def __init__(self, output_image_size=None, num_control_points=None, margins=None):
        super().__init__()
        self.output_image_size, self.num_control_points, self.margins = output_image_size, num_control_points, margins
        H, W = output_image_size
        half = num_control_points // 2
        xs = np.linspace(margins[0], 1.0 - margins[0], half)
        pts = np.concatenate([np.stack([xs, np.full(half, margins[1])], 1),
                              np.stack([xs, np.full(half, 1.0 - margins[1])], 1)], 0)
        ctrl = torch.Tensor(pts)
        N = num_control_points
        fk = torch.zeros(N + 3, N + 3)
        fk[:N, :N] = _partial_repr(ctrl, ctrl)
        fk[:N, -3] = 1
        fk[-3, :N] = 1
        fk[:N, -2:] = ctrl
        fk[-2:, :N] = ctrl.t()
        coord = torch.Tensor(list(itertools.product(range(H), range(W))))       # (y, x)
        Y, X = coord[:, :1] / (H - 1), coord[:, 1:] / (W - 1)
        tc = torch.cat([X, Y], 1)
        self.register_buffer('inverse_kernel', torch.inverse(fk).contiguous())
        self.register_buffer('padding_matrix', torch.zeros(3, 2))
        self.register_buffer('target_coordinate_repr', torch.cat([_partial_repr(tc, ctrl), torch.ones(H * W, 1), tc], 1))
        self.register_buffer('target_control_points', ctrl)

    def forward(self, input, source_control_points):
        """(B,C,Hin,Win), (B,N,2) -> (warped (B,C,H,W), source_coordinate (B,H*W,2))   (tps_spatial_transformer.py:97-112)"""
        from .. import ops
        assert source_control_points.ndimension() == 3
        assert source_control_points.size(1) == self.num_control_points
        assert source_control_points.size(2) == 2
        return ops.tps_sample(input.contiguous().float(), source_control_points.contiguous().float(), self.inverse_kernel.contiguous(),
                              self.target_coordinate_repr.contiguous(), self.output_image_size)

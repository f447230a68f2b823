"""Host-side mirror of the reference's VisionLAN recogniser (model/VisionLAN/VisionLAN.py), the text-prior generator of
DPMN's branch 1 (interfaces/super_resolution.py:100-111, 174-199), restricted to what that path runs: the inference branch
(``Train_in=False``: backbone -> SequenceModeling -> Prediction.pp -> w_vrm -> length decode) in eval mode, BATCHED.

Same constructor, module tree and ``state_dict`` keys as the reference (so ``visionlan.pth`` / ``recognizer_best_{k}.pth``
load unchanged); the ``torch.nn`` layers below are parameter holders that are never called -- all arithmetic runs in
libdpmn_hip.so: ResNet45 as NHWC implicit-GEMM / halo convs with the eval BatchNorm folded and the residual + ReLU in the
epilogue, the three encoder layers as stacked q|k|v GEMM + ``dpmn_mha64_f32`` + GEMM/LayerNorm chains, the position
attention and output layer in ``dpmn_vl_pp_pool_f32``.  The MLM branch (training-only, VisionLAN.py:12-45) is held for the
checkpoint layout and not executed.  Train-mode behaviour of the reference loop (the student recognisers are put in
.train() while being called through the inference branch: batch-1 BatchNorm statistics, live dropout,
super_resolution.py:131-138) is NOT reproduced -- it is not a function of the inputs alone.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from . import packing

LAYERS = [3, 4, 6, 6, 3]
PLANES = [32, 64, 128, 256, 512]
DICT36 = "abcdefghijklmnopqrstuvwxyz1234567890"      # dic_36.txt (class c > 0 -> DICT36[c - 1]; class 0 = end of string)


class _BasicBlock(nn.Module):       # resnet.py:13-37
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class _ResNet45(nn.Module):         # resnet.py:39-112 (compress_layer=False)
    def __init__(self, strides):
        super().__init__()
        self.strides = [tuple(s) for s in strides]
        self.conv1_new = nn.Conv2d(3, 32, 3, self.strides[0], 1, bias=False)
        self.bn1 = nn.BatchNorm2d(32)
        inplanes = 32
        for li, (n, planes) in enumerate(zip(LAYERS, PLANES)):
            stride = self.strides[li + 1]
            blocks = []
            for bi in range(n):
                st = stride if bi == 0 else (1, 1)
                ds = None
                if bi == 0 and (st != (1, 1) or inplanes != planes):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes, 1, st, bias=False), nn.BatchNorm2d(planes))
                blocks.append(_BasicBlock(inplanes, planes, st, ds))
                inplanes = planes
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))


def _pos_table(n_position, d):
    pos = np.arange(n_position)[:, None].astype(np.float64)
    ang = pos / np.power(10000, 2 * (np.arange(d)[None, :] // 2) / d)
    tab = np.zeros((n_position, d))
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(tab).float().unsqueeze(0)


class _PosEnc(nn.Module):
    def __init__(self, d, n_position):
        super().__init__()
        self.register_buffer("pos_table", _pos_table(n_position, d))


class _MHA(nn.Module):              # modules.py:43-58
    def __init__(self, n_head, d_model, d_k, d_v):
        super().__init__()
        self.w_qs, self.w_ks, self.w_vs = nn.Linear(d_model, n_head * d_k), nn.Linear(d_model, n_head * d_k), nn.Linear(d_model, n_head * d_v)
        self.layer_norm = nn.LayerNorm(d_model)
        self.fc = nn.Linear(n_head * d_v, d_model)


class _FFN(nn.Module):
    def __init__(self, d_in, d_hid):
        super().__init__()
        self.w_1, self.w_2 = nn.Conv1d(d_in, d_hid, 1), nn.Conv1d(d_hid, d_in, 1)
        self.layer_norm = nn.LayerNorm(d_in)


class _EncLayer(nn.Module):
    def __init__(self, d_model, d_inner, n_head, d_k, d_v):
        super().__init__()
        self.slf_attn = _MHA(n_head, d_model, d_k, d_v)
        self.pos_ffn = _FFN(d_model, d_inner)


class _Encoder(nn.Module):          # Transforme_Encoder, modules.py:112-131
    def __init__(self, n_layers, n_position=256, d_model=512, d_inner=2048, n_head=8, d_k=64):
        super().__init__()
        self.position_enc = _PosEnc(d_model, n_position)
        self.layer_stack = nn.ModuleList([_EncLayer(d_model, d_inner, n_head, d_k, d_k) for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.n_head, self.d_k = n_head, d_k


class _PP(nn.Module):               # PP_layer, modules.py:153-172
    def __init__(self, n_dim=512, N_max_character=26, n_position=256):
        super().__init__()
        self.f0_embedding = nn.Embedding(N_max_character, n_dim)
        self.w0 = nn.Linear(N_max_character, n_position)
        self.wv = nn.Linear(n_dim, n_dim)
        self.we = nn.Linear(n_dim, N_max_character)


class _Prediction(nn.Module):       # modules.py:174-202
    def __init__(self, n_dim=512, n_class=37, N_max_character=26, n_position=256):
        super().__init__()
        self.pp = _PP(n_dim, N_max_character, n_position)
        self.pp_share = _PP(n_dim, N_max_character, n_position)
        self.w_vrm = nn.Linear(n_dim, n_class)
        self.w_share = nn.Linear(n_dim, n_class)


class _MLM(nn.Module):              # VisionLAN.py:12-25 (training only; held for the checkpoint layout)
    def __init__(self, n_dim=512):
        super().__init__()
        self.MLM_SequenceModeling_mask = _Encoder(2)
        self.MLM_SequenceModeling_WCL = _Encoder(1)
        self.pos_embedding = nn.Embedding(25, 512)
        self.w0_linear = nn.Linear(1, 256)
        self.wv = nn.Linear(n_dim, n_dim)
        self.we = nn.Linear(n_dim, 1)


class _MLM_VRM(nn.Module):
    def __init__(self):
        super().__init__()
        self.MLM = _MLM()
        self.SequenceModeling = _Encoder(3)
        self.Prediction = _Prediction(n_position=256, N_max_character=26, n_class=37)
        self.nclass = 37


class VisionLAN(nn.Module):
    """Drop-in for ``model.VisionLAN.VisionLAN.VisionLAN(strides, input_shape)``; ``forward(input, label_pos, training_stp,
    Train_in=False)`` returns the reference's ``(output, out_length)`` pair for a whole batch; ``recognise(images)`` is the
    batched form the text-prior path uses (logits, classes, lengths stay on the GPU)."""

    def __init__(self, strides=((1, 1), (2, 2), (2, 2), (2, 2), (1, 1), (1, 1)), input_shape=(3, 64, 256)):
        super().__init__()
        self.backbone = _ResNet45(strides)
        self.input_shape = list(input_shape)
        self.MLM_VRM = _MLM_VRM()
        self._packed = None

    # ------------------------------------------------------------------ weight packs (rebuilt when a parameter moves or changes)
    def _packs(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + tuple((b.data_ptr(), b._version) for b in self.buffers())
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        bn = lambda pre: packing.bn_tuple(sd, pre)
        P = {"conv1": packing.pack_conv(sd["backbone.conv1_new.weight"], None, bn("backbone.bn1."), cin_pad=4), "blocks": []}
        for li, n in enumerate(LAYERS):
            for bi in range(n):
                p = "backbone.layer%d.%d." % (li + 1, bi)
                blk = getattr(self.backbone, "layer%d" % (li + 1))[bi]
                e = dict(stride=blk.stride, c1=packing.pack_conv(sd[p + "conv1.weight"], None, bn(p + "bn1.")),
                         c2=packing.pack_conv(sd[p + "conv2.weight"], None, bn(p + "bn2.")), ds=None,
                         planes=sd[p + "conv2.weight"].shape[0])
                if blk.downsample is not None:
                    e["ds"] = packing.pack_conv(sd[p + "downsample.0.weight"], None, bn(p + "downsample.1."))
                P["blocks"].append(e)
        pre = "MLM_VRM.SequenceModeling."
        P["pos"] = sd[pre + "position_enc.pos_table"][0].contiguous()
        P["layers"] = []
        for i in range(3):
            a, f = "%slayer_stack.%d.slf_attn." % (pre, i), "%slayer_stack.%d.pos_ffn." % (pre, i)
            P["layers"].append(dict(
                wqkv=torch.cat([sd[a + "w_qs.weight"], sd[a + "w_ks.weight"], sd[a + "w_vs.weight"]], 0).contiguous(),
                bqkv=torch.cat([sd[a + "w_qs.bias"], sd[a + "w_ks.bias"], sd[a + "w_vs.bias"]], 0).contiguous(),
                fc_w=sd[a + "fc.weight"].contiguous(), fc_b=sd[a + "fc.bias"], ln1=(sd[a + "layer_norm.weight"], sd[a + "layer_norm.bias"]),
                w1=sd[f + "w_1.weight"].squeeze(-1).contiguous(), b1=sd[f + "w_1.bias"], w2=sd[f + "w_2.weight"].squeeze(-1).contiguous(),
                b2=sd[f + "w_2.bias"], ln2=(sd[f + "layer_norm.weight"], sd[f + "layer_norm.bias"])))
        P["ln_out"] = (sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"])
        pp = "MLM_VRM.Prediction.pp."
        # t = w0(f0_embedding^T): (512, 256), a function of the weights only (modules.py:164-167) -> evaluated at pack time on
        # the library's own GEMM: rows = channels, K = 26 characters padded to 32
        emb_t = torch.zeros(512, 32, device=sd[pp + "w0.weight"].device)
        emb_t[:, :26] = sd[pp + "f0_embedding.weight"].t()
        w0p = torch.zeros(256, 32, device=emb_t.device)
        w0p[:, :26] = sd[pp + "w0.weight"]
        P["T"] = ops.linear(emb_t, w0p, sd[pp + "w0.bias"].contiguous()).t().contiguous()        # (256 positions, 512)
        P["wv"], P["bv"] = sd[pp + "wv.weight"].contiguous(), sd[pp + "wv.bias"]
        we = torch.zeros(28, 512, device=emb_t.device)
        we[:26] = sd[pp + "we.weight"]
        be = torch.zeros(28, device=emb_t.device)
        be[:26] = sd[pp + "we.bias"]
        P["we"], P["be"] = we, be
        P["w_vrm"], P["b_vrm"] = sd["MLM_VRM.Prediction.w_vrm.weight"].contiguous(), sd["MLM_VRM.Prediction.w_vrm.bias"].contiguous()
        P["Texp"] = {}
        self._packed = (key, P)
        return P

    # ------------------------------------------------------------------ batched inference
    @torch.no_grad()
    def features(self, x_nhwc4):
        """ResNet45 on NHWC input with 4 channels (channel 3 ignored: zero weights) -> (B, 8, 32, 512)."""
        P = self._packs()
        w, b = P["conv1"]
        x = ops.conv2d([x_nhwc4], w, b, 32, 3, stride=self.backbone.strides[0][0], pad=1, epi_act="relu")
        for e in P["blocks"]:
            st = e["stride"][0]
            out = ops.conv2d([x], e["c1"][0], e["c1"][1], e["planes"], 1, epi_act="relu")
            res = x if e["ds"] is None else ops.conv2d([x], e["ds"][0], e["ds"][1], e["planes"], 1, stride=st)
            x = ops.conv2d([out], e["c2"][0], e["c2"][1], e["planes"], 3, stride=st, pad=1, epi_act="relu_post_res", res=res)
        return x

    @torch.no_grad()
    def logits_from_features(self, feat):
        P = self._packs()
        B = feat.shape[0]
        tok = ops.vl_tokens(feat, P["pos"])                       # (B, 256, 512), + positional table
        L, D = tok.shape[1], tok.shape[2]
        x = tok.reshape(B * L, D)
        for lw in P["layers"]:
            qkv = ops.linear(x, lw["wqkv"], lw["bqkv"])
            att = ops.mha64(qkv, B, L, 8, 1.0 / math.sqrt(64.0))
            x = ops.layernorm(ops.linear(att, lw["fc_w"], lw["fc_b"], res1=x), *lw["ln1"])
            h = ops.linear(x, lw["w1"], lw["b1"], act="relu")
            x = ops.layernorm(ops.linear(h, lw["w2"], lw["b2"], res1=x), *lw["ln2"])
        enc = ops.layernorm(x, *P["ln_out"], eps=1e-6)
        if B not in P["Texp"]:
            P["Texp"] = {B: P["T"].repeat(B, 1).contiguous()}
        z = ops.linear(enc, P["wv"], P["bv"], res1=P["Texp"][B])
        scores = ops.linear(ops.act(z, "tanh"), P["we"], P["be"])          # (B*256, 28): column n = reading position n
        return ops.vl_pp_pool(scores, enc.reshape(B, L, D), P["w_vrm"], P["b_vrm"], 26)

    @torch.no_grad()
    def recognise(self, images):
        """images: (B, >=3, H, W) floats in [0, 1] at any size (resized like parse_visionlan_data), or NHWC4 (B, 64, 256, 4).
        Returns (logits (B, 26, 37), classes (B, 25) int32, lengths (B) int32), all on the GPU."""
        if self.training:
            raise RuntimeError("dpmn_amd VisionLAN: only the eval-mode inference branch is built (see the module docstring)")
        x = images if images.shape[-1] == 4 and images.dim() == 4 and images.shape[1] == self.input_shape[1] else \
            ops.vl_resize(images, self.input_shape[1], self.input_shape[2])
        lg = self.logits_from_features(self.features(x))
        cls, length = ops.vl_decode(lg, 25)
        return lg, cls, length

    def forward(self, input, label_pos=None, training_stp='', Train_in=False):
        """The reference's call signature (VisionLAN.py:156); only Train_in=False exists here.  input: (B, 3, 64, 256)."""
        if Train_in:
            raise NotImplementedError("dpmn_amd VisionLAN: the MLM / training branches are out of the SR hot path")
        x4 = ops.nchw_to_nhwc(input.contiguous().float(), 4)
        lg, cls, length = self.recognise(x4)
        n = length.long()
        rows = torch.cat([lg[b, :int(n[b])] for b in range(lg.shape[0])])      # (sum lengths, 37), like MLM_VRM.forward 127-135
        return rows, length.float()


def decode_strings(cls, length):
    """cha_encdec.decode (model/VisionLAN/utils.py:30-38) on the decoded classes: list of strings."""
    c, n = cls.cpu().tolist(), length.cpu().tolist()
    return ["".join(DICT36[k - 1] if 0 < k <= 36 else "" for k in c[b][:n[b]]) for b in range(len(n))]

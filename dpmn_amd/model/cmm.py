"""Host-side mirror of the reference's ``model/cmm.py`` (ComplementationModulationModule).

Same constructor, ``forward(x1, x2)`` and ``state_dict`` key layout as cmm.py:80-161; parameters are
held by stock ``torch.nn`` containers that are NEVER called -- every convolution runs through the
NHWC implicit-GEMM kernel in libdpmn_hip.so (csrc/conv.hip) and the channel gate through
csrc/cmm.hip.  Activations stay NHWC between kernels; only the module boundary is NCHW.
Eval-mode BatchNorm is folded into the producing conv at pack time (quirk Q12).
"""
import os

import torch
import torch.nn as nn

from .. import ops
from . import packing


def _encode_block(cin, cout):
    # indices match cmm.py:38-55: [act, conv4x4 s2 d2, bn, act, conv3x3, bn]
    return nn.Sequential(nn.Identity(), nn.Conv2d(cin, cin, 4, 2, dilation=2, padding=3), nn.BatchNorm2d(cin),
                         nn.Identity(), nn.Conv2d(cin, cout, 3, 1, padding=1), nn.BatchNorm2d(cout))


def _decode_block(cin, cout):
    # cmm.py:58-77: [act, convT3x3, bn, act, convT4x4 s2, bn]
    return nn.Sequential(nn.Identity(), nn.ConvTranspose2d(cin, cout, 3, 1, padding=1), nn.BatchNorm2d(cout),
                         nn.Identity(), nn.ConvTranspose2d(cout, cout, 4, 2, padding=1), nn.BatchNorm2d(cout))


NATIVE_FORWARD = os.environ.get("DPMN_CMM_NATIVE", "1") != "0"      # eval forward through dpmn_cmm_forward_f32 (one native call)



def _plist(m):
    """list(m.parameters()), walked once per module (Parameter objects are never replaced on this path; same cache as
    train/pgrm_train.py params_of)."""
    ps = m.__dict__.get("_dpmn_plist")
    if ps is None:
        ps = m.__dict__["_dpmn_plist"] = list(m.parameters())
    return ps

class _Holder(nn.Module):
    def __init__(self, seq, name):
        super().__init__()
        setattr(self, name, seq)


class ComplementationModulationModule(nn.Module):
    direct_grad = True   # see train/optim.py (direct mode)
    def __init__(self, c_img=3, norm='batch', act_en='leaky_relu', act_de='relu', cnum=64):
        super().__init__()
        if norm != 'batch' or act_en != 'leaky_relu' or act_de != 'relu':
            raise NotImplementedError("dpmn_amd CMM: built for the only configuration the trainer constructs "
                                      "(norm='batch', act_en='leaky_relu', act_de='relu'; super_resolution.py:72)")
        self.c_img, self.cnum = c_img, cnum
        c = cnum
        for br in ("1", "2"):
            setattr(self, "en_1_" + br, nn.Conv2d(c_img, c, 3, 1, padding=1))
            setattr(self, "en_2_" + br, _Holder(_encode_block(c, 2 * c), "encode"))
            setattr(self, "en_3_" + br, _Holder(_encode_block(2 * c, 4 * c), "encode"))
            setattr(self, "en_4_" + br, _Holder(_encode_block(4 * c, 8 * c), "encode"))
            setattr(self, "en_5_" + br, _Holder(_encode_block(8 * c, 8 * c), "encode"))
            setattr(self, "en_6_" + br, nn.Sequential(nn.Identity(), nn.Conv2d(8 * c, 8 * c, 4, 2, padding=1)))
        self.fc_1 = nn.Linear(16 * c, 4 * c)
        self.fc_2 = nn.Linear(4 * c, 16 * c)
        self.de_6 = nn.Sequential(nn.Identity(), nn.ConvTranspose2d(16 * c, 8 * c, 4, 2, padding=1), nn.BatchNorm2d(8 * c))
        self.de_5 = _Holder(_decode_block(24 * c, 8 * c), "decode")
        self.de_4 = _Holder(_decode_block(24 * c, 4 * c), "decode")
        self.de_3 = _Holder(_decode_block(12 * c, 2 * c), "decode")
        self.de_2 = _Holder(_decode_block(6 * c, c), "decode")
        self.de_1 = nn.Sequential(nn.Identity(), nn.ConvTranspose2d(3 * c, c_img, 3, 1, padding=1))
        self._pack = None

    @staticmethod
    def _bn(m):
        return (m.weight, m.bias, m.running_mean, m.running_var, m.eps)

    def exchange_segments(self):
        """The parameters in the order their gradients are finished by the backward (train/cmm_train.py walks the graph in reverse),
        cut where a gradient collective may start (train/optim.py SegmentedBucket): the decoder levels de_1 ... de_5, then de_6 + the
        channel gate, then the twin encoder levels deepest first -- 48 / 35 / 34 / 52 / 45 MB at cnum = 64."""
        mods = lambda *names: [p for n in names for p in getattr(self, n).parameters()]
        return [mods("de_1", "de_2", "de_3", "de_4", "de_5"), mods("de_6", "fc_2", "fc_1"), mods("en_6_1", "en_6_2"), mods("en_5_1", "en_5_2"),
                mods("en_4_1", "en_4_2", "en_3_1", "en_3_2", "en_2_1", "en_2_2", "en_1_1", "en_1_2")]

    def _packed(self):
        key = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._pack is not None and self._pack[0] == key:
            return self._pack[1]
        P = {}
        with torch.no_grad():
            for br in ("1", "2"):
                e1 = getattr(self, "en_1_" + br)
                P["en_1_" + br] = packing.pack_conv(e1.weight, e1.bias, cin_pad=4)
                for lvl in (2, 3, 4, 5):
                    seq = getattr(self, "en_%d_%s" % (lvl, br)).encode
                    P["en_%d_%s.a" % (lvl, br)] = packing.pack_conv(seq[1].weight, seq[1].bias, self._bn(seq[2]))
                    P["en_%d_%s.b" % (lvl, br)] = packing.pack_conv(seq[4].weight, seq[4].bias, self._bn(seq[5]))
                e6 = getattr(self, "en_6_" + br)[1]
                P["en_6_" + br] = packing.pack_conv(e6.weight, e6.bias)
            # the twin encoder branches (cmm.py:86-99: same shapes, own weights) run as ONE grouped launch per level over the
            # batch-concatenated inputs: stacked (2, Cout, Kp) weights / (2, Cout) biases
            for name in ["en_1"] + ["en_%d.%s" % (lvl, ab) for lvl in (2, 3, 4, 5) for ab in "ab"] + ["en_6"]:
                k1, k2 = (name.replace(".", "_1."), name.replace(".", "_2.")) if "." in name else (name + "_1", name + "_2")
                P[name] = (torch.stack([P[k1][0], P[k2][0]]).contiguous(), torch.stack([P[k1][1], P[k2][1]]).contiguous())
                del P[k1], P[k2]
            P["de_6"] = ops.stack_phase_packs(packing.pack_convT_s2k4(self.de_6[1].weight, self.de_6[1].bias, self._bn(self.de_6[2])))
            for lvl in (5, 4, 3, 2):
                seq = getattr(self, "de_%d" % lvl).decode
                P["de_%d.a" % lvl] = packing.pack_convT_s1(seq[1].weight, seq[1].bias, self._bn(seq[2]))
                P["de_%d.b" % lvl] = ops.stack_phase_packs(packing.pack_convT_s2k4(seq[4].weight, seq[4].bias, self._bn(seq[5])))
            P["de_1"] = packing.pack_convT_s1(self.de_1[1].weight, self.de_1[1].bias)
        self._pack = (key, P)
        return P

    def _native(self, P, H, W):
        """dpmn_cmm_weights over the packed eval weights (include/dpmn_hip.h), rebuilt when the pack is."""
        from .. import _abi
        nat = getattr(self, "_nat", None)
        if nat is not None and nat[0] is P and nat[1] == (H, W):
            return nat[2], nat[3]
        w = _abi.CmmWeights()
        w.c_img, w.cnum, w.img_h, w.img_w = self.c_img, self.cnum, H, W
        names = ["en_1"] + ["en_%d.%s" % (lvl, ab) for lvl in (2, 3, 4, 5) for ab in "ab"] + ["en_6"]
        keep = [self.fc_1.weight, self.fc_1.bias, self.fc_2.weight, self.fc_2.bias]
        for i, n in enumerate(names):
            w.en_w[i], w.en_b[i] = P[n][0].data_ptr(), P[n][1].data_ptr()
        w.fc1_w, w.fc1_b, w.fc2_w, w.fc2_b = (t.data_ptr() for t in keep)
        w.de6_w, w.de6_b = P["de_6"][0].data_ptr(), P["de_6"][1].data_ptr()
        for j, lvl in enumerate((5, 4, 3, 2)):
            w.dea_w[j], w.dea_b[j] = P["de_%d.a" % lvl][0].data_ptr(), P["de_%d.a" % lvl][1].data_ptr()
            w.deb_w[j], w.deb_b[j] = P["de_%d.b" % lvl][0].data_ptr(), P["de_%d.b" % lvl][1].data_ptr()
        w.de1_w, w.de1_b = P["de_1"][0].data_ptr(), P["de_1"][1].data_ptr()
        self._nat = (P, (H, W), w, keep)
        if not hasattr(self, "_native_ws"):
            self._native_ws = {}
        return w, keep

    def forward(self, x1, x2):
        if self.training:
            from ..train import cmm_train       # train-mode BatchNorm (batch statistics) + explicit HIP backward
            if torch.is_grad_enabled():
                return cmm_train.apply(self, x1, x2)
            with torch.no_grad():
                return cmm_train.build(self, x1, x2)[0]
        if torch.is_grad_enabled() and any(p.requires_grad for p in _plist(self)):
            raise NotImplementedError("dpmn_amd CMM: gradients in eval mode (running-stat BatchNorm) are not built; use .train()")
        P = self._packed()
        c = self.cnum
        Bn = x1.shape[0]
        if NATIVE_FORWARD and x1.shape[2] % 32 == 0 and x1.shape[3] % 32 == 0 and c % 8 == 0:
            return ops.cmm_forward(*self._native(P, x1.shape[2], x1.shape[3]), x1, x2, self.c_img, self._native_ws)
        xin = torch.cat([ops.nchw_to_nhwc(x1.contiguous().float(), 4), ops.nchw_to_nhwc(x2.contiguous().float(), 4)], 0)
        o = [ops.conv2d([xin], *P["en_1"], c, 3, pad=1, groups=2)]
        chans = {2: (c, 2 * c), 3: (2 * c, 4 * c), 4: (4 * c, 8 * c), 5: (8 * c, 8 * c)}
        for lvl in (2, 3, 4, 5):
            ci, co = chans[lvl]
            # the block's inner activation (cmm.py:48 / 69) runs in the producer's epilogue, once per element, instead of on load
            # in every (tap, output-tile) pass of the consumer; the block's outputs stay raw (skip connections use them twice)
            t = ops.conv2d([o[-1]], *P["en_%d.a" % lvl], ci, 4, stride=2, pad=3, dil=2, pro_act="leaky02", epi_act="leaky02", groups=2)
            o.append(ops.conv2d([t], *P["en_%d.b" % lvl], co, 3, pad=1, groups=2))
        o.append(ops.conv2d([o[-1]], *P["en_6"], 8 * c, 4, stride=2, pad=1, pro_act="leaky02", groups=2))
        a, b = [t[:Bn] for t in o], [t[Bn:] for t in o]          # batch halves: contiguous views
        bott = torch.cat([a[5], b[5]], dim=3)  # (B,1,4,16c) tiny concat feeding the gate (device memory plumbing)
        gated = ops.se_gate(bott, self.fc_1.weight, self.fc_1.bias, self.fc_2.weight, self.fc_2.bias)
        d = ops.convT_s2k4([gated], P["de_6"], 8 * c, pro_act="relu")
        outc = {5: 8 * c, 4: 4 * c, 3: 2 * c, 2: c}
        for lvl, skip in ((5, 4), (4, 3), (3, 2), (2, 1)):
            t = ops.conv2d([d, a[skip], b[skip]], *P["de_%d.a" % lvl], outc[lvl], 3, pad=1, pro_act="relu", epi_act="relu")
            d = ops.convT_s2k4([t], P["de_%d.b" % lvl], outc[lvl])
        return ops.conv2d([d, a[0], b[0]], *P["de_1"], self.c_img, 3, pad=1, pro_act="relu", out_nchw=True)

"""ORACLE (test infrastructure only) -- CPU restatement of the reference's VisionLAN recogniser in eval mode, the text-prior
generator of branch 1, plus the string decode and a glyph-atlas text-prior composer.

Only tests/ may import this file.  Pinned by tests/golden/visionlan.npz (tools/gen_golden.py gen_visionlan: the imported
model/VisionLAN/VisionLAN.py run on name-seeded synthetic weights) for everything up to the decoded strings.  The glyph
composer (`compose_text_prior`) is OUR replacement for utils/render_standard_text.py (pygame FreeType rasteriser + cv2.resize,
neither available here): it is specified by this file, parity with the reference renderer is UNPINNED and stated so.

Reference lines restated:
  model/VisionLAN/modules/resnet.py: BasicBlock.forward 24-37, ResNet.forward 80-112 (resnet45: layers [3,4,6,6,3],
      strides [(1,1),(2,2),(2,2),(2,2),(1,1),(1,1)], compress_layer=False; BatchNorm in eval mode)
  model/VisionLAN/modules/modules.py: PositionalEncoding 6-20, MultiHeadAttention.forward 60-81 (8 heads x 64, post-LN),
      PositionwiseFeedForward.forward 93-100, Transforme_Encoder.forward 126-131 (3 layers, final LayerNorm eps 1e-6),
      PP_layer.forward 163-172, Prediction.forward 199-202 (eval branch)
  model/VisionLAN/VisionLAN.py: MLM_VRM.forward 70-75 (token order w * 8 + h), 107-135 (length = first EOS step + 1, else 25),
      VisionLAN.forward 156-166
  model/VisionLAN/utils.py: cha_encdec.decode 30-38 (class c > 0 -> dict[c - 1], dic_36.txt = a-z, 1-9, 0)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = [3, 4, 6, 6, 3]
PLANES = [32, 64, 128, 256, 512]
STRIDES = [(1, 1), (2, 2), (2, 2), (2, 2), (1, 1), (1, 1)]
DICT36 = "abcdefghijklmnopqrstuvwxyz1234567890"      # dic_36.txt: a-z, then 1-9, then 0


def _bn(x, sd, pre):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"], False, 0.0, 1e-5)


def backbone(sd, x, pre="backbone."):
    x = F.relu(_bn(F.conv2d(x, sd[pre + "conv1_new.weight"], None, STRIDES[0], 1), sd, pre + "bn1."))
    inplanes = 32
    for li, (n, planes) in enumerate(zip(LAYERS, PLANES)):
        for bi in range(n):
            p = "%slayer%d.%d." % (pre, li + 1, bi)
            stride = STRIDES[li + 1] if bi == 0 else (1, 1)
            out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
            out = _bn(F.conv2d(out, sd[p + "conv2.weight"], None, stride, 1), sd, p + "bn2.")
            res = x
            if bi == 0 and (stride != (1, 1) or inplanes != planes):
                res = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, stride), sd, p + "downsample.1.")
            x = F.relu(out + res)
        inplanes = planes
    return x


def pos_table(n_position=256, d=512):
    pos = np.arange(n_position)[:, None].astype(np.float64)
    j = np.arange(d)[None, :]
    ang = pos / np.power(10000, 2 * (j // 2) / d)
    tab = np.zeros((n_position, d))
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(tab).float()


def encoder(sd, x, pre, n_layers=3, n_head=8, d_k=64):
    B, L, D = x.shape
    # the table is a registered buffer (a checkpoint carries it; pos_table() is what the constructor fills it with)
    tab = sd[pre + "position_enc.pos_table"][0] if pre + "position_enc.pos_table" in sd else pos_table(256, D)
    x = x + tab[:L]
    for i in range(n_layers):
        p = "%slayer_stack.%d." % (pre, i)
        a = p + "slf_attn."
        q = F.linear(x, sd[a + "w_qs.weight"], sd[a + "w_qs.bias"]).view(B, L, n_head, d_k).transpose(1, 2)
        k = F.linear(x, sd[a + "w_ks.weight"], sd[a + "w_ks.bias"]).view(B, L, n_head, d_k).transpose(1, 2)
        v = F.linear(x, sd[a + "w_vs.weight"], sd[a + "w_vs.bias"]).view(B, L, n_head, d_k).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d_k), -1)
        o = (att @ v).transpose(1, 2).reshape(B, L, n_head * d_k)
        x = F.layer_norm(F.linear(o, sd[a + "fc.weight"], sd[a + "fc.bias"]) + x, (D,), sd[a + "layer_norm.weight"], sd[a + "layer_norm.bias"], 1e-5)
        f = p + "pos_ffn."
        h = F.relu(F.linear(x, sd[f + "w_1.weight"].squeeze(-1), sd[f + "w_1.bias"]))
        h = F.linear(h, sd[f + "w_2.weight"].squeeze(-1), sd[f + "w_2.bias"])
        x = F.layer_norm(h + x, (D,), sd[f + "layer_norm.weight"], sd[f + "layer_norm.bias"], 1e-5)
    return F.layer_norm(x, (D,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-6)


def pp_layer(sd, enc, pre):
    t = F.linear(sd[pre + "f0_embedding.weight"].t(), sd[pre + "w0.weight"], sd[pre + "w0.bias"])       # (512, 256)
    t = torch.tanh(t.t()[None] + F.linear(enc, sd[pre + "wv.weight"], sd[pre + "wv.bias"]))              # (B, 256, 512)
    t = F.linear(t, sd[pre + "we.weight"], sd[pre + "we.bias"])                                          # (B, 256, 26)
    t = torch.softmax(t.transpose(1, 2), 2)                                                              # (B, 26, 256)
    return t @ enc, t


def logits(sd, images):
    """images (B, 3, 64, 256) in [0, 1] -> per-step class logits (B, 26, 37); only the first 25 steps are ever read."""
    feat = backbone(sd, images)                                   # (B, 512, 8, 32)
    B, C, H, W = feat.shape
    tokens = feat.permute(0, 1, 3, 2).reshape(B, C, W * H).permute(0, 2, 1)      # token = w * 8 + h   (VisionLAN.py:71-74)
    enc = encoder(sd, tokens, "MLM_VRM.SequenceModeling.")
    g, _ = pp_layer(sd, enc, "MLM_VRM.Prediction.pp.")
    return F.linear(g, sd["MLM_VRM.Prediction.w_vrm.weight"], sd["MLM_VRM.Prediction.w_vrm.bias"])


def decode(lg, n_steps=25):
    """argmax classes (B, 25), lengths (first EOS step + 1, else 25; VisionLAN.py:114-126) and the strings of cha_encdec.decode."""
    cls = lg[:, :n_steps].argmax(-1)
    B = cls.shape[0]
    lengths = torch.full((B,), n_steps, dtype=torch.long)
    for b in range(B):
        z = (cls[b] == 0).nonzero()
        if len(z):
            lengths[b] = int(z[0]) + 1
    texts = ["".join(DICT36[c - 1] if 0 < c <= 36 else "" for c in cls[b, :lengths[b]].tolist()) for b in range(B)]
    return cls, lengths, texts


def resize_for_visionlan(img3, out_h=64, out_w=256):
    """parse_visionlan_data (base.py:473-478): uint8 quantisation (ToPILImage: mul(255).byte()), cv2.resize INTER_LINEAR to
    256x64 (half-pixel centres, edge clamp), ToTensor (/255).  cv2's fixed-point interpolation (11-bit coefficients) is
    restated in floating point: parity with cv2 itself is UNPINNED (cv2 absent), differences are <= 1 grey level."""
    B, C, H, W = img3.shape
    q = torch.floor(img3.clamp(0, 1) * 255.0 + 0.0)               # in-range inputs only (SURVEY quirk Q13 for the rest)
    ys = (torch.arange(out_h, dtype=torch.float32) + 0.5) * (H / out_h) - 0.5
    xs = (torch.arange(out_w, dtype=torch.float32) + 0.5) * (W / out_w) - 0.5
    y0, x0 = torch.floor(ys), torch.floor(xs)
    fy, fx = (ys - y0)[None, None, :, None], (xs - x0)[None, None, None, :]
    y0i, y1i = y0.long().clamp(0, H - 1), (y0.long() + 1).clamp(0, H - 1)
    x0i, x1i = x0.long().clamp(0, W - 1), (x0.long() + 1).clamp(0, W - 1)
    g = lambda yi, xi: q[:, :, yi][:, :, :, xi]
    v = (g(y0i, x0i) * (1 - fx) + g(y0i, x1i) * fx) * (1 - fy) + (g(y1i, x0i) * (1 - fx) + g(y1i, x1i) * fx) * fy
    return torch.floor(v + 0.5).clamp(0, 255) / 255.0


def compose_text_prior(cls, lengths, atlas, advance, out_h=32, out_w=128, border_frac=0.1):
    """Text prior (B, 2, out_h, out_w), uint8-valued floats (quirk Q6), from decoded classes: channel 0 = the lower-case
    string, channel 1 = the upper-case string, each laid out from a pre-rendered glyph atlas and stretched to the full
    image like make_standard_text's final cv2.resize(canvas, (W, H)) (render_standard_text.py:70-71).
      atlas   (2, 37, GH, GW) floats 0..255: [case][class] glyph bitmaps, class 0 = the blank drawn for an empty string ('\\t')
      advance (2, 37) ints: horizontal extent of each glyph inside its GW-wide cell
    Layout: the glyph cells of the string's characters (EOS and out-of-dictionary classes skipped) are concatenated at
    their advances into a virtual GH x sum(advance) canvas; output pixel (y, x) samples it bilinearly at half-pixel centres
    with edge clamp.  An empty string uses glyph 0."""
    B = cls.shape[0]
    GH, GW = atlas.shape[2], atlas.shape[3]
    out = torch.zeros(B, 2, out_h, out_w)
    for b in range(B):
        chars = [int(c) for c in cls[b, :int(lengths[b])].tolist() if 0 < int(c) <= 36] or [0]
        for case in range(2):
            canvas = torch.cat([atlas[case, c, :, :int(advance[case, c])] for c in chars], 1)      # (GH, Wc)
            Wc = canvas.shape[1]
            ys = (torch.arange(out_h, dtype=torch.float32) + 0.5) * (GH / out_h) - 0.5
            xs = (torch.arange(out_w, dtype=torch.float32) + 0.5) * (Wc / out_w) - 0.5
            y0, x0 = torch.floor(ys), torch.floor(xs)
            fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
            y0i, y1i = y0.long().clamp(0, GH - 1), (y0.long() + 1).clamp(0, GH - 1)
            x0i, x1i = x0.long().clamp(0, Wc - 1), (x0.long() + 1).clamp(0, Wc - 1)
            v = (canvas[y0i][:, x0i] * (1 - fx) + canvas[y0i][:, x1i] * fx) * (1 - fy) + \
                (canvas[y1i][:, x0i] * (1 - fx) + canvas[y1i][:, x1i] * fx) * fy
            out[b, case] = torch.floor(v + 0.5).clamp(0, 255)
    return out

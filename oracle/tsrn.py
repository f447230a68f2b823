"""ORACLE (test infrastructure only) -- CPU restatement of the PSN backbones TSRN and TATT
(TSRN_TL_TRANS) in eval mode (the only mode DPMN uses them in: super_resolution.py:56-59).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
Pinned by tests/golden/{tsrn,tatt,tbsrn}.npz (tools/gen_golden.py, imported reference).

Reference lines restated:
  model/tsrn.py: TSRN.forward 58-74, RecurrentResidualBlock.forward 89-101, UpsampleBLock 113-117,
                 mish 125-129, GruBlock.forward 139-150
  model/tbsrn.py: TBSRN.forward 214-226, RecurrentResidualBlock.forward 245-257 (gru1/gru2 are constructed
                 but never called), FeatureEnhancer.forward 76-92, positionalencoding2d 39-60, MultiHeadedAttention
                 110-129 + attention 132-150, PositionwiseFeedForward 161-162, LayerNorm 33-36 (unbiased std, eps
                 added to the std)
  model/tatt.py: TSRN_TL_TRANS.forward 645-691, TPInterpreter.forward 193-223,
                 RecurrentResidualBlockTL.forward 891-909, GruBlock 1070-1083
  model/transformer_v2.py: PositionalEncoding 22-43, InfoTransformer.forward 198-244 (quirk Q5: the
                 query-embed GRU is batch_first but is fed (W, B, H*C), so it recurs over the batch),
                 TransformerEncoder.forward 256-281 (layer input is output + src = 2*src),
                 TransformerEncoderLayer.forward_post 455-469, TransformerDecoder.forward 353-390,
                 TransformerDecoderLayer_TP.forward_post 806-833
"""
import math

import torch
import torch.nn.functional as F


def mish(x):
    return x * torch.tanh(F.softplus(x))


def bn_eval(x, sd, pre, eps=1e-5):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                        False, 0.0, eps)


def gru_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x (N, T, I) -> (N, T, H).  PyTorch gate order [r, z, n]."""
    N, T, _ = x.shape
    H = w_hh.shape[1]
    gi = F.linear(x, w_ih, b_ih)
    h = x.new_zeros(N, H)
    out = x.new_zeros(N, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        out[:, t] = h
    return out


def bigru(x, sd, pre):
    f = gru_dir(x, sd[pre + "weight_ih_l0"], sd[pre + "weight_hh_l0"], sd[pre + "bias_ih_l0"], sd[pre + "bias_hh_l0"], False)
    b = gru_dir(x, sd[pre + "weight_ih_l0_reverse"], sd[pre + "weight_hh_l0_reverse"], sd[pre + "bias_ih_l0_reverse"],
                sd[pre + "bias_hh_l0_reverse"], True)
    return torch.cat([f, b], -1)


def gru_block(x, sd, pre):
    """conv1x1 then a BiGRU along the LAST spatial axis, rows as batch (tsrn.py:139-150)."""
    y = F.conv2d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"])
    B, C, H, W = y.shape
    s = y.permute(0, 2, 3, 1).reshape(B * H, W, C)
    s = bigru(s, sd, pre + "gru.")
    return s.reshape(B, H, W, C).permute(0, 3, 1, 2)


def srb(x, sd, pre, tp=None):
    r = bn_eval(F.conv2d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], padding=1), sd, pre + "bn1.")
    r = mish(r)
    r = bn_eval(F.conv2d(r, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], padding=1), sd, pre + "bn2.")
    if tp is not None:
        r = torch.cat([r, tp], 1)
    r = gru_block(r.transpose(-1, -2), sd, pre + "gru1.").transpose(-1, -2)
    return gru_block(x + r, sd, pre + "gru2.")


def _tail(b1, feat, sd, n):
    x = b1 + bn_eval(F.conv2d(feat, sd["block%d.0.weight" % (n + 2)], sd["block%d.0.bias" % (n + 2)], padding=1),
                     sd, "block%d.1." % (n + 2))
    pre = "block%d." % (n + 3)
    x = mish(F.pixel_shuffle(F.conv2d(x, sd[pre + "0.conv.weight"], sd[pre + "0.conv.bias"], padding=1), 2))
    x = F.conv2d(x, sd[pre + "1.weight"], sd[pre + "1.bias"], padding=4)
    return torch.tanh(x)


def tsrn_forward(sd, x, srb_nums=5):
    b1 = F.prelu(F.conv2d(x, sd["block1.0.weight"], sd["block1.0.bias"], padding=4), sd["block1.1.weight"])
    f = b1
    for i in range(srb_nums):
        f = srb(f, sd, "block%d." % (i + 2))
    return _tail(b1, f, sd, srb_nums)


# ---------------------------------------------------------------------------------- TBSRN
def tbsrn_pos2d(d_model, height, width):
    """positionalencoding2d (tbsrn.py:39-60): first half of the channels encodes the column, second half the row."""
    pe = torch.zeros(d_model, height, width)
    half = d_model // 2
    div = torch.exp(torch.arange(0., half, 2) * -(math.log(10000.0) / half))
    pw = torch.arange(0., width)[:, None] * div       # (W, half/2)
    ph = torch.arange(0., height)[:, None] * div
    pe[0:half:2] = torch.sin(pw).t()[:, None, :].expand(-1, height, -1)
    pe[1:half:2] = torch.cos(pw).t()[:, None, :].expand(-1, height, -1)
    pe[half::2] = torch.sin(ph).t()[:, :, None].expand(-1, -1, width)
    pe[half + 1::2] = torch.cos(ph).t()[:, :, None].expand(-1, -1, width)
    return pe


def tbsrn_ln(x, sd, pre, eps=1e-6):
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)                      # unbiased, like torch.Tensor.std
    return sd[pre + "a_2"] * (x - mean) / (std + eps) + sd[pre + "b_2"]


def feature_enhancer(feat, sd, pre):
    """feat (B, 64, 1024) -> (B, 64, 1024) (tbsrn.py:76-92); dropout layers are identity in eval."""
    B = feat.shape[0]
    pos = tbsrn_pos2d(64, 16, 64).reshape(1, 64, 1024).expand(B, -1, -1)
    x = torch.cat([feat, pos], 1).permute(0, 2, 1)      # (B, 1024, 128)
    h, dk = 4, 32
    lin = lambda i, v: F.linear(v, sd[pre + "multihead.linears.%d.weight" % i], sd[pre + "multihead.linears.%d.bias" % i])
    q, k, v = (lin(i, x).reshape(B, -1, h, dk).transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(dk), -1)
    att = (p @ v).transpose(1, 2).reshape(B, -1, h * dk)
    x = tbsrn_ln(x + lin(3, att), sd, pre + "mul_layernorm1.")
    ff = F.linear(F.relu(F.linear(x, sd[pre + "pff.w_1.weight"], sd[pre + "pff.w_1.bias"])), sd[pre + "pff.w_2.weight"],
                  sd[pre + "pff.w_2.bias"])
    x = tbsrn_ln(x + ff, sd, pre + "mul_layernorm3.")
    return F.linear(x, sd[pre + "linear.weight"], sd[pre + "linear.bias"]).permute(0, 2, 1)


def tbsrn_block(x, sd, pre):
    r = mish(bn_eval(F.conv2d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], padding=1), sd, pre + "bn1."))
    r = bn_eval(F.conv2d(r, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], padding=1), sd, pre + "bn2.")
    B, C, H, W = r.shape
    return x + feature_enhancer(r.reshape(B, C, H * W), sd, pre + "feature_enhancer.").reshape(B, C, H, W)


def tbsrn_forward(sd, x, srb_nums=5):
    """TBSRN.forward in eval mode (the STN branch is gated on self.training, tbsrn.py:215)."""
    b1 = F.prelu(F.conv2d(x, sd["block1.0.weight"], sd["block1.0.bias"], padding=4), sd["block1.1.weight"])
    f = b1
    for i in range(srb_nums):
        f = tbsrn_block(f, sd, "block%d." % (i + 2))
    return _tail(b1, f, sd, srb_nums)


# ---------------------------------------------------------------------------------- TATT
def positional_encoding(L, d):
    pe = torch.zeros(L, d)
    pos = torch.arange(0, L).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def mha(q_in, k_in, v_in, sd, pre, nhead=4):
    """nn.MultiheadAttention forward, (L, N, E) inputs; returns (out (L,N,E), head-averaged weights (N,L,S))."""
    E = q_in.shape[-1]
    W, bI = sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"]
    q = F.linear(q_in, W[:E], bI[:E])
    k = F.linear(k_in, W[E:2 * E], bI[E:2 * E])
    v = F.linear(v_in, W[2 * E:], bI[2 * E:])
    L, N, _ = q.shape
    S = k.shape[0]
    d = E // nhead
    q = q.reshape(L, N, nhead, d).permute(1, 2, 0, 3)
    k = k.reshape(S, N, nhead, d).permute(1, 2, 0, 3)
    v = v.reshape(S, N, nhead, d).permute(1, 2, 0, 3)
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1)  # (N, h, L, S)
    o = (p @ v).permute(2, 0, 1, 3).reshape(L, N, E)
    return F.linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"]), p.mean(1)


def _ln(x, sd, pre):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + "weight"], sd[pre + "bias"])


def tp_interpreter(sd, feat, text_emb, pre="infoGen."):
    """TPInterpreter.forward (tatt.py:193-223) -> (tp_map (N,64,H,W), pr_weights (N, HW, 26))."""
    N, C, H, W = feat.shape
    x = text_emb.permute(0, 3, 1, 2).squeeze(-1)                       # (N, 26, 37)
    x = F.prelu(F.linear(x, sd[pre + "fc_in.weight"], sd[pre + "fc_in.bias"]), sd[pre + "activation.weight"])
    Ls = x.shape[1]
    pos = positional_encoding(Ls, x.shape[-1])[:, None, :].expand(Ls, N, -1)  # (26, N, 64)
    src = x.permute(1, 0, 2)                                           # (26, N, 64)
    tgt = feat.reshape(N, C, H * W).permute(2, 0, 1)                    # (HW, N, 64)
    t = pre + "upsample_transformer."
    qe = sd[pre + "init_factor.weight"][:, None, :].expand(-1, N, -1)   # (HW, N, 64)
    hc = qe.shape[-1]
    qe = qe.reshape(H, W, N, hc).permute(1, 2, 0, 3).reshape(W, N, H * hc)
    qe = bigru(qe, sd, t + "gru_encoding.")                             # batch_first: recurs over N (quirk Q5)
    qe = qe.reshape(W, N, H, hc).permute(2, 0, 1, 3).reshape(H * W, N, hc)
    # encoder (1 layer): layer input is output + src = 2*src (transformer_v2.py:272-277)
    e = t + "encoder.layers.0."
    s2 = src + src
    a, _ = mha(s2 + pos, s2 + pos, s2, sd, e + "self_attn.")
    s2 = _ln(s2 + a, sd, e + "norm1.")
    ff = F.linear(F.relu(F.linear(s2, sd[e + "linear1.weight"], sd[e + "linear1.bias"])), sd[e + "linear2.weight"],
                  sd[e + "linear2.bias"])
    mem = _ln(s2 + ff, sd, e + "norm2.")
    out = tgt
    inter = []
    w = None
    for li in range(2):
        d = t + "decoder.layers.%d." % li
        a, w = mha(out + qe, mem + pos, mem, sd, d + "multihead_attn.")
        out = _ln(out + a, sd, d + "norm2.")
        ff = F.linear(F.relu(F.linear(out, sd[d + "linear1.weight"], sd[d + "linear1.bias"])), sd[d + "linear2.weight"],
                      sd[d + "linear2.bias"])
        out = _ln(out + ff, sd, d + "norm3.")
        inter.append(_ln(out, sd, t + "decoder.norm."))
    tp = torch.stack(inter).mean(0)                                     # (HW, N, 64)
    return tp.permute(1, 2, 0).reshape(N, hc, H, W), w


def tatt_forward(sd, x, text_emb, srb_nums=5):
    """TSRN_TL_TRANS.forward in eval mode -> (output (N,4,2H,2W), pr_weights)."""
    b1 = F.prelu(F.conv2d(x, sd["block1.0.weight"], sd["block1.0.bias"], padding=4), sd["block1.1.weight"])
    tp_map, prw = tp_interpreter(sd, b1, text_emb)
    f = b1
    for i in range(srb_nums):
        f = srb(f, sd, "block%d." % (i + 2), tp_map)
    return _tail(b1, f, sd, srb_nums), prw


# ---------------------------------------------------------------------------------- TPGSR (--arch tpgsr)
def info_gen(sd, text_emb, pre="infoGen."):
    """InfoGen.forward (tsrn.py:280-304): four ConvTranspose2d (k3, stride 2 / 2 / 2 / (2,1), padding 1 / 1 / 1 / (1,0), no bias) each
    followed by eval BatchNorm and ReLU: (N, 37, 1, 26) -> (N, 32, 1, 203)."""
    x = text_emb
    for i, (st, pad) in enumerate(((2, 1), (2, 1), (2, 1), ((2, 1), (1, 0)))):
        x = F.conv_transpose2d(x, sd[pre + "tconv%d.weight" % (i + 1)], None, stride=st, padding=pad)
        x = F.relu(bn_eval(x, sd, pre + "bn%d." % (i + 1)))
    return x


def tsrn_tl_forward(sd, x, text_emb=None, srb_nums=5):
    """TSRN_TL.forward in eval mode (tsrn.py:214-247): the text embedding goes through InfoGen, is stretched bilinearly
    (align_corners=True) to the LR feature map and concatenated into every RecurrentResidualBlockTL (tsrn.py:262-277, the same
    block as TATT's)."""
    b1 = F.prelu(F.conv2d(x, sd["block1.0.weight"], sd["block1.0.bias"], padding=4), sd["block1.1.weight"])
    if text_emb is None:
        text_emb = torch.zeros(x.shape[0], 37, 1, 26)
    tp = F.interpolate(info_gen(sd, text_emb), (x.shape[2], x.shape[3]), mode="bilinear", align_corners=True)
    f = b1
    for i in range(srb_nums):
        f = srb(f, sd, "block%d." % (i + 2), tp)
    return _tail(b1, f, sd, srb_nums)

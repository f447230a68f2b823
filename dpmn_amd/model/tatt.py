"""Host-side mirror of ``model/tatt.py::TSRN_TL_TRANS`` (the TATT PSN), eval forward.

forward(x, text_emb) -> (sr (N,4,2H,2W), pr_weights (N, H*W, 26)) exactly like tatt.py:645-691 in eval.
The query-embedding BiGRU of InfoTransformer (transformer_v2.py:177,215-218) depends only on the
weights and on the batch size (quirk Q5: it recurs over the batch axis), so -- like the eval-mode
BatchNorm fold -- it is evaluated when the weights are packed and reused until they or B change
(set ``cache_query_embed=False`` to recompute it on every call like the reference does).
"""
import math

import torch
from .._abi import stream_of as _abi_stream_of
import torch.nn as nn

from .. import ops
from .tsrn import _PSNBase


class _EncLayer(nn.Module):
    def __init__(self, d, nhead, ff):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, nhead)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class _DecLayer(nn.Module):
    def __init__(self, d, nhead, ff):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, nhead)       # present in the checkpoint, unused (transformer_v2.py:817-819)
        self.multihead_attn = nn.MultiheadAttention(d, nhead)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)
        self.norm3 = nn.LayerNorm(d)


class _Stack(nn.Module):
    def __init__(self, layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        if norm is not None:
            self.norm = norm


class _InfoTransformer(nn.Module):
    def __init__(self, d, nhead, ff, n_enc, n_dec, feat_h):
        super().__init__()
        self.encoder = _Stack([_EncLayer(d, nhead, ff) for _ in range(n_enc)])
        self.decoder = _Stack([_DecLayer(d, nhead, ff) for _ in range(n_dec)], nn.LayerNorm(d))
        self.gru_encoding = nn.GRU(d * feat_h, d * feat_h // 2, bidirectional=True, batch_first=True)


class _PE(nn.Module):
    def __init__(self, d, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d)
        pos = torch.arange(0, max_len).unsqueeze(1).float()
        div = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe.unsqueeze(0))


class _TPInterpreter(nn.Module):
    def __init__(self, t_emb, d, output_size):
        super().__init__()
        self.fc_in = nn.Linear(t_emb, d)
        self.fc_feature_in = nn.Linear(64, d)                   # unused by forward, kept for the key layout
        self.activation = nn.PReLU()
        self.upsample_transformer = _InfoTransformer(d, 4, d, 1, 2, output_size[0])
        self.pe = _PE(d)
        self.init_factor = nn.Embedding(output_size[0] * output_size[1], d)


class TSRN_TL_TRANS(_PSNBase):
    """Drop-in for ``model.tatt.TSRN_TL_TRANS`` (tatt.py:575-691)."""

    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32,
                 word_vec_d=300, text_emb=37, out_text_channels=64, feature_rotate=False, rotate_train=3.,
                 cache_query_embed=True, need_pr_weights=True):
        super().__init__()
        if out_text_channels != 64 or hidden_units != 32:
            raise NotImplementedError("dpmn_amd TATT: built for out_text_channels=64, hidden_units=32 (base.py:144-148)")
        self._build_trunk(scale_factor, width, height, STN, srb_nums, mask, hidden_units, out_text_channels)
        self.infoGen = _TPInterpreter(text_emb, out_text_channels, (height // scale_factor, width // scale_factor))
        self._build_tail(srb_nums)
        self.feat_hw = (height // scale_factor, width // scale_factor)
        self.cache_query_embed = cache_query_embed
        self.need_pr_weights = need_pr_weights
        self._qe = None

    def _extra_pack(self, P):
        ig = self.infoGen
        P["fc_in_slope"] = float(ig.activation.weight.reshape(-1)[0].item())
        e = ig.upsample_transformer.encoder.layers[0]
        P["enc"] = [e.self_attn.in_proj_weight, e.self_attn.in_proj_bias, e.self_attn.out_proj.weight,
                    e.self_attn.out_proj.bias, e.linear1.weight, e.linear1.bias, e.linear2.weight, e.linear2.bias,
                    e.norm1.weight, e.norm1.bias, e.norm2.weight, e.norm2.bias]
        P["dec"] = []
        E = 64
        for d in ig.upsample_transformer.decoder.layers:
            W, bI = d.multihead_attn.in_proj_weight, d.multihead_attn.in_proj_bias
            P["dec"].append(dict(wq=W[:E].contiguous(), bq=bI[:E].contiguous(), wk=W[E:2 * E].contiguous(),
                                 bk=bI[E:2 * E].contiguous(), wv=W[2 * E:].contiguous(), bv=bI[2 * E:].contiguous()))
        self._qe = None

    # ------------------------------------------------------------------ query embedding (quirk Q5)
    def _query_embed(self, B, dev):
        if self.cache_query_embed and self._qe is not None and self._qe[0] == B:
            return self._qe[1]
        H, W = self.feat_hw
        g = self.infoGen.upsample_transformer.gru_encoding
        init = self.infoGen.init_factor.weight
        hc = init.shape[1]
        X = init.reshape(H, W, hc).permute(1, 0, 2).reshape(W, H * hc).contiguous()   # layout plumbing
        Hh = g.hidden_size
        hist = torch.empty(W, B, 2 * Hh, device=dev)
        for d, (wih, whh, bih, bhh) in enumerate(((g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0),
                                                  (g.weight_ih_l0_reverse, g.weight_hh_l0_reverse, g.bias_ih_l0_reverse,
                                                   g.bias_hh_l0_reverse))):
            gi = ops.linear(X, wih, bih)                       # the input is the same at every step
            h = torch.zeros(W, Hh, device=dev)
            steps = range(B) if d == 0 else range(B - 1, -1, -1)
            for t in steps:
                gh = ops.linear(h, whh, bhh)
                ops.gru_gate(gi, gh, h, hist[:, t, d * Hh:], B * 2 * Hh)
        qe = hist.reshape(W, B, H, hc).permute(1, 2, 0, 3).reshape(B, H * W, hc).contiguous()   # (N, L=h*W+w, E)
        self._qe = (B, qe)
        return qe

    def _interp_native(self, P):
        """dpmn_tatt_interp_weights over the module's own parameter tensors (include/dpmn_hip.h), rebuilt with the pack."""
        from .. import _abi
        nat = getattr(self, "_inat", None)
        if nat is not None and nat[0] is P:
            return nat[1]
        ig = self.infoGen
        layers = ig.upsample_transformer.decoder.layers
        w = _abi.TattInterpWeights()
        w.n_dec, w.nhead = len(layers), 4
        w.fc_in_w, w.fc_in_b, w.fc_in_slope = ig.fc_in.weight.data_ptr(), ig.fc_in.bias.data_ptr(), P["fc_in_slope"]
        for i, t in enumerate(P["enc"]):
            w.enc[i] = t.data_ptr()
        for i, (d, pk) in enumerate(zip(layers, P["dec"])):
            q = w.dec[i]
            for n in ("wq", "bq", "wk", "bk", "wv", "bv"):
                setattr(q, n, pk[n].data_ptr())
            q.out_w, q.out_b = d.multihead_attn.out_proj.weight.data_ptr(), d.multihead_attn.out_proj.bias.data_ptr()
            q.norm2_w, q.norm2_b, q.norm3_w, q.norm3_b = (t.data_ptr() for t in (d.norm2.weight, d.norm2.bias, d.norm3.weight, d.norm3.bias))
            q.lin1_w, q.lin1_b, q.lin2_w, q.lin2_b = (t.data_ptr() for t in (d.linear1.weight, d.linear1.bias, d.linear2.weight, d.linear2.bias))
        dn = ig.upsample_transformer.decoder.norm
        w.dec_norm_w, w.dec_norm_b = dn.weight.data_ptr(), dn.bias.data_ptr()
        self._inat = (P, w)
        return w

    def _tp_interpreter(self, b1, text_emb, P):
        ig = self.infoGen
        B, H, W, E = b1.shape
        L = H * W
        x = text_emb.float().squeeze(2).transpose(1, 2).contiguous()          # (N, 26, 37) layout plumbing
        S = x.shape[1]
        from . import tsrn as _tsrn
        if _tsrn.NATIVE_TRUNK and len(ig.upsample_transformer.decoder.layers) <= 4:
            # the whole interpreter from one native call (csrc/psn_forward.hip dpmn_tatt_interpreter_f32)
            import ctypes as _C
            from .._abi import lib, check, dptr, stream
            w = self._interp_native(P)
            pos = ig.pe.pe[0, :S].contiguous()
            qe = self._query_embed(B, b1.device)
            tp = torch.empty(B * L, E, device=b1.device)
            pw = torch.empty(B, L, S, device=b1.device) if self.need_pr_weights else None
            if not hasattr(self, "_interp_ws"):
                self._interp_ws = {}
            key = (B, L, S, _abi_stream_of(b1.device))
            if key not in self._interp_ws:
                self._interp_ws[key] = torch.empty(lib.dpmn_tatt_interpreter_workspace_bytes(B, L, S) // 4, device=b1.device)
            ws = self._interp_ws[key]
            check(lib.dpmn_tatt_interpreter_f32(_C.byref(w), dptr(x), x.shape[2], dptr(b1), dptr(qe), dptr(pos), dptr(tp), dptr(pw, True),
                                                dptr(ws), ws.numel() * 4, B, L, S, stream()))
            return tp.reshape(B, H, W, E), pw
        src = ops.small_linear(x.reshape(B * S, -1), ig.fc_in.weight, ig.fc_in.bias, act="prelu", slope=P["fc_in_slope"])
        pos = ig.pe.pe[0, :S].contiguous()                                     # (26, 64)
        mem = ops.tatt_encoder_layer(src.reshape(B, S, E), pos, P["enc"]).reshape(B * S, E)
        qe = self._query_embed(B, b1.device).reshape(B * L, E)
        out = b1.reshape(B * L, E)
        tp = torch.empty(B * L, E, device=b1.device)
        dn = ig.upsample_transformer.decoder.norm
        pw = None
        layers = ig.upsample_transformer.decoder.layers
        for li, (d, pk) in enumerate(zip(layers, P["dec"])):
            q = ops.add_linear(out, qe, pk["wq"], pk["bq"])
            k = ops.small_linear(mem, pk["wk"], pk["bk"], add=pos)
            v = ops.small_linear(mem, pk["wv"], pk["bv"])
            last = li == len(layers) - 1
            o, w_ = ops.cross_attn(q.reshape(B, L, E), k.reshape(B, S, E), v.reshape(B, S, E),
                                   need_weights=last and self.need_pr_weights)
            pw = w_ if last else pw
            t1 = ops.linear(o.reshape(B * L, E), d.multihead_attn.out_proj.weight, d.multihead_attn.out_proj.bias, res1=out)
            out = ops.add_layernorm64(t1, None, d.norm2.weight, d.norm2.bias)
            f = ops.linear(out, d.linear1.weight, d.linear1.bias, act="relu")
            t2 = ops.linear(f, d.linear2.weight, d.linear2.bias, res1=out)
            out = ops.add_layernorm64(t2, None, d.norm3.weight, d.norm3.bias, dn.weight, dn.bias, tp,
                                      alpha=1.0 / len(layers), accumulate=li > 0)
        return tp.reshape(B, H, W, E), pw

    def forward(self, x, text_emb=None, text_emb_gt=None, feature_arcs=None, rand_offs=None):
        self._check_mode()
        P = self._trunk_pack()
        if text_emb is None:
            text_emb = torch.zeros(1, 37, 1, 26, device=x.device)
        if text_emb.shape[0] != x.shape[0]:
            raise ValueError("TATT: text_emb batch must match the image batch")
        b1 = self._head(x, P)
        tp_map, prw = self._tp_interpreter(b1, text_emb, P)
        return self._trunk(b1, P, tp_map), prw

"""Thin Python wrappers over the per-kernel C-ABI entry points (include/dpmn_hip.h).

Used by the parity tests and by host code that composes kernels outside the native module
drivers.  Every wrapper allocates its output with torch (device memory plumbing only) and
enqueues on the current HIP stream.
"""
import torch
from ._abi import stream_of as _abi_stream_of

from . import _abi
from ._abi import dptr, lib, check, stream

ACT = {"none": 0, "gelu": 1, "relu": 2, "leaky02": 3, "leaky001": 4, "mish": 5, "prelu": 6, "tanh": 7, "sigmoid": 8,
       "relu_post_res": 9}      # conv epilogue only: ReLU after the residual add


def linear(x, w, bias=None, res1=None, res2=None, act="none", slope=0.0):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device)
    check(lib.dpmn_linear_f32(dptr(x), dptr(w), dptr(bias, True), dptr(res1, True), dptr(res2, True), dptr(y), M, N, K,
                              ACT[act], float(slope), stream()))
    return y


def add_linear(x, addv, w, bias=None, act="none"):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device)
    check(lib.dpmn_add_linear_f32(dptr(x), dptr(addv), dptr(w), dptr(bias, True), dptr(y), M, N, K, ACT[act], stream()))
    return y


def cat2_linear(x1, x2, w, bias=None, act="none"):
    M, k1 = x1.shape
    k2 = x2.shape[1]
    N = w.shape[0]
    y = torch.empty(M, N, device=x1.device)
    check(lib.dpmn_cat2_linear_f32(dptr(x1), k1, dptr(x2), k2, dptr(w), dptr(bias, True), dptr(y), M, N, ACT[act], stream()))
    return y


def ln_linear(x, ln_w, ln_b, w, bias, act="none", eps=1e-5):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device)
    check(lib.dpmn_ln_linear_f32(dptr(x), dptr(ln_w), dptr(ln_b), eps, dptr(w), dptr(bias, True), dptr(y), M, N, K,
                                 ACT[act], stream()))
    return y


def patch_embed_ln(img, pe_w, pe_b, ln_w, ln_b, patch, pf_w=None, pf_b=None, p_drop=0.0, seed=0):
    """p_drop > 0: pos_drop (pgrm.py:550-551) in the kernel's epilogue, the mask of dropout(tokens, p_drop, seed)."""
    B, cin, Hi, Wi = img.shape
    Cd = pe_w.shape[0]
    tok = torch.empty(B, (Hi // patch) * (Wi // patch), Cd, device=img.device)
    check(lib.dpmn_patch_embed_ln_drop_f32(dptr(img), cin, dptr(pf_w, True), dptr(pf_b, True), dptr(pe_w), dptr(pe_b),
                                           dptr(ln_w), dptr(ln_b), dptr(tok), B, Hi, Wi, patch, Cd, float(p_drop), int(seed), stream()))
    return tok


def window_attn(q, kv, tables, windows, shifts, heads_per_group, H, W, p_drop=0.0, seed=0):
    """p_drop > 0: train-mode attn_drop (pgrm.py:248) with the mask regenerated from `seed` (include/dpmn_hip.h)."""
    B, L, Cd = q.shape
    out = torch.empty_like(q)
    check(lib.dpmn_window_attn_drop_f32(dptr(q), dptr(kv), _abi.ptr_array(tables), _abi.int_array(windows),
                                        _abi.int_array(shifts), len(windows), heads_per_group, dptr(out), B, H, W, Cd,
                                        float(p_drop), int(seed), stream()))
    return out


def ln_qkv_window_attn_supported(Cd, windows, heads_per_group, H, W):
    return bool(lib.dpmn_ln_qkv_window_attn_supported(Cd, len(windows), heads_per_group, _abi.int_array(windows), H, W))


def ln_qkv_window_attn(tq, tkv, lnq_w, lnq_b, lnkv_w, lnkv_b, wq, bq, wkv, bkv, tables, windows, shifts, heads_per_group, H, W,
                       eps=1e-5):
    """norm1_q / norm1_kv + q / kv Linear + multi-size window attention in ONE kernel (pgrm.py:322-323, 188-266); tq / tkv
    (B, L, C) are the token streams before the LayerNorms; returns the window-major `cat` tensor (B, L, C)."""
    B, L, Cd = tq.shape
    out = torch.empty_like(tq)
    ws = torch.empty(lib.dpmn_ln_qkv_window_attn_workspace_bytes() // 4, device=tq.device)      # folded weights of this call
    check(lib.dpmn_ln_qkv_window_attn_f32(dptr(tq), dptr(tkv), dptr(lnq_w), dptr(lnq_b), dptr(lnkv_w), dptr(lnkv_b), float(eps),
                                          dptr(wq), dptr(bq), dptr(wkv), dptr(bkv), _abi.ptr_array(tables), _abi.int_array(windows),
                                          _abi.int_array(shifts), len(windows), heads_per_group, dptr(out), dptr(ws), 1, B, H, W, Cd, stream()))
    return out


def ln_qkv_window_attn_d32_supported(Cd, windows, heads_per_group, H, W):
    return bool(lib.dpmn_ln_qkv_window_attn_d32_supported(Cd, len(windows), heads_per_group, _abi.int_array(windows), H, W))


def ln_qkv_window_attn_d32(tq, tkv, lnq_w, lnq_b, lnkv_w, lnkv_b, wq, bq, wkv, bkv, tables, windows, shifts, heads_per_group, H, W, eps=1e-5):
    """ln_qkv_window_attn at embed_dim 192 = 3 groups x 2 heads x 32, windows in {4, 8, 16} (csrc/attn_fused192.hip)."""
    B, L, Cd = tq.shape
    out = torch.empty_like(tq)
    ws = torch.empty(lib.dpmn_ln_qkv_window_attn_d32_workspace_bytes() // 4, device=tq.device)
    check(lib.dpmn_ln_qkv_window_attn_d32_f32(dptr(tq), dptr(tkv), dptr(lnq_w), dptr(lnq_b), dptr(lnkv_w), dptr(lnkv_b), float(eps),
                                              dptr(wq), dptr(bq), dptr(wkv), dptr(bkv), _abi.ptr_array(tables), _abi.int_array(windows),
                                              _abi.int_array(shifts), len(windows), heads_per_group, dptr(out), dptr(ws), 1, B, H, W, Cd, stream()))
    return out


def ln_qkv_window_attn_train(tq, tkv, lnq_w, lnq_b, lnkv_w, lnkv_b, wq, bq, wkv, bkv, tables, windows, shifts, heads_per_group, H, W,
                             p_drop=0.0, seed=0, eps=1e-5, save_qkv=True, fold_out=None):
    """Training forward of ln_qkv_window_attn: returns (cat, q, kv) -- q (B, L, C) and kv (B, L, 2C) are what the unfused backward
    reads; save_qkv=False: (cat, None, None) for the recomputing backward (ln_qkv_window_attn_bwd).  fold_out: a list that receives
    the folded-weight workspace of this call (the backward reuses it)."""
    B, L, Cd = tq.shape
    out = torch.empty_like(tq)
    q = torch.empty_like(tq) if save_qkv else None
    kv = torch.empty(B, L, 2 * Cd, device=tq.device) if save_qkv else None
    ws = torch.empty(lib.dpmn_ln_qkv_window_attn_workspace_bytes() // 4, device=tq.device)
    check(lib.dpmn_ln_qkv_window_attn_train_f32(dptr(tq), dptr(tkv), dptr(lnq_w), dptr(lnq_b), dptr(lnkv_w), dptr(lnkv_b), float(eps),
                                                dptr(wq), dptr(bq), dptr(wkv), dptr(bkv), _abi.ptr_array(tables), _abi.int_array(windows),
                                                _abi.int_array(shifts), len(windows), heads_per_group, dptr(out), dptr(q, True), dptr(kv, True),
                                                float(p_drop), int(seed), dptr(ws), B, H, W, Cd, stream()))
    if fold_out is not None:
        fold_out.append(ws)
    return out, q, kv


def ln_qkv_window_attn_bwd(tq, tkv, lnq_w, lnq_b, lnkv_w, lnkv_b, wq, bq, wkv, bkv, tables, windows, shifts, heads_per_group, H, W, dout,
                           p_drop=0.0, seed=0, eps=1e-5, fold=None):
    """Backward of ln_qkv_window_attn_train with q / k / v recomputed: returns (dq (B L, C), dkv (B L, 2C), [per-group partial rows of
    the bias-table gradients, (rows, table.numel())]).  fold: the forward's folded-weight workspace (None: fold again)."""
    B, L, Cd = tq.shape
    dq = torch.empty(B * L, Cd, device=tq.device)
    dkv = torch.empty(B * L, 2 * Cd, device=tq.device)
    rows = lib.dpmn_ln_qkv_window_attn_bwd_part_rows(B, H, W)
    parts = [torch.empty(rows, t.numel(), device=tq.device) for t in tables]
    ws = fold if fold is not None else torch.empty(lib.dpmn_ln_qkv_window_attn_workspace_bytes() // 4, device=tq.device)
    check(lib.dpmn_ln_qkv_window_attn_bwd_f32(dptr(tq), dptr(tkv), dptr(lnq_w), dptr(lnq_b), dptr(lnkv_w), dptr(lnkv_b), float(eps),
                                              dptr(wq), dptr(bq), dptr(wkv), dptr(bkv), _abi.ptr_array(tables), _abi.int_array(windows),
                                              _abi.int_array(shifts), len(windows), heads_per_group, dptr(dout), dptr(dq), dptr(dkv),
                                              _abi.ptr_array(parts), float(p_drop), int(seed), dptr(ws), 0 if fold is not None else 1,
                                              B, H, W, Cd, stream()))
    return dq, dkv, parts


def collate_u8(img_u8, with_mask):
    """(B, H, W, 3) uint8 (PIL-resized RGB, on the GPU) -> (B, 3 + mask, H, W) float: ToTensor + the mask channel of
    resizeNormalize (dataset.py:1266-1319)."""
    if not img_u8.is_cuda or img_u8.dtype != torch.uint8:
        raise _abi.DpmnError("collate_u8: a CUDA uint8 (B, H, W, 3) tensor is required")
    img_u8 = img_u8.contiguous()
    B, H, W, _ = img_u8.shape
    out = torch.empty(B, 4 if with_mask else 3, H, W, device=img_u8.device)
    check(lib.dpmn_collate_u8_f32(img_u8.data_ptr(), dptr(out), B, H, W, int(bool(with_mask)), stream()))
    return out


def maxpool(x, kh, kw, scale=None, shift=None):
    """nn.MaxPool2d((kh,kw), stride (kh,kw)) over NHWC; scale/shift: the producer's BatchNorm affine + ReLU applied on load."""
    B, H, W, Cc = x.shape
    y = torch.empty(B, H // kh, W // kw, Cc, device=x.device)
    check(lib.dpmn_maxpool_f32(dptr(x), dptr(scale, True), dptr(shift, True), dptr(y), B, H, W, Cc, kh, kw, stream()))
    return y


def stn_fc(x, in_affine, w1t, b1, bn, training, w2, b2):
    """STNHead tail (stn_head.py:94-100) -> (img_feat (B,512), ctrl (B, n_out)).  x: (B,1,W2,C) NHWC, bn: nn.BatchNorm1d holder."""
    B, _, W2, _ = x.shape
    feat = torch.empty(B, 512, device=x.device)
    ctrl = torch.empty(B, w2.shape[0], device=x.device)
    sc, sh = in_affine if in_affine is not None else (None, None)
    check(lib.dpmn_stn_fc_f32(dptr(x), dptr(sc, True), dptr(sh, True), W2, dptr(w1t), dptr(b1), dptr(bn.weight), dptr(bn.bias),
                              dptr(bn.running_mean), dptr(bn.running_var), int(training), float(bn.momentum), float(bn.eps),
                              dptr(w2), dptr(b2), dptr(feat), dptr(ctrl), B, w2.shape[0], stream()))
    return feat, ctrl


def tps_sample(img, ctrl, inverse_kernel, coord_repr, out_hw):
    """TPSSpatialTransformer.forward (tps_spatial_transformer.py:97-112) -> (warped (B,C,H,W), source_coordinate (B,H*W,2))."""
    B, Cc, Hin, Win = img.shape
    H, W = out_hw
    out = torch.empty(B, Cc, H, W, device=img.device)
    src = torch.empty(B, H * W, 2, device=img.device)
    check(lib.dpmn_tps_sample_f32(dptr(img), dptr(ctrl), dptr(inverse_kernel), dptr(coord_repr), dptr(out), dptr(src), B, Cc,
                                  Hin, Win, H, W, ctrl.shape[1], stream()))
    return out, src


def dropout(x, p_elem=0.0, seed_elem=0, p_row=0.0, seed_row=0, row_len=0, res=None, out=None):
    """out = res + x * dropout_mask(p_elem) * droppath_mask(p_row per row_len elements); in place on x unless `out` is given."""
    y = x if out is None else out
    check(lib.dpmn_dropout_f32(dptr(x), dptr(res, True), dptr(y), x.numel(), int(row_len), float(p_elem), int(seed_elem),
                               float(p_row), int(seed_row), stream()))
    return y


def sk_fuse(cat, shortcut, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b, head_w, head_b, groups):
    """shortcut + SKConv(cat)  (pgrm.py:79-96 + 329) on (B, L, C) tokens."""
    B, L, Cd = cat.shape
    M = B * L
    feats = torch.empty_like(cat)
    parts = (L + 31) // 32
    partial = torch.empty(B * parts, Cd, device=cat.device)
    avec = torch.empty(B, groups, Cd // groups, device=cat.device)
    out = torch.empty_like(cat)
    check(lib.dpmn_sk_proj_f32(dptr(cat), dptr(proj_w), dptr(proj_b), dptr(feats), dptr(partial), M, Cd, stream()))
    check(lib.dpmn_sk_gate_f32(dptr(partial), parts, L, dptr(fc1_w), dptr(fc1_b), dptr(fc2_w), dptr(fc2_b), dptr(avec), B,
                               Cd, groups, fc1_w.shape[0], stream()))
    check(lib.dpmn_sk_select_f32(dptr(cat), dptr(avec), dptr(head_w), dptr(head_b), dptr(feats), dptr(shortcut), dptr(out),
                                 M, L, Cd, groups, stream()))
    return out, avec


def sk_mlp_in_supported(M, L, Cd, groups, N):
    return bool(lib.dpmn_sk_mlp_in_supported(M, L, Cd, groups, N))


def sk_mlp_in(cat, avec, head_w, head_b, feats, shortcut, ln_w, ln_b, fc1_w, fc1_b, L, save=False, eps=1e-5, p_row=0.0, seed_row=0):
    """One launch for x1 = proj_head(sum_g avec_g cat_g) + b + feats + shortcut and y = fc1(LayerNorm(x1)) on (M, C) rows.
    Returns (x1, y) or, save=True (training forward), (x1, y, V, n2).  p_row > 0: DropPath (one draw per sample) on the
    attention branch, x1 = shortcut + m_b (proj_head(...) + b + feats)."""
    M, Cd = cat.shape
    G, N = avec.shape[1], fc1_w.shape[0]
    x1, y = torch.empty_like(cat), torch.empty(M, N, device=cat.device)
    V = torch.empty(M, Cd // G, device=cat.device) if save else None
    n2 = torch.empty_like(cat) if save else None
    check(lib.dpmn_sk_mlp_in_drop_f32(dptr(cat), dptr(avec), dptr(head_w), dptr(head_b), dptr(feats), dptr(shortcut), dptr(x1), dptr(ln_w), dptr(ln_b),
                                      eps, dptr(fc1_w), dptr(fc1_b, True), dptr(y), dptr(V, True), dptr(n2, True), M, L, Cd, G, N, float(p_row),
                                      int(seed_row), stream()))
    return (x1, y, V, n2) if save else (x1, y)


def sk_fuse_mlp_in(cat, shortcut, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b, head_w, head_b, groups, ln_w, ln_b, mlp_fc1_w, mlp_fc1_b,
                   eps=1e-5):
    """x1 = shortcut + SKConv(cat) and y = Mlp.fc1(LayerNorm2(x1)) (no activation) with the select / proj_head / residual /
    LayerNorm / fc1 part in ONE launch (dpmn_sk_mlp_in_f32; pgrm.py:79-96, 327-331, 31).  Returns (x1, y)."""
    B, L, Cd = cat.shape
    M, N = B * L, mlp_fc1_w.shape[0]
    feats = torch.empty_like(cat)
    parts = (L + 31) // 32
    partial = torch.empty(B * parts, Cd, device=cat.device)
    avec = torch.empty(B, groups, Cd // groups, device=cat.device)
    x1 = torch.empty_like(cat)
    y = torch.empty(B, L, N, device=cat.device)
    check(lib.dpmn_sk_proj_f32(dptr(cat), dptr(proj_w), dptr(proj_b), dptr(feats), dptr(partial), M, Cd, stream()))
    check(lib.dpmn_sk_gate_f32(dptr(partial), parts, L, dptr(fc1_w), dptr(fc1_b), dptr(fc2_w), dptr(fc2_b), dptr(avec), B,
                               Cd, groups, fc1_w.shape[0], stream()))
    check(lib.dpmn_sk_mlp_in_f32(dptr(cat), dptr(avec), dptr(head_w), dptr(head_b), dptr(feats), dptr(shortcut), dptr(x1), dptr(ln_w), dptr(ln_b),
                                 eps, dptr(mlp_fc1_w), dptr(mlp_fc1_b, True), dptr(y), None, None, M, L, Cd, groups, N, stream()))
    return x1, y


def dwconv3x3_gelu(y, w, bias, r):
    B, L, Ch = y.shape
    g = torch.empty_like(y)
    check(lib.dpmn_dwconv3x3_gelu_f32(dptr(y), dptr(w), dptr(bias), dptr(g), B, Ch, r, stream()))
    return g


def pointwise(g, w, bias):
    B, L, Ch = g.shape
    z = torch.empty_like(g)
    check(lib.dpmn_pointwise_f32(dptr(g), dptr(w), dptr(bias), dptr(z), B, Ch, L, stream()))
    return z


def pgrm_tail(tokens, w0, b0, w1, b1, weight_list, residuals, H, W, hidden, patch):
    B, L, Cd = tokens.shape
    mid = torch.empty(B * L * hidden * patch * patch + 16 * (9 * Cd + 32), device=tokens.device)
    out = torch.empty(B, hidden, H * patch, W * patch, device=tokens.device)
    check(lib.dpmn_pgrm_tail_f32(dptr(tokens), dptr(w0), dptr(b0), dptr(w1), dptr(b1), _abi.ptr_array(weight_list),
                                 _abi.ptr_array(residuals), len(residuals), dptr(mid), dptr(out), B, H, W, Cd, hidden,
                                 patch, stream()))
    return out


# ------------------------------------------------------------------------------ conv family
_SPLITK_WS = {}


def splitk_workspace(device, floats=24 << 20):
    """One 96 MB scratch per (device, stream) for split-K partial sums (stream-ordered reuse: every conv consumes it before
    the next launch on the same stream; the two refinement branches run on two streams, interfaces/super_resolution.py)."""
    key = (device.type, device.index, _abi_stream_of(device))
    if key not in _SPLITK_WS:
        _SPLITK_WS[key] = torch.empty(floats, device=device)
    return _SPLITK_WS[key]


_ARRIVE_CNT = {}
ARRIVE_CNT_LEN = 32768      # 128 KB: the in-launch split-K reduction keeps its arrival words one 128-byte line apart
STREAM_K = True      # False: the fixed-split path with its reduce launch (tests / A-B runs)


def _attach_workspace(d, device):
    """split-K scratch + the zero-initialised tile-arrival counters of the stream-K conv launch (include/dpmn_hip.h
    dpmn_conv_desc.arrive_cnt: the kernel leaves them zero), one pair per (device, stream)."""
    ws = splitk_workspace(device)
    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
    if not STREAM_K:
        d.arrive_cnt, d.arrive_cnt_len = None, 0
        return
    key = (device.type, device.index, _abi_stream_of(device))
    if key not in _ARRIVE_CNT:
        _ARRIVE_CNT[key] = torch.zeros(ARRIVE_CNT_LEN, dtype=torch.int32, device=device)
    d.arrive_cnt, d.arrive_cnt_len = _ARRIVE_CNT[key].data_ptr(), ARRIVE_CNT_LEN


def cmm_forward(weights, keep, x1, x2, c_img, workspaces):
    """ComplementationModulationModule.forward in eval mode as ONE native call (csrc/cmm_forward.hip).  weights: a filled
    _abi.CmmWeights (keep = the tensors it points to); workspaces: dict (B, H, W, stream) -> activation workspace, owned by the module."""
    import ctypes as _C
    B, _, H, W = x1.shape
    x1, x2 = x1.contiguous().float(), x2.contiguous().float()
    key = (B, H, W, _abi_stream_of(x1.device))
    if key not in workspaces:
        workspaces[key] = torch.empty(lib.dpmn_cmm_workspace_bytes(_C.byref(weights), B) // 4, device=x1.device)
    ws = workspaces[key]
    d = _abi.ConvDesc()
    _attach_workspace(d, x1.device)
    sc = _abi.CmmScratch(d.splitk_ws, d.splitk_ws_bytes, d.arrive_cnt, d.arrive_cnt_len)
    out = torch.empty(B, c_img, H, W, device=x1.device)
    check(lib.dpmn_cmm_forward_f32(_C.byref(weights), dptr(x1), dptr(x2), dptr(out), dptr(ws), ws.numel() * 4, _C.byref(sc), B, stream()))
    return out


def psn_trunk(weights, keep, b1, tp, in_planes, workspaces):
    """SRBs + tail of the frozen PSN (TSRN / TATT trunk) as ONE native call (csrc/psn_forward.hip).  b1: block1's NHWC output,
    tp: TATT's NHWC text-prior map or None; returns the NCHW image (B, in_planes, 2H, 2W)."""
    import ctypes as _C
    B, H, W, _ = b1.shape
    key = (B, H, W, _abi_stream_of(b1.device))
    if key not in workspaces:
        workspaces[key] = torch.empty(lib.dpmn_psn_trunk_workspace_bytes(_C.byref(weights), B, H, W) // 4, device=b1.device)
    ws = workspaces[key]
    d = _abi.ConvDesc()
    _attach_workspace(d, b1.device)
    sc = _abi.CmmScratch(d.splitk_ws, d.splitk_ws_bytes, d.arrive_cnt, d.arrive_cnt_len)
    out = torch.empty(B, in_planes, 2 * H, 2 * W, device=b1.device)
    check(lib.dpmn_psn_trunk_f32(_C.byref(weights), dptr(b1), dptr(tp, True), 0 if tp is None else tp.shape[3], dptr(out), dptr(ws),
                                 ws.numel() * 4, _C.byref(sc), B, H, W, stream()))
    return out


def nchw_to_nhwc(x, cpad=None):
    B, Cc, H, W = x.shape
    cpad = cpad or Cc
    out = torch.empty(B, H, W, cpad, device=x.device)
    check(lib.dpmn_nchw_to_nhwc_f32(dptr(x), dptr(out), B, Cc, H, W, cpad, stream()))
    return out


def nhwc_to_nchw(x):
    B, H, W, Cc = x.shape
    out = torch.empty(B, Cc, H, W, device=x.device)
    check(lib.dpmn_nhwc_to_nchw_f32(dptr(x), dptr(out), B, Cc, H, W, stream()))
    return out


def conv_desc(inputs, k, stride=1, pad=0, dil=1, cout=0, pro_act="none", affine=None, phase=None, geom=None):
    """Geometry part of a dpmn_conv_desc.  geom: optional dict overriding (stride, dil, pad_y, pad_x, Hp, Wp, Hout, Wout,
    ostep, ooy, oox) for the odd-pixel scatter of the dilated stride-2 conv's data gradient."""
    d = _abi.ConvDesc()
    B, Hin, Win, _ = inputs[0].shape
    for i, t in enumerate(inputs):
        d.inp[i] = dptr(t)
        d.cseg[i] = t.shape[3]
        if affine is not None and affine[i] is not None:
            d.in_scale[i], d.in_shift[i] = dptr(affine[i][0]), dptr(affine[i][1])
    kh, kw = (k, k) if isinstance(k, int) else k
    d.B, d.Hin, d.Win, d.KH, d.KW = B, Hin, Win, kh, kw
    if geom is not None:
        for key, val in geom.items():
            setattr(d, key, val)
    elif phase is None:
        ph, pw = (pad, pad) if isinstance(pad, int) else pad
        d.stride, d.dil_y, d.dil_x, d.pad_y, d.pad_x = stride, dil, dil, ph, pw
        Ho = (Hin + 2 * ph - dil * (kh - 1) - 1) // stride + 1
        Wo = (Win + 2 * pw - dil * (kw - 1) - 1) // stride + 1
        d.Hp, d.Wp, d.Hout, d.Wout, d.ostep, d.ooy, d.oox = Ho, Wo, Ho, Wo, 1, 0, 0
    else:
        py, px = phase
        d.stride, d.dil_y, d.dil_x, d.pad_y, d.pad_x = 1, -1, -1, -py, -px
        d.Hp, d.Wp, d.Hout, d.Wout, d.ostep, d.ooy, d.oox = Hin, Win, 2 * Hin, 2 * Win, 2, py, px
    d.pro_act, d.Cout = ACT[pro_act], cout
    return d


def _stats_ptr(stats, cout):
    """BatchNorm-statistics accumulator of a conv launch: (32, 2, cout) float64 (or a float32 buffer of twice as many elements
    that is zero: the kernels add into it with fp64 atomics, include/dpmn_hip.h dpmn_conv_desc.stats)."""
    if stats is None:
        return None
    ok = stats.is_cuda and stats.is_contiguous() and stats.data_ptr() % 8 == 0 and (
        (stats.dtype == torch.float64 and stats.numel() >= 64 * cout) or (stats.dtype == torch.float32 and stats.numel() >= 128 * cout))
    if not ok:
        raise _abi.DpmnError("conv2d: stats must be a contiguous CUDA (32, 2, Cout) float64 accumulator")
    return stats.data_ptr()


def conv2d(inputs, wp, bias, cout, k, stride=1, pad=0, dil=1, pro_act="none", epi_act="none", slope=0.0, res=None,
           affine=None, out=None, out_nchw=False, pixel_shuffle=False, stats=None, phase=None, geom=None, out_coff=None, groups=1):
    """inputs: list of 1..3 NHWC tensors (channel-concatenated on the fly).  k: int or (KH, KW).
    phase=(py, px): one output phase of ConvTranspose2d(4,2,1) (k=2, dil=-1, pad=-phase).
    affine: optional list of (scale, shift) per input segment.
    groups=2: wp (2, Cout, Kp) / bias (2, Cout) -- images [B/2, B) are convolved with the second weight set (the CMM's twin
    encoder branches in one launch); falls back to two launches over the batch halves when a half does not fill whole
    128-pixel row tiles."""
    if groups == 2:
        B0, h = inputs[0].shape[0], inputs[0].shape[0] // 2
        d0 = conv_desc(inputs, k, stride, pad, dil, cout, pro_act, affine, phase, geom)
        if B0 % 2 or (h * d0.Hp * d0.Wp) % (64 if h * d0.Hp * d0.Wp <= 512 else 128) or stats is not None or res is not None or out_nchw or pixel_shuffle:
            assert out is None and out_coff is None and B0 % 2 == 0
            outs = [conv2d([t[g * h:(g + 1) * h] for t in inputs], wp[g], None if bias is None else bias[g], cout, k, stride, pad, dil,
                           pro_act, epi_act, slope, None if res is None else res[g * h:(g + 1) * h], affine, None, out_nchw,
                           pixel_shuffle, stats, phase, geom) for g in range(2)]
            return torch.cat(outs, 0)
    d = conv_desc(inputs, k, stride, pad, dil, cout, pro_act, affine, phase, geom)
    if groups == 2:
        assert wp.dim() == 3 and wp.is_contiguous() and (bias is None or (bias.dim() == 2 and bias.is_contiguous()))
        d.groups, d.w_group_stride = 2, wp.shape[1] * wp.shape[2]
    B = d.B
    Ho, Wo = d.Hout, d.Wout
    d.epi_act, d.slope = ACT[epi_act], float(slope)
    d.w, d.bias = dptr(wp), dptr(bias, True)
    d.res = dptr(res, True)
    if out is None:
        if out_nchw:
            out = torch.empty(B, cout, Ho, Wo, device=wp.device)
        elif pixel_shuffle:
            out = torch.empty(B, 2 * Ho, 2 * Wo, cout // 4, device=wp.device)
        else:
            out = torch.empty(B, Ho, Wo, cout, device=wp.device)
    d.out = dptr(out)
    d.out_ld, d.out_coff, d.out_nchw, d.pixel_shuffle = 0, 0, int(out_nchw), int(pixel_shuffle)
    if out_coff is not None:      # write channels [out_coff, out_coff + cout) of a wider NHWC buffer
        d.out_ld, d.out_coff = out.shape[3], int(out_coff)
    d.stats = _stats_ptr(stats, cout)
    _attach_workspace(d, wp.device)
    import ctypes as _C
    check(lib.dpmn_conv2d_nhwc_f32(_C.byref(d), stream()))
    return out


def stack_phase_packs(packs):
    """[(wp, bias)] * 4 from packing.pack_convT_s2k4 -> ((4, Cout, Kp) contiguous, bias) for the phase-fused launch."""
    return torch.stack([p[0] for p in packs]).contiguous(), packs[0][1]


def convT_s2k4(inputs, packs, cout, pro_act="none", affine=None, stats=None):
    """ConvTranspose2d(4, stride 2, padding 1): all 4 output phases in ONE launch (nphase = 4).
    packs: ((4, Cout, Kp) phase-major packed weights, bias or None) -- see stack_phase_packs / packing.tpack_convT_s2k4."""
    if isinstance(packs, list):
        packs = stack_phase_packs(packs)
    wp4, bias = packs
    B, Hin, Win, _ = inputs[0].shape
    out = torch.empty(B, 2 * Hin, 2 * Win, cout, device=inputs[0].device)
    d = conv_desc(inputs, 2, cout=cout, pro_act=pro_act, affine=affine, phase=(0, 0))
    d.nphase, d.w_phase_stride = 4, wp4.shape[1] * wp4.shape[2]
    d.w, d.bias = dptr(wp4), dptr(bias, True)
    d.out, d.stats = dptr(out), _stats_ptr(stats, cout)
    _attach_workspace(d, out.device)
    import ctypes as _C
    check(lib.dpmn_conv2d_nhwc_f32(_C.byref(d), stream()))
    return out


def se_gate(x, fc1_w, fc1_b, fc2_w, fc2_b):
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    hid = torch.empty(B, fc1_w.shape[0], device=x.device)
    check(lib.dpmn_se_gate_f32(dptr(x), dptr(fc1_w), dptr(fc1_b), dptr(fc2_w), dptr(fc2_b), dptr(out), dptr(hid), B, H * W,
                               Cc, fc1_w.shape[0], stream()))
    return out


# ------------------------------------------------------------------------------ TSRN / TATT pieces
def bigru(gi, w_hh, b_hh, B, H, W, axis, res=None, hidden=32):
    """gi: NHWC (B,H,W,6*hidden) input projections.  axis='w': sequences run along W (rows as batch);
    axis='h': along H (the transposed gru1 call).  Returns NHWC (B,H,W,2*hidden) (+ res)."""
    out = torch.empty(B, H, W, 2 * hidden, device=gi.device)
    if axis == "w":
        nseq, T, inner, outer, inner_s, step = B * H, W, 1, W, 0, 1
    else:
        nseq, T, inner, outer, inner_s, step = B * W, H, W, H * W, 1, W
    check(lib.dpmn_bigru_f32(dptr(gi), dptr(w_hh), dptr(b_hh), dptr(res, True), dptr(out), nseq, T, inner, outer, inner_s,
                             step, hidden, stream()))
    return out


def small_linear(x, w, b=None, add=None, act="none", slope=0.0):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device)
    check(lib.dpmn_small_linear_f32(dptr(x), dptr(add, True), 0 if add is None else add.shape[0], dptr(w), dptr(b, True),
                                    dptr(y), M, N, K, ACT[act], float(slope), stream()))
    return y


def tatt_encoder_layer(src, pos, w12, nhead=4):
    N, L, E = src.shape
    mem = torch.empty_like(src)
    check(lib.dpmn_tatt_encoder_layer_f32(dptr(src), dptr(pos), _abi.ptr_array(w12), dptr(mem), N, L, E, nhead, stream()))
    return mem


def cross_attn(q, k, v, nhead=4, need_weights=False):
    N, L, E = q.shape
    S = k.shape[1]
    o = torch.empty_like(q)
    pw = torch.empty(N, L, S, device=q.device) if need_weights else None
    check(lib.dpmn_cross_attn_f32(dptr(q), dptr(k), dptr(v), dptr(o), dptr(pw, True), N, L, S, E, nhead, stream()))
    return o, pw


def add_layernorm64(x, res, g, b, g2=None, b2=None, acc_out=None, alpha=1.0, accumulate=False):
    M = x.numel() // 64
    y = torch.empty_like(x)
    check(lib.dpmn_add_layernorm64_f32(dptr(x), dptr(res, True), dptr(g), dptr(b), dptr(y), dptr(g2, True), dptr(b2, True),
                                       dptr(acc_out, True), float(alpha), int(accumulate), M, stream()))
    return y


def gru_gate(gi, gh, h, hist, hist_row_stride):
    R, H = h.shape
    check(lib.dpmn_gru_gate_f32(dptr(gi), dptr(gh), dptr(h), hist.data_ptr(), hist_row_stride, R, H, stream()))


# ------------------------------------------------------------------------------ image-space helpers
def _nchw_view(x):
    """(pointer, per-image stride) of an NCHW tensor whose images are contiguous (channel-sliced views allowed)."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise _abi.DpmnError("dpmn_amd: expected a float32 CUDA tensor")
    B, Cc, H, W = x.shape
    if x.stride()[1:] != (H * W, W, 1):
        x = x.contiguous()
    return x, x.data_ptr(), x.stride(0)


def to_mask(img):
    """toMask (utils/util.py:27-35) for a whole batch; img (B,>=3,H,W) -> (B,3,H,W) in {0,1}."""
    img, p, st = _nchw_view(img)
    B, _, H, W = img.shape
    out = torch.empty(B, 3, H, W, device=img.device)
    check(lib.dpmn_to_mask_f32(p, st, dptr(out), B, H, W, stream()))
    return out


def mha32(qkv, B, L, heads, scale):
    """MultiHeadedAttention core of TBSRN (tbsrn.py:110-150): qkv (B*L, 3*heads*32) -> (B*L, heads*32)."""
    out = torch.empty(B * L, heads * 32, device=qkv.device)
    check(lib.dpmn_mha32_f32(dptr(qkv), dptr(out), B, L, heads, float(scale), stream()))
    return out


def layernorm_std(x, a2, b2, eps=1e-6):
    """tbsrn.py:23-36 LayerNorm: a2 * (x - mean) / (unbiased std + eps) + b2 over the last axis of (M, C)."""
    y = torch.empty_like(x)
    check(lib.dpmn_layernorm_std_f32(dptr(x), dptr(a2), dptr(b2), float(eps), dptr(y), x.shape[0], x.shape[1], stream()))
    return y


def rotate_img(img, arc, rand_offs, off_range=0.2):
    """torch_rotate_img (utils/util.py:37-58): img (N,C,H,W), arc / rand_offs (N) -> rotated, aspect-jittered batch."""
    img = img.contiguous().float()
    N, Cc, H, W = img.shape
    out = torch.empty_like(img)
    check(lib.dpmn_rotate_img_f32(dptr(img), dptr(arc.contiguous().float()), dptr(rand_offs.contiguous().float()), float(off_range),
                                  dptr(out), N, Cc, H, W, stream()))
    return out


def blend(a, b, alpha):
    """alpha*a + (1-alpha)*b[:, :C] (super_resolution.py:449)."""
    a, pa, sa = _nchw_view(a)
    b, pb, sb = _nchw_view(b)
    B, Cc, H, W = a.shape
    out = torch.empty(B, Cc, H, W, device=a.device)
    check(lib.dpmn_blend_f32(pa, sa, pb, sb, dptr(out), float(alpha), B, Cc * H * W, stream()))
    return out


def psnr_ssim(x, y):
    """(psnr, ssim) device scalars over the first 3 channels (utils/ssim_psnr.py)."""
    x, px, sx = _nchw_view(x)
    y, py, sy = _nchw_view(y)
    B, _, H, W = x.shape
    ws = torch.empty(lib.dpmn_psnr_ssim_workspace_bytes(B, 3, H, W), dtype=torch.uint8, device=x.device)
    out = torch.empty(2, device=x.device)
    check(lib.dpmn_psnr_ssim_f32(px, sx, py, sy, dptr(out), ws.data_ptr(), B, 3, H, W, stream()))
    return out[0], out[1]


# ------------------------------------------------------------------ in-loop text prior (csrc/visionlan.hip)
def mha64(qkv, B, L, heads, scale):
    """VisionLAN MultiHeadAttention core (modules.py:43-81): qkv (B*L, 3*heads*64) rows [q|k|v] -> (B*L, heads*64)."""
    out = torch.empty(B * L, heads * 64, device=qkv.device)
    check(lib.dpmn_mha64_f32(dptr(qkv), dptr(out), B, L, heads, float(scale), stream()))
    return out


def layernorm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last axis of (M, C), C in {64, 96, 192, 512}."""
    y = torch.empty_like(x)
    check(lib.dpmn_layernorm_f32(dptr(x), dptr(w), dptr(b), float(eps), dptr(y), x.shape[0], x.shape[1], stream()))
    return y


def act(x, kind):
    y = torch.empty_like(x)
    check(lib.dpmn_act_fwd_f32(dptr(x), dptr(y), ACT[kind], 0.0, x.numel(), stream()))
    return y


def vl_resize(img, out_h=64, out_w=256):
    """parse_visionlan_data (base.py:473-478) for a batch: (B, >=3, H, W) -> NHWC (B, out_h, out_w, 4), channel 3 zero."""
    B, _, H, W = img.shape
    v, ptr, stride = _nchw_view(img)
    out = torch.empty(B, out_h, out_w, 4, device=img.device)
    check(lib.dpmn_vl_resize_f32(ptr, stride, dptr(out), B, H, W, out_h, out_w, stream()))
    return out


def vl_tokens(feat, pos_table):
    B, Hf, Wf, Cc = feat.shape
    tok = torch.empty(B, Hf * Wf, Cc, device=feat.device)
    check(lib.dpmn_vl_tokens_f32(dptr(feat), dptr(pos_table), dptr(tok), B, Hf, Wf, Cc, stream()))
    return tok


def vl_pp_pool(scores, enc, w_vrm, b_vrm, n_steps):
    B, L, Cc = enc.shape
    n_class = w_vrm.shape[0]
    logits = torch.empty(B, n_steps, n_class, device=enc.device)
    check(lib.dpmn_vl_pp_pool_f32(dptr(scores), scores.shape[1], dptr(enc), dptr(w_vrm), dptr(b_vrm), dptr(logits), B, L, Cc, n_steps,
                                  n_class, stream()))
    return logits


def vl_decode(logits, max_len=25):
    B, n_steps, n_class = logits.shape
    cls = torch.empty(B, max_len, dtype=torch.int32, device=logits.device)
    length = torch.empty(B, dtype=torch.int32, device=logits.device)
    check(lib.dpmn_vl_decode_i32(dptr(logits), cls.data_ptr(), length.data_ptr(), B, n_steps, n_class, max_len, stream()))
    return cls, length


def text_prior_compose(cls, length, atlas, advance, out_h, out_w):
    """cls (B, max_len) int32, length (B) int32, atlas (2, n_glyph, GH, GW) float 0..255, advance (2, n_glyph) int32."""
    B, max_len = cls.shape
    _, n_glyph, GH, GW = atlas.shape
    out = torch.empty(B, 2, out_h, out_w, device=atlas.device)
    for t_ in (cls, length, advance):
        if not (t_.is_cuda and t_.dtype == torch.int32 and t_.is_contiguous()):
            raise _abi.DpmnError("dpmn_amd: text_prior_compose expects contiguous int32 CUDA index tensors")
    check(lib.dpmn_text_prior_compose_f32(cls.data_ptr(), length.data_ptr(), dptr(atlas), advance.data_ptr(), dptr(out), B, max_len,
                                          n_glyph, GH, GW, out_h, out_w, stream()))
    return out

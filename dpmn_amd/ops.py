"""Thin Python wrappers over the per-kernel C-ABI entry points (include/dpmn_hip.h).

Used by the parity tests and by host code that composes kernels outside the native module
drivers.  Every wrapper allocates its output with torch (device memory plumbing only) and
enqueues on the current HIP stream.
"""
import torch

from . import _abi
from ._abi import dptr, lib, check, stream

ACT = {"none": 0, "gelu": 1, "relu": 2, "leaky02": 3, "leaky001": 4, "mish": 5, "prelu": 6, "tanh": 7, "sigmoid": 8}


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


def ln_linear(x, ln_w, ln_b, w, bias, act="none", eps=1e-5):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device)
    check(lib.dpmn_ln_linear_f32(dptr(x), dptr(ln_w), dptr(ln_b), eps, dptr(w), dptr(bias, True), dptr(y), M, N, K,
                                 ACT[act], stream()))
    return y


def patch_embed_ln(img, pe_w, pe_b, ln_w, ln_b, patch, pf_w=None, pf_b=None):
    B, cin, Hi, Wi = img.shape
    Cd = pe_w.shape[0]
    tok = torch.empty(B, (Hi // patch) * (Wi // patch), Cd, device=img.device)
    check(lib.dpmn_patch_embed_ln_f32(dptr(img), cin, dptr(pf_w, True), dptr(pf_b, True), dptr(pe_w), dptr(pe_b),
                                      dptr(ln_w), dptr(ln_b), dptr(tok), B, Hi, Wi, patch, Cd, stream()))
    return tok


def window_attn(q, kv, tables, windows, shifts, heads_per_group, H, W):
    B, L, Cd = q.shape
    out = torch.empty_like(q)
    check(lib.dpmn_window_attn_f32(dptr(q), dptr(kv), _abi.ptr_array(tables), _abi.int_array(windows),
                                   _abi.int_array(shifts), len(windows), heads_per_group, dptr(out), B, H, W, Cd, stream()))
    return out


def sk_fuse(cat, shortcut, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b, head_w, head_b, groups):
    """shortcut + SKConv(cat)  (pgrm.py:79-96 + 329) on (B, L, C) tokens."""
    B, L, Cd = cat.shape
    M = B * L
    feats = torch.empty_like(cat)
    parts = (L + 63) // 64
    partial = torch.empty(B * parts, Cd, device=cat.device)
    avec = torch.empty(B, groups, Cd // groups, device=cat.device)
    out = torch.empty_like(cat)
    check(lib.dpmn_sk_proj_f32(dptr(cat), dptr(proj_w), dptr(proj_b), dptr(feats), dptr(partial), M, Cd, stream()))
    check(lib.dpmn_sk_gate_f32(dptr(partial), parts, L, dptr(fc1_w), dptr(fc1_b), dptr(fc2_w), dptr(fc2_b), dptr(avec), B,
                               Cd, groups, fc1_w.shape[0], stream()))
    check(lib.dpmn_sk_select_f32(dptr(cat), dptr(avec), dptr(head_w), dptr(head_b), dptr(feats), dptr(shortcut), dptr(out),
                                 M, L, Cd, groups, stream()))
    return out, avec


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
    mid = torch.empty(B * L * hidden * patch * patch, device=tokens.device)
    out = torch.empty(B, hidden, H * patch, W * patch, device=tokens.device)
    check(lib.dpmn_pgrm_tail_f32(dptr(tokens), dptr(w0), dptr(b0), dptr(w1), dptr(b1), _abi.ptr_array(weight_list),
                                 _abi.ptr_array(residuals), len(residuals), dptr(mid), dptr(out), B, H, W, Cd, hidden,
                                 patch, stream()))
    return out

#!/usr/bin/env python3
"""Stand-alone timing of the fused attention backward (and forward) at the headline shape: B = 48, 16x64 tokens, dim 96,
windows 2/4/8.  usage: python tools/prof_attn_bwd.py [shift 0|1] [p_drop]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpmn_amd import ops

dev = torch.device("cuda:0")
B, H, W, C = 48, 16, 64, 96
shifted = len(sys.argv) > 1 and sys.argv[1] == "1"
pd = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
g = torch.Generator(device="cpu").manual_seed(1)
r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
tq, tkv, dout = r(B, H * W, C), r(B, H * W, C), r(B, H * W, C)
lnw, lnb = r(C) * 0.2 + 1, r(C) * 0.1
wq, bq, wkv, bkv = r(C, C) * 0.1, r(C) * 0.1, r(2 * C, C) * 0.1, r(2 * C) * 0.1
wins = [2, 4, 8]
tables = [r((2 * w - 1) ** 2, 2) * 0.2 for w in wins]
shifts = [1, 2, 4] if shifted else [0, 0, 0]
args = (tq, tkv, lnw, lnb, lnw, lnb, wq, bq, wkv, bkv, tables, wins, shifts, 2, H, W)
fold = []
ops.ln_qkv_window_attn_train(*args, p_drop=pd, seed=3, save_qkv=False, fold_out=fold)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd = timeit(lambda: ops.ln_qkv_window_attn_train(*args, p_drop=pd, seed=3, save_qkv=False))
fwd_s = timeit(lambda: ops.ln_qkv_window_attn_train(*args, p_drop=pd, seed=3, save_qkv=True))
bwd = timeit(lambda: ops.ln_qkv_window_attn_bwd(*args, dout, p_drop=pd, seed=3, fold=fold[0]))
print("shift %d drop %.2f env %s: forward %.1f us (saving q/kv %.1f us), backward %.1f us" % (
    shifted, pd, {k: v for k, v in os.environ.items() if k.startswith("DPMN_FA")}, fwd, fwd_s, bwd))

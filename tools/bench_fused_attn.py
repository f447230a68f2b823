#!/usr/bin/env python3
"""Times k_ln_qkv_window_attn alone (HIP events inside the library) at the bench shape: python tools/bench_fused_attn.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops, _abi
from dpmn_amd.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H, W, C = 16, 64, 96
dev = torch.device("cuda:0")
u = lambda n, s, lo=-1, hi=1: synth.uniform(n, s, lo, hi, 3).to(dev)
tq, tkv = u("tq", (B, H * W, C)), u("tkv", (B, H * W, C))
ln = [u("a", (C,), .5, 1.5), u("b", (C,)), u("c", (C,), .5, 1.5), u("d", (C,))]
wq, bq, wkv, bkv = u("wq", (C, C), -.1, .1), u("bq", (C,)), u("wkv", (2 * C, C), -.1, .1), u("bkv", (2 * C,))
tables = [u("t%d" % i, ((2 * w - 1) ** 2, 2)) for i, w in enumerate((2, 4, 8))]
for shifts in ([0, 0, 0], [1, 2, 4]):
    for _ in range(20):
        ops.ln_qkv_window_attn(tq, tkv, *ln, wq, bq, wkv, bkv, tables, [2, 4, 8], shifts, 2, H, W)
    torch.cuda.synchronize()
    _abi.profile_begin(["k_ln_qkv_window_attn"])
    for _ in range(50):
        ops.ln_qkv_window_attn(tq, tkv, *ln, wq, bq, wkv, bkv, tables, [2, 4, 8], shifts, 2, H, W)
    torch.cuda.synchronize()
    r = _abi.profile_end()[0]
    print("dbg=%s shifts=%s: %.1f us/launch  %.1f TFLOP/s" % (os.environ.get("DPMN_FUSED_DBG", "0"), shifts, r["total_ms"] * 1e3 / r["launches"],
                                                          r["flops"] / (r["total_ms"] * 1e-3) / 1e12))

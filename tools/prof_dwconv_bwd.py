#!/usr/bin/env python3
"""Stand-alone timing of the fused depthwise-conv backward at the headline shape (B = 48, 384 channels, 32 x 32 planes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpmn_amd._abi import lib, check, dptr
from dpmn_amd import ops
dev = torch.device("cuda:0")
B, Ch, r = 48, 384, 32
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.rand(*s, generator=g) * 4 - 2).to(dev)
P, dg, gpre, w = rnd(B, Ch, r, r), rnd(B, Ch, r, r), rnd(B, Ch, r, r), rnd(Ch, 9)
dP, dw, db = torch.empty_like(P), torch.zeros(Ch, 9, device=dev), torch.zeros(Ch, device=dev)
ws = torch.empty(B * Ch * 10, device=dev)
def run():
    check(lib.dpmn_dwconv3x3_bwd_fused_det_f32(dptr(P), dptr(dg), dptr(gpre), dptr(w), dptr(dP), dptr(dw), dptr(db), 1, 1, 0.0, 0, B, Ch, r,
                                               dptr(ws), ws.numel() * 4, ops.stream()))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print("dwconv3x3 backward (+ rows reduce): %.1f us per call, %.2f TB/s on 4 x %.1f MB" % (us, 4 * P.numel() * 4 / us / 1e6, P.numel() * 4 / 1e6))

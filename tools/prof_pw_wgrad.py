"""Pointwise-conv weight gradient (dpmn_pointwise_wgrad_det_f32) at the bench shape: us per call, TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd._abi import lib, dptr, check, stream
dev = torch.device("cuda:0")
B, Ch, L = 48, 384, 1024
dz, g = torch.randn(B, Ch, L, device=dev), torch.randn(B, Ch, L, device=dev)
dw = torch.zeros(Ch, Ch, device=dev)
nb = lib.dpmn_pointwise_wgrad_det_bytes(Ch, L)
ws = torch.empty(nb // 4, device=dev)
f = lambda: check(lib.dpmn_pointwise_wgrad_det_f32(dptr(dz), dptr(g), dptr(dw), B, Ch, L, dptr(ws), nb, stream()))
for _ in range(5): f()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): f()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) * 1e3 / 30)
ref = torch.einsum("bms,bns->mn", dz[:4].double(), g[:4].double())
dw.zero_(); check(lib.dpmn_pointwise_wgrad_det_f32(dptr(dz[:4].contiguous()), dptr(g[:4].contiguous()), dptr(dw), 4, Ch, L, dptr(ws), nb, stream()))
print("workspace %.1f MB, %.1f us per call incl. the reduce, %.1f TFLOP/s, rel err (B=4) %.2e" % (nb / 1e6, best, 2.0 * Ch * Ch * B * L / best / 1e6,
      float((dw.double() - ref).abs().max() / ref.abs().max())))

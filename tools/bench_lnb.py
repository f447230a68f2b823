import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd._abi import lib, check, dptr, stream
dev = torch.device("cuda:0")
M, C = 49152, 96
x = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev); g = torch.ones(C, device=dev)
dx = torch.empty_like(x); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
for _ in range(30):
    check(lib.dpmn_layernorm_bwd_f32(dptr(x), dptr(dy), dptr(g), 1e-5, dptr(dx), 0, dptr(dg), dptr(db), M, C, stream()))
torch.cuda.synchronize()

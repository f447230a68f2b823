import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd._abi import lib, dptr, check, stream
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, Hi, Wi, C = 3, 32, 128, 96
M = B * (Hi // 2) * (Wi // 2)
img = torch.rand(B, 3, Hi, Wi, device=dev)
pe_w, pe_b, ln_w = torch.randn(C, 3, 2, 2, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
dtok = torch.randn(M, C, device=dev)
dconv, patches = torch.empty(M, C, device=dev), torch.empty(M, 16, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
check(lib.dpmn_patch_embed_bwd_f32(dptr(img), 3, None, None, dptr(pe_w), dptr(pe_b), dptr(ln_w), dptr(dtok), dptr(dconv), dptr(patches), dptr(dg), dptr(db), B, Hi, Wi, C, stream()))
lnp = torch.empty((M + 63) // 64, 2 * C, device=dev)
dconv2 = torch.empty_like(dconv)
check(lib.dpmn_patch_embed_bwd_det_f32(dptr(img), 3, None, None, dptr(pe_w), dptr(pe_b), dptr(ln_w), dptr(dtok), dptr(dconv2), dptr(patches), dptr(lnp), B, Hi, Wi, C, stream()))
torch.cuda.synchronize()
s = lnp.sum(0)
print("dconv equal", torch.equal(dconv, dconv2))
print("dgamma: atomic vs rowsum", float((dg - s[:C]).abs().max()), float(dg.abs().max()))
print("dbeta : atomic vs rowsum", float((db - s[C:]).abs().max()), float(db.abs().max()))
dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
check(lib.dpmn_rows_reduce_f32(dptr(lnp), dptr(dg2), dptr(db2), C, C, lnp.shape[0], stream()))
torch.cuda.synchronize()
print("rows_reduce vs rowsum", float((dg2 - s[:C]).abs().max()), float((db2 - s[C:]).abs().max()))
bad = (dg - s[:C]).abs() > 1e-3
print("bad channels", bad.nonzero().flatten().tolist()[:40])

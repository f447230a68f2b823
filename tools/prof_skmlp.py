"""Time dpmn_sk_mlp_in_f32 against dpmn_sk_select_f32 + dpmn_ln_linear_f32 at the bench shape (B = 48, L = 1024, C = 96 -> 384)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops
from dpmn_amd._abi import lib, dptr, check, stream
dev = torch.device("cuda:0")
B, L, C, N, G = 48, 1024, 96, 384, 3
M = B * L
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
cat, feats, sc = r(M, C), r(M, C), r(M, C)
avec = torch.softmax(r(B, G, C // G), 1).contiguous()
wh, bh, lnw, lnb, w1, b1 = r(C, C // G), r(C), r(C) + 1, r(C), r(N, C), r(N)
x1 = torch.empty(M, C, device=dev); y = torch.empty(M, N, device=dev); V = torch.empty(M, 32, device=dev); n2 = torch.empty(M, C, device=dev)
def two():
    check(lib.dpmn_sk_select_f32(dptr(cat), dptr(avec), dptr(wh), dptr(bh), dptr(feats), dptr(sc), dptr(x1), M, L, C, G, stream()))
    check(lib.dpmn_ln_linear_f32(dptr(x1), dptr(lnw), dptr(lnb), 1e-5, dptr(w1), dptr(b1), dptr(y), M, N, C, 0, stream()))
def one():
    check(lib.dpmn_sk_mlp_in_f32(dptr(cat), dptr(avec), dptr(wh), dptr(bh), dptr(feats), dptr(sc), dptr(x1), dptr(lnw), dptr(lnb), 1e-5, dptr(w1), dptr(b1),
                                 dptr(y), None, None, M, L, C, G, N, stream()))
def one_save():
    check(lib.dpmn_sk_mlp_in_f32(dptr(cat), dptr(avec), dptr(wh), dptr(bh), dptr(feats), dptr(sc), dptr(x1), dptr(lnw), dptr(lnb), 1e-5, dptr(w1), dptr(b1),
                                 dptr(y), dptr(V), dptr(n2), M, L, C, G, N, stream()))
def t(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best
print("two launches %.1f us | one %.1f us | one + V, n2 %.1f us" % (t(two), t(one), t(one_save)))

#!/usr/bin/env python3
"""Times the K <= 192 token GEMMs through the C ABI (events inside the library) with the rows-in-registers kernel (default) and,
re-executed with DPMN_ROWREG=0, with k_gemm_wstat; prints max |difference| between the two as a sanity check:
python tools/bench_rowreg.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops, _abi
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
B, L, C = 48, 1024, 96
M = B * L
u = lambda n, s, lo=-1, hi=1: synth.uniform(n, s, lo, hi, 3).to(dev)
x, x2, r1, r2 = u("x", (M, C)), u("x2", (M, 2 * C)), u("r1", (M, C)), u("r2", (M, C))
w, b = u("w", (C, C), -.1, .1), u("b", (C,))
w4, b4 = u("w4", (4 * C, C), -.1, .1), u("b4", (4 * C,))
w2 = u("w2", (C, 2 * C), -.1, .1)
wh, sel = u("wh", (C, 32), -.1, .1), u("sel", (B, 3, 32), 0, 1)
lnw, lnb = u("lnw", (C,), .5, 1.5), u("lnb", (C,))
cases = {
    "linear 96->96 bias": lambda: ops.linear(x, w, b),
    "linear 96->96 bias + res": lambda: ops.linear(x, w, b, res1=r1),
    "linear 96->96 2 res": lambda: ops.linear(x, w, b, res1=r1, res2=r2),
    "linear 192->96": lambda: ops.linear(x2, w2, None),
    "linear 96->384 bias": lambda: ops.linear(x, w4, b4),
    "ln_linear 96->384 gelu": lambda: ops.ln_linear(x, lnw, lnb, w4, b4, act="gelu"),
    "ln_linear 96->96": lambda: ops.ln_linear(x, lnw, lnb, w, b),
    "sk_fuse (proj + gate + select)": lambda: ops.sk_fuse(x.reshape(B, L, C), r1.reshape(B, L, C), w, b, u("f1", (16, C), -.1, .1), u("f1b", (16,)),
                                                          u("f2", (C, 16), -.1, .1), u("f2b", (C,)), wh, b, 3)[0],
}
outs = {}
for name, fn in cases.items():
    for _ in range(5):
        o = fn()
    torch.cuda.synchronize()
    _abi.profile_begin(None)
    for _ in range(20):
        o = fn()
    torch.cuda.synchronize()
    rows = _abi.profile_end()
    print("%-34s %s" % (name, "  ".join("%s %.1f us" % (r["kernel"], r["total_ms"] * 1e3 / r["launches"]) for r in rows)))
    outs[name] = o.float().cpu()
if os.environ.get("DPMN_ROWREG") is None:
    torch.save(outs, "/tmp/rowreg_outs.pt")
    subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, DPMN_ROWREG="0"))
elif os.path.exists("/tmp/rowreg_outs.pt"):
    ref = torch.load("/tmp/rowreg_outs.pt")
    for k in outs:
        print("max|rowreg - wstat| %-34s %.3e  (|ref| max %.2f)" % (k, float((ref[k] - outs[k]).abs().max()), float(outs[k].abs().max())))

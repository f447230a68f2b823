"""Debug: (1) x3 kernels under stream concurrency vs alone, (2) x3 halo conv vs fp32 over the cfg4 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops, _abi
from dpmn_amd.model import packing
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
u = lambda n, s, lo=-1.0, hi=1.0: synth.uniform(n, s, lo, hi, 5).to(dev)
mode = lambda m: _abi.check(_abi.lib.dpmn_set_compute_dtype(m))

# ---- (1) linear K = 384 (k-loop) and a 128-tile conv on two streams at once
x = [u("x%d" % i, (49152, 384)) for i in range(2)]
w, b = u("w", (96, 384), -0.1, 0.1), u("b", (96,))
res = [u("r%d" % i, (49152, 96)) for i in range(2)]
xc = [u("xc%d" % i, (8, 16, 64, 128)) for i in range(2)]
wc = u("wc", (256, 128, 4, 4), -0.05, 0.05)
wp, _ = packing.pack_conv(wc, None)
mode(2)
lin = lambda i: ops.linear(x[i], w, b, res1=res[i])
conv = lambda i: ops.conv2d([xc[i]], wp, None, 256, 4, stride=2, pad=3, dil=2, pro_act="leaky02")
alone = [(lin(i).clone(), conv(i).clone()) for i in range(2)]
torch.cuda.synchronize()
s = [torch.cuda.Stream() for _ in range(2)]
bad = [0, 0]
for rep in range(20):
    out = [None, None]
    for i in range(2):
        with torch.cuda.stream(s[i]):
            for _ in range(3):
                out[i] = (lin(i), conv(i))
    torch.cuda.synchronize()
    for i in range(2):
        bad[0] += int(not torch.equal(out[i][0], alone[i][0]))
        bad[1] += int(not torch.equal(out[i][1], alone[i][1]))
print("concurrent vs alone: linear mismatches %d / 40, conv mismatches %d / 40" % tuple(bad))
if bad[0]:
    d = (out[0][0] - alone[0][0]).abs()
    print("  linear max diff %.3e, rows affected %d" % (float(d.max()), int((d.amax(1) > 0).sum())), torch.nonzero(d.amax(1) > 0).reshape(-1)[:10].tolist())

# ---- (2) halo conv x3 vs f32
for B, H, W, segs, cout in [(2, 64, 256, (64,), 64), (2, 64, 256, (64,), 128), (2, 32, 128, (128,), 256), (96, 64, 256, (64,), 64), (2, 64, 256, (64, 64, 64), 64),
                            (2, 32, 128, (128, 128, 128), 128), (96, 16, 64, (64,), 64), (2, 64, 256, (192,), 64), (96, 64, 256, (192,), 64)]:
    xs = [u("h%d" % i, (B, H, W, c)) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = u("hw", (cout, cin, 3, 3)) * (1.0 / (cin * 9) ** 0.5)
    wph, _ = packing.pack_conv(wt, None)
    run = lambda: ops.conv2d(xs, wph, None, cout, 3, pad=1, pro_act="relu")
    mode(0); r0 = run(); mode(2); r2 = run(); r2b = run()
    print("halo B %3d %3dx%3d segs %-16s cout %3d: x3 vs f32 max %.3e  rerun equal %s" % (B, H, W, segs, cout, float((r0 - r2).abs().max()), torch.equal(r2, r2b)))
mode(0)

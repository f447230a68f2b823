"""Debug: small-shape linears (k-loop x3) on two streams at once vs alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops, _abi
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
u = lambda n, s, lo=-1.0, hi=1.0: synth.uniform(n, s, lo, hi, 5).to(dev)
_abi.check(_abi.lib.dpmn_set_compute_dtype(int(os.environ.get("DBG_MODE", "2"))))
shapes = [(156, 64, 256), (1248, 64, 1024), (6144, 96, 384), (384, 192, 2048), (96, 256, 512), (6144, 64, 256), (156, 2048, 512)]
data = []
for i, (M, N, K) in enumerate(shapes):
    for s_ in range(2):
        data.append((u("x%d_%d" % (i, s_), (M, K)), u("w%d" % i, (N, K), -0.1, 0.1), u("b%d" % i, (N,)), u("r%d_%d" % (i, s_), (M, N))))
alone = [ops.linear(x, w, b, res1=r).clone() for x, w, b, r in data]
torch.cuda.synchronize()
st = [torch.cuda.Stream() for _ in range(2)]
bad = [0] * len(shapes)
for rep in range(30):
    outs = [None] * len(data)
    for i in range(len(shapes)):
        for s_ in range(2):
            with torch.cuda.stream(st[s_]):
                x, w, b, r = data[2 * i + s_]
                outs[2 * i + s_] = ops.linear(x, w, b, res1=r)
    torch.cuda.synchronize()
    for j, (o, a) in enumerate(zip(outs, alone)):
        if not torch.equal(o, a):
            bad[j // 2] += 1
print("mismatches per shape over 60 launches:", list(zip(shapes, bad)))

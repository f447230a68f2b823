"""Debug: x3 k-loop linears (stream A) next to fp32 kernels of other families (stream B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops, _abi
from dpmn_amd.model import packing
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
u = lambda n, s, lo=-1.0, hi=1.0: synth.uniform(n, s, lo, hi, 5).to(dev)
mode = lambda m: _abi.check(_abi.lib.dpmn_set_compute_dtype(m))
shapes = [(156, 64, 256), (1248, 64, 1024), (6144, 96, 384), (156, 2048, 512)]
data = [(u("x%d" % i, (M, K)), u("w%d" % i, (N, K), -0.1, 0.1), u("b%d" % i, (N,)), u("r%d" % i, (M, N))) for i, (M, N, K) in enumerate(shapes)]
mode(2)
alone = [ops.linear(x, w, b, res1=r).clone() for x, w, b, r in data]
# stream B work, fp32: pointwise GEMM, whole-K linear, conv, layernorm-ish
g, wp_, bp_ = u("g", (6, 1024, 384)), u("wpw", (384, 384), -0.1, 0.1), u("bpw", (384,))
x96, w96 = u("x96", (6144, 96)), u("w96", (96, 96), -0.1, 0.1)
xc = u("xc", (6, 16, 64, 64)); wc, _ = packing.pack_conv(u("wc", (64, 64, 3, 3), -0.05, 0.05), None)
def other():
    ops.pointwise(g, wp_, bp_); ops.linear(x96, w96); ops.conv2d([xc], wc, None, 64, 3, pad=1)
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for bmode in (0, 2):
    bad = [0] * len(shapes)
    for rep in range(40):
        outs = []
        for i, (x, w, b, r) in enumerate(data):
            mode(bmode)
            with torch.cuda.stream(sB):
                other(); other()
            mode(2)
            with torch.cuda.stream(sA):
                outs.append(ops.linear(x, w, b, res1=r))
        torch.cuda.synchronize()
        for j, (o, a) in enumerate(zip(outs, alone)):
            bad[j] += int(not torch.equal(o, a))
    print("stream B in mode %d: mismatches of the x3 linears over 40 reps:" % bmode, list(zip(shapes, bad)))
mode(0)

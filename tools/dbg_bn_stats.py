"""How accurate are the train-mode BatchNorm statistics of the CMM forward (conv-epilogue sum / sum-of-squares)?  Per BatchNorm
layer: rstd / mean from the library vs a float64 two-pass computation over the library's own raw conv output.
usage: python tools/dbg_bn_stats.py [cnum] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd.model.cmm import ComplementationModulationModule
from dpmn_amd.train import cmm_train
from dpmn_amd.utils import synth

cnum = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
m = ComplementationModulationModule(cnum=cnum)
sd = m.state_dict()
synth.synth_fill_(sd, 31)
m.load_state_dict(sd)
m = m.to(dev).train()
x1 = synth.uniform("cmm_x1", (B, 3, 32, 128), 0, 1, 7).to(dev)
x2 = synth.uniform("cmm_x2", (B, 3, 32, 128), 0, 1, 7).to(dev)
with torch.no_grad():
    out, graph = cmm_train.build(m, x1, x2)
torch.cuda.synchronize()
for i, u in enumerate(graph["units"]):
    if u.bn is None or u.out.mean is None:
        continue
    r = u.out.r.double().reshape(-1, u.out.r.shape[-1])
    mean = r.mean(0)
    var = ((r - mean) ** 2).mean(0)
    rstd = 1.0 / torch.sqrt(var + u.bn.eps)
    e_r = ((u.out.rstd.double() - rstd).abs() / rstd)
    e_m = ((u.out.mean.double() - mean).abs() / (var.sqrt() + 1e-30))
    ratio = (mean ** 2 / (var + 1e-30))
    print("unit %2d %-10s px %6d C %4d | rstd rel err max %.2e median %.2e | mean err / std max %.2e | mean^2/var max %.1e median %.1e" % (
        i, u.kind, r.shape[0], r.shape[1], float(e_r.max()), float(e_r.median()), float(e_m.max()), float(ratio.max()), float(ratio.median())))

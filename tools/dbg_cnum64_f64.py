"""Per-tensor table of the cnum-64 CMM gradient fixture against float64 (tests/golden/grads_cmm_cnum64_f64.npz): our error, the
reference's own fp32 error, ratio -- in network order.  usage: python tools/dbg_cnum64_f64.py   (DPMN_COMPUTE_DTYPE=x3 for mode 2)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from dpmn_amd.model.cmm import ComplementationModulationModule
from dpmn_amd.utils import synth
from helpers import load_golden, fixture_grad_names, grad_error_vs_fixture

dev = torch.device("cuda:0")
B = 2
m = ComplementationModulationModule(cnum=64)
sd = m.state_dict()
synth.synth_fill_(sd, 31)
m.load_state_dict(sd)
m = m.to(dev).train()
x1 = synth.uniform("cmm_x1", (B, 3, 32, 128), 0, 1, 7).to(dev).requires_grad_(True)
x2 = synth.uniform("cmm_x2", (B, 3, 32, 128), 0, 1, 7).to(dev).requires_grad_(True)
cot = synth.uniform("cmm_cot", (B, 3, 32, 128), -1, 1, 7).to(dev)
out = m(x1, x2)
(out * cot).sum().backward()
named = {"x1": x1.grad, "x2": x2.grad}
named.update({n: p.grad for n, p in m.named_parameters()})
z = load_golden("grads_cmm_cnum64_f64")
g32 = load_golden("grads_cmm_cnum64")
print("forward max|err| vs the reference's fp32 output: %.3e" % float((out.detach().cpu() - torch.from_numpy(g32["out"])).abs().max()))
rows = []
for n in fixture_grad_names(z):
    err, amax = grad_error_vs_fixture(z, n, named[n])
    if amax < 1e-9:
        continue
    ref = float(z[n + "::ref32_err"])
    rows.append((n, err, ref, amax, tuple(named[n].shape)))
for n, err, ref, amax, shp in rows:
    print("%-34s %-22s |f64|max %.2e  ours %.2e  ref32 %.2e  ratio %7.2f" % (n, shp, amax, err, ref, err / max(ref, 1e-9)))

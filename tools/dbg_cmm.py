import sys; sys.path.insert(0,'/root/repo')
import torch
from dpmn_amd.utils import synth
from dpmn_amd.model.cmm import ComplementationModulationModule
from dpmn_amd.train import cmm_train
dev=torch.device('cuda:0')
def u(name, shape, lo=-1.0, hi=1.0, seed=80): return synth.uniform(name, shape, lo, hi, seed)
B=4; cnum=8
m = ComplementationModulationModule(cnum=cnum)
sd = m.state_dict(); synth.synth_fill_(sd, 95); m.load_state_dict(sd)
x1, x2 = u("x1", (B, 3, 32, 128), 0, 1).to(dev), u("x2", (B, 3, 32, 128), 0, 1).to(dev)
cot = u("cot", (B, 3, 32, 128), -1, 1).to(dev)
m = m.to(dev).train()
runs=[]
with torch.no_grad():
    for it in range(4):
        out, graph = cmm_train.build(m, x1, x2)
        dxs, gr, _ = cmm_train.backward(m, graph, cot.clone())
        torch.cuda.synchronize()
        runs.append(([u_.out.G.clone() if (u_.out.G is not None and u_.out.G is not False) else None for u_ in graph['units']], [u_.out.r.clone() if u_.out.r is not None else None for u_ in graph['units']], dxs, graph))
names=[(u_.kind, tuple(u_.out.r.shape) if u_.out.r is not None else None) for u_ in runs[0][3]['units']]
for it in range(1,4):
    print("run", it)
    for i in reversed(range(len(names))):
        g0, g1 = runs[0][0][i], runs[it][0][i]
        r0, r1 = runs[0][1][i], runs[it][1][i]
        dr = float((r0-r1).abs().max()) if r0 is not None else -1
        dg = float((g0-g1).abs().max()/ (g0.abs().max()+1e-12)) if g0 is not None else -1
        if dg > 1e-4 or dr > 1e-5: print("  unit", i, names[i], "fwd diff %.2e  G reldiff %.2e"%(dr, dg))

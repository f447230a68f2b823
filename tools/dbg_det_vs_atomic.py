"""PGRM backward: atomics-free (DET_SMALL) vs atomic variants, per-parameter relative difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd.utils import synth
from dpmn_amd.model.pgrm import PGRM
from dpmn_amd.train import pgrm_train
dev = torch.device("cuda:0")
n = 6
args = dict(patch_size=[2] * n, embed_dim=[96] * n, depths=[1] * n, num_heads=[[6]] * n, window_size=[[2, 4, 8]] * n,
            mlp_ratio=[4.] * n, drop_rate=[0.] * n, attn_drop_rate=[0.] * n, drop_path_rate=[0.] * n)
B = 3
res = {}
for det in (False, True):
    pgrm_train.DET_SMALL = det
    m = PGRM(iter=0, mode=False, hidden_size=3, **args)
    sd = m.state_dict(); synth.synth_fill_(sd, 11); m.load_state_dict(sd)
    m = m.to(dev).train()
    u = lambda name, shape, lo, hi: synth.uniform(name, shape, lo, hi, 5).to(dev)
    x_q = torch.floor(u("xq", (B, 2, 32, 128), 0, 256)); x_kv = u("xkv", (B, 3, 32, 128), 0, 1).requires_grad_(True)
    out = m(x_q, x_kv, [])
    (out * u("cot", (B, 3, 32, 128), -1, 1)).sum().backward()
    torch.cuda.synchronize()
    res[det] = {k: p.grad.clone() for k, p in m.named_parameters()}
    res[det]["x_kv"] = x_kv.grad.clone()
for k in res[True]:
    a, b = res[False][k], res[True][k]
    e = float((a - b).abs().max() / (a.abs().max() + 1e-30))
    if e > 1e-5:
        print("%-60s rel diff %.3e  |atomic| %.3e |det| %.3e" % (k, e, float(a.abs().max()), float(b.abs().max())))
print("done")

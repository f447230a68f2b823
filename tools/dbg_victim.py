"""Debug: which fp32 op of the TATT trunk changes its result next to an x3 implicit-GEMM conv on another stream?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops, workload, _abi
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
_abi.check(_abi.lib.dpmn_set_compute_dtype(2))
sr, models, psn, inp = workload.build("cfg1", batch=6)
P = psn._trunk_pack()
b = synth.synth_batch(6, seed=61)
lr, lv = b["images_lr"].to(dev), b["label_vecs"].to(dev)
b1 = psn._head(lr, P)
tp, _ = psn._tp_interpreter(b1, lv, P)
B, H, W, ch = b1.shape
hid = ch // 2
c1 = lambda x: ops.conv2d([x], *P["srb0.c1"], ch, 3, pad=1, epi_act="mish")
c2 = lambda x: ops.conv2d([x], *P["srb0.c2"], ch, 3, pad=1)
gw, gb, whh, bhh = P["srb0.g1"]
gw2, gb2, whh2, bhh2 = P["srb0.g2"]
r1 = c1(b1); r2 = c2(r1)
gi = ops.cat2_linear(r2.reshape(-1, ch), tp.reshape(-1, tp.shape[-1]), gw, gb)
s = ops.bigru(gi.reshape(B, H, W, -1), whh, bhh, B, H, W, "h", res=b1, hidden=hid)
gi2 = ops.linear(s.reshape(-1, ch), gw2, gb2)
f = ops.bigru(gi2.reshape(B, H, W, -1), whh2, bhh2, B, H, W, "w", hidden=hid)
t7 = ops.conv2d([f], *P["b7"], ch, 3, pad=1, res=b1)
victims = {
    "halo conv + mish": lambda: c1(b1), "halo conv": lambda: c2(r1), "cat2_linear": lambda: ops.cat2_linear(r2.reshape(-1, ch), tp.reshape(-1, tp.shape[-1]), gw, gb),
    "bigru h": lambda: ops.bigru(gi.reshape(B, H, W, -1), whh, bhh, B, H, W, "h", res=b1, hidden=hid),
    "linear K64": lambda: ops.linear(s.reshape(-1, ch), gw2, gb2), "bigru w": lambda: ops.bigru(gi2.reshape(B, H, W, -1), whh2, bhh2, B, H, W, "w", hidden=hid),
    "b7 conv + res": lambda: ops.conv2d([f], *P["b7"], ch, 3, pad=1, res=b1),
    "up conv x3 (pixel shuffle)": lambda: ops.conv2d([t7], *P["up"], 4 * ch, 3, pad=1, epi_act="mish", pixel_shuffle=True),
}
alone = {k: v().clone() for k, v in victims.items()}
u = alone["up conv x3 (pixel shuffle)"]
victims["last 9x9 conv"] = lambda: ops.conv2d([u], *P["last"], psn.in_planes, 9, pad=4, epi_act="tanh", out_nchw=True)
alone["last 9x9 conv"] = victims["last 9x9 conv"]().clone()
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
t7b = t7.clone()
for name, fn in victims.items():
    bad, info = 0, ""
    for rep in range(30):
        with torch.cuda.stream(sB):
            for _ in range(6):
                ops.conv2d([t7b], *P["up"], 4 * ch, 3, pad=1, epi_act="mish", pixel_shuffle=True)       # the aggressor: x3 implicit GEMM + split-K reduce
        with torch.cuda.stream(sA):
            outs = [fn() for _ in range(4)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, alone[name]):
                bad += 1
                if bad == 1:
                    d = (o - alone[name]).abs()
                    info = "first: %d of %d elements differ, max %.2e (|ref| max %.2e)" % (int((d > 0).sum()), d.numel(), float(d.max()), float(alone[name].abs().max()))
    print("%-28s mismatches %3d / 120  %s" % (name, bad, info if bad else ""))

"""Per-call timing of every ops.conv2d / lib.dpmn_conv2d_nhwc_f32 launch in one eager cfg1 TRAINING step (forward convs and the
data-gradient convs): geometry, time.   usage: python tools/prof_train_convs.py [max_cout]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import _abi, workload
from dpmn_amd.loss.image_loss import ImageLoss
from dpmn_amd.model.distill_module import DistillModule
from dpmn_amd.train.optim import Trainer

max_cout = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
sr, models, psn, inp = workload.build("cfg1")
arch, b1, b2, _ = workload.CONFIGS["cfg1"]
distill = [DistillModule().to(sr.device) for _ in range(b1 + b2 - 2)]
crit = ImageLoss(gradient=True, loss_weight=[1, 1])
for m in models + distill:
    m.train()
    for p in m.parameters():
        p.requires_grad = True
trainer = Trainer(models + distill, lr=1e-3, beta1=0.5, max_norm=0.25)
step = lambda: sr.train_step(models, psn, distill, crit, trainer, inp["images_lr"], inp["images_hr"], inp.get("label_vecs"),
                             text_priors=inp["text_priors"])
for _ in range(2):
    step()
torch.cuda.synchronize()
rec = []
orig = _abi.lib.dpmn_conv2d_nhwc_f32


def timed(dref, st):
    d = dref._obj
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = orig(dref, st)
    e.record()
    cin = sum(d.cseg[i] for i in range(3) if d.inp[i])
    rec.append((s, e, d.B, d.Hin, d.Win, cin, d.Cout, d.KH, d.KW, d.stride, d.dil_y, d.Hp, d.Wp, d.pro_act, bool(d.in_scale[0]), bool(d.stats),
                d.nphase))
    return rc


import dpmn_amd.ops as ops
import dpmn_amd.train.cmm_train as ct
import dpmn_amd.train.pgrm_train as pt


class LibProxy:
    def __getattr__(self, name):
        return timed if name == "dpmn_conv2d_nhwc_f32" else getattr(_abi.lib, name)


for mod in (ops, ct, pt):
    if hasattr(mod, "lib"):
        mod.lib = LibProxy()
step()
torch.cuda.synchronize()
tot = 0.0
for s, e, B, H, W, cin, cout, kh, kw, stride, dil, hp, wp, act, aff, stats, nph in rec:
    if cout > max_cout:
        continue
    ms = s.elapsed_time(e)
    tot += ms
    fl = 2.0 * B * hp * wp * cin * kh * kw * cout * max(nph, 1)
    print("in %3dx%3dx%4d cout %4d k %dx%d s%d d%2d M %6d act %d aff %d stats %d ph %d %7.1f us %6.1f TF" % (
        H, W, cin, cout, kh, kw, stride, dil, B * hp * wp, act, aff, stats, nph, ms * 1e3, fl / ms / 1e9))
print("total %.2f ms in %d calls" % (tot, len(rec)))

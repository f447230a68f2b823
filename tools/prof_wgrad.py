"""Per-call timing of every conv weight-gradient launch (dpmn_conv2d_wgrad_f32 + its unpack) and every data-gradient /
forward conv in one eager cfg1 training step: geometry, time, TFLOP/s.   usage: python tools/prof_wgrad.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload
from dpmn_amd.loss.image_loss import ImageLoss
from dpmn_amd.model.distill_module import DistillModule
from dpmn_amd.train import pgrm_train
from dpmn_amd.train.optim import Trainer

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
orig = pgrm_train.conv_wgrad_into


def timed(d, dy, dweight, layout="conv", phase=None):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    orig(d, dy, dweight, layout, phase)
    e.record()
    cin = sum(d.cseg[i] for i in range(3) if d.inp[i])
    rec.append((s, e, d.B, d.Hin, d.Win, cin, d.Cout, d.KH, d.KW, d.stride, d.Hp, d.Wp, layout))


pgrm_train.conv_wgrad_into = timed
import dpmn_amd.train.cmm_train as cmm_train
cmm_train.conv_wgrad_into = timed
step()
torch.cuda.synchronize()
tot = 0.0
for s, e, B, H, W, cin, cout, kh, kw, stride, hp, wp, layout in rec:
    ms = s.elapsed_time(e)
    tot += ms
    fl = 2.0 * B * hp * wp * cin * kh * kw * cout
    print("in %3dx%3dx%4d cout %4d k %dx%d s%d M %6d K %5d %-10s %7.1f us %6.1f TF" % (H, W, cin, cout, kh, kw, stride, B * hp * wp, cin * kh * kw,
                                                                                     layout, ms * 1e3, fl / ms / 1e9))
print("wgrad total %.2f ms in %d calls" % (tot, len(rec)))

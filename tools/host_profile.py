#!/usr/bin/env python3
"""cProfile of the host thread over training steps (forward issue; the backward's issue runs in autograd's thread and shows up as
time inside loss.backward()).  usage: python tools/host_profile.py"""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload
from dpmn_amd.loss.image_loss import ImageLoss
from dpmn_amd.model.distill_module import DistillModule
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
step = lambda: sr.train_step(models, psn, distill, crit, trainer, inp["images_lr"], inp["images_hr"], inp.get("label_vecs"), text_priors=inp["text_priors"])
for _ in range(4): step()
torch.cuda.synchronize()
# the backward runs in autograd's device thread: profile it too by running backward nodes in this thread is not possible; instead
# profile all threads with the threading hook
import threading
prof = cProfile.Profile()
threading.setprofile(lambda *a: None)
prof.enable()
for _ in range(5): step()
prof.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(l[:150] for l in s.getvalue().split("\n")[:45]))

#!/usr/bin/env python3
"""Where the training step's wall time goes with all streams live: HIP events around every module forward / backward and the
optimizer (no profiler: rocprofv3's kernel trace serialises the streams).  Prints start / end in ms since the step began.
usage: python tools/phase_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload
from dpmn_amd.loss.image_loss import ImageLoss
from dpmn_amd.model.distill_module import DistillModule
from dpmn_amd.train.optim import Trainer
from dpmn_amd.train import pgrm_train, cmm_train
from dpmn_amd.model import distill_module

sr, models, psn, inp = workload.build("cfg1")
arch, b1, b2, _ = workload.CONFIGS["cfg1"]
distill = [DistillModule().to(sr.device) for _ in range(b1 + b2 - 2)]
crit = ImageLoss(gradient=True, loss_weight=[1, 1])
for m in models + distill:
    m.train()
    for p in m.parameters():
        p.requires_grad = True
trainer = Trainer(models + distill, lr=1e-3, beta1=0.5, max_norm=0.25)
import time
REC = []
ON = [False]


def wrap(fn, name):
    def inner(*a, **k):
        if not ON[0]:
            return fn(*a, **k)
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        h0 = time.perf_counter()
        out = fn(*a, **k)
        h1 = time.perf_counter()
        e1.record(st)
        REC.append((name, st.cuda_stream, e0, e1, h0, h1))
        return out
    return inner


for i, m in enumerate(models):
    m.forward = wrap(m.forward, ("PGRM%d" % i if i < b1 + b2 else "CMM") + ".fwd")
for i, d in enumerate(distill):
    d.forward = wrap(d.forward, "distill%d.fwd" % i)
psn.forward = wrap(psn.forward, "PSN.fwd")
pgrm_train.PGRMFunction.backward = staticmethod(wrap(pgrm_train.PGRMFunction.backward, "PGRM.bwd"))
cmm_train.CMMFunction.backward = staticmethod(wrap(cmm_train.CMMFunction.backward, "CMM.bwd"))
distill_module._DistillFn.backward = staticmethod(wrap(distill_module._DistillFn.backward, "distill.bwd"))
trainer.step = wrap(trainer.step, "optimizer")
sr.prefetch_psn = wrap(sr.prefetch_psn, "PSN.prefetch(issue)")

step = lambda h: sr.train_step(models, psn, distill, crit, trainer, inp["images_lr"], inp["images_hr"], inp.get("label_vecs"),
                               text_priors=inp["text_priors"], psn_out=h, prefetch=(inp["images_lr"], inp.get("label_vecs")))
h = None
for _ in range(5):
    step(h); h = sr.psn_prefetched
torch.cuda.synchronize()
ON[0] = True
import time
t0 = torch.cuda.Event(enable_timing=True); t0.record()
c0 = time.perf_counter()
step(h)
c1 = time.perf_counter()
t1 = torch.cuda.Event(enable_timing=True); t1.record()
torch.cuda.synchronize()
print("step %.2f ms on the GPU; the host thread needed %.2f ms to issue it" % (t0.elapsed_time(t1), (c1 - c0) * 1e3))
streams = {}
# GPU window of every phase next to the HOST window in which its launches were issued (same origin: the step's start).  A phase whose
# GPU end trails its host end by less than a launch or two is issue-bound: the GPU ran out of queued work while the host was still in it.
for name, sid, e0, e1, h0, h1 in sorted(REC, key=lambda r: t0.elapsed_time(r[2])):
    s_ = streams.setdefault(sid, len(streams))
    print("  stream %d  %-22s GPU %7.2f .. %7.2f  (%5.2f ms)   host issue %7.2f .. %7.2f  (%5.2f ms)   GPU end - host end %6.2f" % (
        s_, name, t0.elapsed_time(e0), t0.elapsed_time(e1), e0.elapsed_time(e1), (h0 - c0) * 1e3, (h1 - c0) * 1e3, (h1 - h0) * 1e3,
        t0.elapsed_time(e1) - (h1 - c0) * 1e3))

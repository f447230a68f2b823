"""Which Python lines launch torch's own (non-dpmn) kernels in one step?  Groups aten ops that reach the GPU by source line.
usage: python tools/prof_torch_ops.py [train|fwd]"""
import os
import sys
import collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
from dpmn_amd import workload
sr, models, psn, inp = workload.build("cfg1")
arch, b1, b2, _ = workload.CONFIGS["cfg1"]
if mode == "train":
    from dpmn_amd.loss.image_loss import ImageLoss
    from dpmn_amd.model.distill_module import DistillModule
    from dpmn_amd.train.optim import Trainer
    distill = [DistillModule().to(sr.device) for _ in range(b1 + b2 - 2)]
    crit = ImageLoss(gradient=True, loss_weight=[1, 1])
    for m in models + distill:
        m.train()
        for p in m.parameters():
            p.requires_grad = True
    trainer = Trainer(models + distill, lr=1e-3, beta1=0.5, max_norm=0.25)
    step = lambda: sr.train_step(models, psn, distill, crit, trainer, inp["images_lr"], inp["images_hr"], inp.get("label_vecs"),
                                 text_priors=inp["text_priors"])
else:
    step = lambda: sr.refine(models, psn, inp["images_lr"], inp.get("label_vecs"), text_priors=inp["text_priors"])
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
by = collections.Counter()
tm = collections.Counter()
for ev in prof.events():
    # leaf attribution: the op that launched the kernel / memcpy itself (aten::copy_ under aten::to, ...)
    if getattr(ev, "self_device_time_total", 0) <= 0 or not ev.name.startswith("aten::"):
        continue
    st = [s for s in (ev.stack or []) if ("dpmn_amd" in s or "bench" in s) and "prof_torch_ops" not in s]
    if not st:          # ops issued from the autograd thread carry no Python frames of ours: name the enclosing backward node
        par = ev.cpu_parent
        while par is not None and not ("Backward" in par.name or "autograd" in par.name):
            par = par.cpu_parent
        st = ["<" + par.name + ">"] if par is not None else [str((ev.stack or ["?"])[:1])]
    key = (ev.name, st[0].strip()[-90:] if st else "?")
    by[key] += 1
    tm[key] += ev.self_device_time_total
for key, us in tm.most_common(45):
    print("%7.1f us %4d  %-22s %s" % (us, by[key], key[0], key[1]))

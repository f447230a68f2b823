"""Who issues the device-to-device memcpys (rocprof: __amd_rocclr_copyBuffer) of one training step?  Counts the runtime-level
hipMemcpyAsync calls per enclosing aten op / autograd node / Python line.  usage: python tools/prof_memcpy.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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
if len(sys.argv) > 1 and sys.argv[1] == "fwd":        # the eval forward instead of the training step
    for m in models:
        m.eval()
    _fwd = lambda: sr.refine(models, psn, inp["images_lr"], inp.get("label_vecs"), text_priors=inp["text_priors"])
    def step():
        with torch.no_grad():
            return _fwd()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
names = collections.Counter()
who = collections.Counter()
for ev in prof.events():
    n = ev.name
    if "emcpy" in n or "emset" in n or "copyBuffer" in n or "fillBuffer" in n:
        names[(n, str(ev.device_type))] += 1
        if "DeviceType.CPU" in str(ev.device_type):
            par, chain = ev.cpu_parent, []
            while par is not None and len(chain) < 4:
                chain.append(par.name)
                par = par.cpu_parent
            st = [s for s in (ev.stack or []) if "dpmn_amd" in s]
            who[(n, " < ".join(chain), st[0].strip()[-70:] if st else "")] += 1
for k, v in names.most_common(20):
    print("%5d  %s" % (v, k))
print()
for k, v in who.most_common(40):
    print("%5d  %s" % (v, k))

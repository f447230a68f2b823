import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dpmn_amd import workload
from dpmn_amd.model import packing
from dpmn_amd.loss.image_loss import ImageLoss
from dpmn_amd.model.distill_module import DistillModule
from dpmn_amd.train.optim import Trainer
sr, models, psn, inp = workload.build("cfg1", batch=4)
distill = [DistillModule().to(sr.device) for _ in range(4)]
crit = ImageLoss(gradient=True, loss_weight=[1, 1])
for m in models + distill:
    m.train()
    for p in m.parameters(): p.requires_grad = True
tr = Trainer(models + distill)
seen = set()
for step in range(3):
    sr.train_step(models, psn, distill, crit, tr, inp["images_lr"], inp["images_hr"], inp.get("label_vecs"), text_priors=inp["text_priors"])
    new = [k for k in tr.pack_cache.order if k not in seen]
    print("step", step, "registered", len(new), "total", len(tr.pack_cache.order))
    if step > 0:
        for k in new[:12]: print("   ", k)
    seen |= set(new)

"""Python call sites of torch's own device ops (fill_/copy_/cat/add ...) in one training step, forward thread only (the autograd
thread is attributed per backward node by tools/prof_torch_ops.py).  usage: python tools/trace_torch_ops.py [train|fwd]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from dpmn_amd import workload
from dpmn_amd.loss.image_loss import ImageLoss
from dpmn_amd.model.distill_module import DistillModule
from dpmn_amd.train.optim import Trainer

sr, models, psn, inp = workload.build("cfg1")
arch, b1, b2, _ = workload.CONFIGS["cfg1"]
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
if mode == "train":
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
SKIP = ("view", "reshape", "permute", "transpose", "slice", "select", "expand", "detach", "alias", "as_strided", "unsqueeze", "squeeze",
        "t.default", "empty", "_unsafe_view", "split", "unbind", "size", "stride", "is_", "record_stream", "_local_scalar")
count = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            fr = [f for f in traceback.extract_stack() if "dpmn_amd" in f.filename]
            where = "%s:%d" % (os.path.relpath(fr[-1].filename), fr[-1].lineno) if fr else "?"
            count[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    step()
torch.cuda.synchronize()
for (name, where), n in count.most_common(70):
    print("%4d  %-28s %s" % (n, name, where))

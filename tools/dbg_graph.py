"""Run-to-run spread of the eager training step vs the hipGraph replay (debug helper)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload
from dpmn_amd.interfaces.super_resolution import TextSR
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
B, b1, b2 = 2, 2, 2


def fresh():
    sr_ = TextSR(workload.make_config(B), workload.make_args("tsrn", b1, b2, B))
    models, psn, distill, crit, trainer = sr_.build_training()
    for i, m in enumerate([psn] + models + distill):
        sd = m.state_dict()
        synth.synth_fill_(sd, 500 + i)
        with torch.no_grad():
            for k, v in m.state_dict().items():
                v.copy_(sd[k])
    psn.eval()
    return sr_, models, psn, distill, crit, trainer


batches = [synth.synth_batch(B, seed=20 + i) for i in range(4)]
priors = [[torch.floor(synth.uniform("gtp%d_%d" % (i, k), (B, 2, 32, 128), 0, 256, 4)).to(dev) for k in range(b1)] for i in range(4)]
lr0, hr0 = batches[0]["images_lr"].to(dev), batches[0]["images_hr"].to(dev)
for mode in ("eager", "eager", "graph", "graph"):
    sr_, models, psn, distill, crit, trainer = fresh()
    if mode == "eager":
        for _ in range(2):
            sr_.train_step(models, psn, distill, crit, trainer, lr0, hr0, None, text_priors=priors[0])
        ls = [float(sr_.train_step(models, psn, distill, crit, trainer, b["images_lr"].to(dev), b["images_hr"].to(dev), None,
                                   text_priors=priors[i])) for i, b in enumerate(batches)]
    else:
        run = sr_.graphed_train_step(models, psn, distill, crit, trainer, lr0, hr0, None, priors[0], warmup=2)
        ls = [float(run(b["images_lr"].to(dev), b["images_hr"].to(dev), None, priors[i])) for i, b in enumerate(batches)]
    print(mode, ["%.3f" % x for x in ls])

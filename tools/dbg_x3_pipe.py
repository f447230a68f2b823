"""Debug: cfg1 at B = 6 in mode 2 -- are sequential refine() calls bitwise repeatable, and which stage first differs under two lanes?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload, _abi
from dpmn_amd.utils import synth
from dpmn_amd.interfaces.super_resolution import RefinePipeline
dev = torch.device("cuda:0")
_abi.check(_abi.lib.dpmn_set_compute_dtype(int(os.environ.get("DBG_MODE", "2"))))
sr, models, psn, inp = workload.build("cfg1", batch=6)
batches = []
for i in range(5):
    b = synth.synth_batch(6, seed=60 + i)
    pri = [torch.floor(synth.uniform("pp%d_%d" % (i, k), (6, 2, 32, 128), 0.0, 256.0, 3)).to(dev) for k in range(3)]
    batches.append((b["images_lr"].to(dev), b["label_vecs"].to(dev), pri))
seq = [sr.refine(models, psn, lr, lv, text_priors=pri).clone() for lr, lv, pri in batches]
for rep in range(3):
    again = [sr.refine(models, psn, lr, lv, text_priors=pri).clone() for lr, lv, pri in batches]
    torch.cuda.synchronize()
    print("sequential pass %d equal:" % rep, [bool(torch.equal(a, b)) for a, b in zip(again, seq)], ["%.2e" % float((a - b).abs().max()) for a, b in zip(again, seq)])
# with a device synchronize between calls
again = []
for lr, lv, pri in batches:
    again.append(sr.refine(models, psn, lr, lv, text_priors=pri).clone()); torch.cuda.synchronize()
print("sequential + sync equal:", [bool(torch.equal(a, b)) for a, b in zip(again, seq)])
pipe = RefinePipeline(sr, models, psn, depth=2)
for rep in range(int(os.environ.get("DBG_REPS", "3"))):
    outs = [pipe.submit(lr, lv, text_priors=pri) for lr, lv, pri in batches]
    pipe.synchronize()
    print("two lanes pass %d equal:" % rep, [bool(torch.equal(a, b)) for a, b in zip(outs, seq)], ["%.2e" % float((a - b).abs().max()) for a, b in zip(outs, seq)])

"""Debug: two refine() calls on two streams at once (mode 2), every intermediate against the sequential run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload, _abi
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
_abi.check(_abi.lib.dpmn_set_compute_dtype(int(os.environ.get("DBG_MODE", "2"))))
sr, models, psn, inp = workload.build("cfg1", batch=6)
batches = []
for i in range(4):
    b = synth.synth_batch(6, seed=60 + i)
    pri = [torch.floor(synth.uniform("pp%d_%d" % (i, k), (6, 2, 32, 128), 0.0, 256.0, 3)).to(dev) for k in range(3)]
    batches.append((b["images_lr"].to(dev), b["label_vecs"].to(dev), pri))
def flat(mid, out):
    d = {"psn": mid["psn"], "cmm": mid["cmm"], "out": out}
    for k, t in enumerate(mid["branch1"]): d["b1_%d" % k] = t
    for k, t in enumerate(mid["branch2"]): d["b2_%d" % k] = t
    return {k: v.clone() for k, v in d.items()}
seq = []
for lr, lv, pri in batches:
    out, mid = sr.refine(models, psn, lr, lv, text_priors=pri, return_all=True)
    seq.append(flat(mid, out))
torch.cuda.synchronize()
lanes = [torch.cuda.Stream(dev) for _ in range(2)]
for rep in range(int(os.environ.get("DBG_REPS", "4"))):
    got = []
    for i, (lr, lv, pri) in enumerate(batches):
        with torch.cuda.stream(lanes[i % 2]):
            out, mid = sr.refine(models, psn, lr, lv, text_priors=pri, return_all=True)
            got.append(flat(mid, out))
    torch.cuda.synchronize()
    for i in range(len(batches)):
        bad = [(k, "%.1e" % float((got[i][k] - seq[i][k]).abs().max())) for k in seq[i] if not torch.equal(got[i][k], seq[i][k])]
        if bad:
            print("pass %d batch %d first mismatches:" % (rep, i), bad[:6])
print("done")

"""Debug: the frozen TATT PSN alone on two streams at once (mode 2) vs sequentially."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload, _abi
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
_abi.check(_abi.lib.dpmn_set_compute_dtype(int(os.environ.get("DBG_MODE", "2"))))
sr, models, psn, inp = workload.build("cfg1", batch=6)
bs = []
for i in range(4):
    b = synth.synth_batch(6, seed=60 + i)
    bs.append((b["images_lr"].to(dev), b["label_vecs"].to(dev)))
seq = [psn(lr, lv)[0].clone() for lr, lv in bs]
torch.cuda.synchronize()
lanes = [torch.cuda.Stream(dev) for _ in range(2)]
if os.environ.get("DBG_PAD"):       # first use of each lane with a large live allocation in between: the lanes' workspaces are not neighbours
    keep = []
    for i in range(2):
        with torch.cuda.stream(lanes[i]):
            keep.append(torch.full((64 << 20,), float("nan"), device=dev))
            psn(*bs[i])
            keep.append(torch.full((64 << 20,), float("nan"), device=dev))
    torch.cuda.synchronize()
bad = 0
for rep in range(int(os.environ.get("DBG_REPS", "10"))):
    got = []
    for i, (lr, lv) in enumerate(bs):
        with torch.cuda.stream(lanes[i % 2]):
            got.append(psn(lr, lv)[0])
    torch.cuda.synchronize()
    for i in range(4):
        if not torch.equal(got[i], seq[i]):
            bad += 1
            if bad <= 6:
                d = (got[i] - seq[i]).abs()
                nz = torch.nonzero(d > 0)
                print("rep %d batch %d psn differs by %.2e: %d of %d elements, images %s, rows %d..%d, cols %d..%d" % (
                    rep, i, float(d.max()), nz.shape[0], d.numel(), sorted(set(nz[:, 0].tolist())), int(nz[:, 2].min()), int(nz[:, 2].max()),
                    int(nz[:, 3].min()), int(nz[:, 3].max())))
print("PSN-only two-lane mismatches: %d" % bad)

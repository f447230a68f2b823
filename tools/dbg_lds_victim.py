"""Debug: does any kernel of the cfg1 forward depend on LDS words it did not write?  The forward alone vs next to an LDS-poison
kernel on another stream (fp32 mode and mode 2), stage by stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload, _abi
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
_abi.check(_abi.lib.dpmn_set_compute_dtype(int(os.environ.get("DBG_MODE", "0"))))
B = int(os.environ.get("DBG_B", "6"))
sr, models, psn, inp = workload.build("cfg1", batch=B)
def flat(mid, out):
    d = {"psn": mid["psn"], "cmm": mid["cmm"], "out": out}
    for k, t in enumerate(mid["branch1"]): d["b1_%d" % k] = t
    for k, t in enumerate(mid["branch2"]): d["b2_%d" % k] = t
    return {k: v.clone() for k, v in d.items()}
run = lambda: flat(*reversed(sr.refine(models, psn, inp["images_lr"], inp.get("label_vecs"), text_priors=inp["text_priors"], return_all=True)))
base = run(); base2 = run()
torch.cuda.synchronize()
print("repeatable alone:", all(torch.equal(base[k], base2[k]) for k in base))
side = torch.cuda.Stream(dev)
for pattern in (0x7fc00000, 0x7f800000, 0x3f803f80, 0x7f7f7f7f):
    worst = {}
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(400):
                _abi.check(_abi.lib.dpmn_selftest_lds_poison(pattern, 256, _abi.stream()))
        got = run()
        torch.cuda.synchronize()
        for k in base:
            if not torch.equal(got[k], base[k]):
                d = float((got[k] - base[k]).abs().max())
                worst[k] = max(worst.get(k, 0.0), d if d == d else float("inf"))
    print("pattern %08x: stages that changed:" % pattern, {k: "%.1e" % v for k, v in worst.items()})

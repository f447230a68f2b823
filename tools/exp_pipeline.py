"""Experiment: two independent cfg1 stacks (own modules, own caches) stepping alternately on two HIP streams vs one stack on one lane:
how much forward throughput is left in overlapping consecutive batches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload
dev = torch.device("cuda:0")
stacks = [workload.build("cfg1") for _ in range(2)]
lanes = [torch.cuda.Stream(dev) for _ in range(2)]
from dpmn_amd.interfaces import super_resolution as srm
_own = {}
def _side(device):      # per-lane branch streams (the library shares one pair per device)
    key = torch.cuda.current_stream(device).cuda_stream
    if key not in _own:
        _own[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _own[key]
srm.side_streams = _side
def step(i):
    sr, models, psn, inp = stacks[i]
    return sr.refine(models, psn, inp["images_lr"], inp.get("label_vecs"), text_priors=inp["text_priors"])
def run(n, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        i = k % 2 if two else 0
        if two:
            with torch.cuda.stream(lanes[i]):
                step(i)
        else:
            step(0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for two in (False, True, False, True):
    run(6, two)
    print("two lanes" if two else "one lane ", "%.3f ms per step" % run(30, two))

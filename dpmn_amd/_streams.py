"""Process-wide HIP streams with a FIXED creation order.

HIP maps streams onto a handful of hardware queues in creation order (four by default; with GPU_MAX_HW_QUEUES=8 the training step
measured 30-35 ms instead of 27), so which streams share a queue -- and serialise -- depends on when they were created.  The four
streams of the training step are therefore created together, first, in the order a stand-alone training process creates them
(branch 1, branch 2, PSN prefetch lane, conv weight-gradient stream): whatever ran earlier in the process (bench.py times the
forward pipeline first), the step gets the same queue assignment; four streams are one full round of the queues, so everything
created afterwards (forward lanes and their branch streams) keeps the assignment it has in a forward-only process.
"""
import torch

_POOL = {}


def pool(device):
    key = (device.type, device.index)
    p = _POOL.get(key)
    if p is None:
        b1, b2 = torch.cuda.Stream(device), torch.cuda.Stream(device)
        psn = torch.cuda.Stream(device)
        wgrad = torch.cuda.Stream(device)
        p = _POOL[key] = dict(branch=(b1, b2), psn=psn, wgrad=wgrad)
    return p

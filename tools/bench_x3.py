"""Kernel times of the "f32 via bf16x3" variants next to the fp32-MFMA and plain-bf16 kernels at the bench shapes (B = 48).
usage: python tools/bench_x3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import _abi, ops
from dpmn_amd.utils import synth

dev = torch.device("cuda:0")


def u(name, shape, lo=-1.0, hi=1.0):
    return synth.uniform(name, shape, lo, hi, 70).to(dev)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e3


def modes(name, fn, flops):
    row = []
    for m in (0, 1, 2):
        _abi.check(_abi.lib.dpmn_set_compute_dtype(m))
        us = timeit(fn)
        row.append("%s %7.1f us %6.1f TF" % (("f32", "bf16", "x3")[m], us, flops / us / 1e6))
    _abi.check(_abi.lib.dpmn_set_compute_dtype(0))
    print("%-40s %s" % (name, " | ".join(row)), flush=True)


for B, Ch, L in ((48, 384, 1024), (96, 768, 4096)):
    g, w, b = u("g", (B, L, Ch)), u("w", (Ch, Ch), -0.2, 0.2), u("b", (Ch,))
    modes("pointwise B=%d Ch=%d L=%d" % (B, Ch, L), lambda: ops.pointwise(g, w, b), 2.0 * Ch * Ch * L * B)
    del g
if len(sys.argv) > 1 and sys.argv[1] == "pw":
    sys.exit(0)
try:
    import bench_x3_convs      # noqa: F401  (tools/bench_x3_convs.py: the conv families, once they have the variant)
except ImportError:
    pass

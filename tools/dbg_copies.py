"""Which host lines issue the `__amd_rocclr_copyBuffer` launches of a step?  (hipMemcpyAsync: torch's contiguous copy_ / clone, and
every host -> device transfer of a small table.)  One forward (sr.refine) and one training step of the cfg1 workload under
torch.profiler with Python stacks; prints, per (op, innermost dpmn_amd / bench frame), the number of memcpy launches.
  python tools/dbg_copies.py [fwd|train]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

import bench
from dpmn_amd import workload, _abi

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if mode == "fwd":
    sr, models, psn, inp = workload.build("cfg1", batch=None)
    kw = dict(text_priors=inp["text_priors"])

    def step():
        return sr.refine(models, psn, inp["images_lr"], inp.get("label_vecs"), **kw)
else:
    import torch.distributed as dist
    args = bench.parse.__wrapped__() if hasattr(bench.parse, "__wrapped__") else None
    sys.argv = [sys.argv[0], "--mode", "train"]
    args = bench.parse()
    step, trainer, B = bench.build_train_step(args, workload, 1, False, dist, torch, args.drop)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
by = collections.Counter()
names = collections.Counter()
for e in ev:
    n = e.name
    if "Memcpy" in n or "memcpy" in n or "copyBuffer" in n:
        names[n] += 1
for e in ev:
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::to", "aten::fill_", "aten::zero_", "aten::cat", "aten::empty_like") and e.stack:
        fr = [s for s in e.stack if ("dpmn_amd" in s or "bench.py" in s) and "dbg_copies" not in s]
        by[(e.name, fr[0] if fr else (e.stack[0] if e.stack else "?"))] += 1
print("memcpy-like device events:", dict(names))
for (n, fr), c in by.most_common(60):
    print("%4d  %-18s %s" % (c, n, fr[-110:]))

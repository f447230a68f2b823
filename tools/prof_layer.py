"""Time single CMM conv layers at the bench batch (event-timed, many repetitions): python tools/prof_layer.py
layers: en_3a en_4a en_4b en_5a en_5b de_5a de_4a de_3a (3x3 / 4x4 convs of cmm.py:38-77 at B = 48, twin encoder branches grouped)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops
from dpmn_amd.model import packing

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
LAYERS = {  # name: (segs, cout, k, stride, pad, dil, B, H, W, groups)
    "en_3a": ((128,), 128, 4, 2, 3, 2, 96, 16, 64, 2), "en_4a": ((256,), 256, 4, 2, 3, 2, 96, 8, 32, 2),
    "en_4b": ((256,), 512, 3, 1, 1, 1, 96, 4, 16, 2), "en_5a": ((512,), 512, 4, 2, 3, 2, 96, 4, 16, 2),
    "en_5b": ((512,), 512, 3, 1, 1, 1, 96, 2, 8, 2), "de_5a": ((512, 512, 512), 512, 3, 1, 1, 1, 48, 2, 8, 1),
    "de_4a": ((512, 512, 512), 256, 3, 1, 1, 1, 48, 4, 16, 1), "de_3a": ((256, 256, 256), 128, 3, 1, 1, 1, 48, 8, 32, 1),
}
names = sys.argv[1:] or list(LAYERS)
for n in names:
    segs, cout, k, stride, pad, dil, B, H, W, groups = LAYERS[n]
    xs = [torch.rand(B, H, W, c, generator=g).to(dev) for c in segs]
    cin = sum(segs)
    packs = [packing.pack_conv((torch.rand(cout, cin, k, k, generator=g) - 0.5).to(dev), torch.rand(cout, generator=g).to(dev)) for _ in range(groups)]
    if groups == 2:
        wp, bp = torch.stack([p[0] for p in packs]).contiguous(), torch.stack([p[1] for p in packs]).contiguous()
    else:
        wp, bp = packs[0]
    run = lambda: ops.conv2d(xs, wp, bp, cout, k, stride=stride, pad=pad, dil=dil, pro_act="leaky02", groups=groups)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    reps, best = 30, 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            run()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    us = best
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    fl = 2.0 * B * Ho * Wo * cout * k * k * cin
    print("%-6s %7.1f us %6.1f TF" % (n, us, fl / us / 1e6))
